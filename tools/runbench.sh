timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency --align-only > gpurun_out/b.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/b.json')); s=d['stage_ms_per_step']; print(round(d['value']), round(d['align_only_alignments_per_s']), {k: round(v,2) for k,v in s.items()}, 'frac', round(d['roofline']['frac'],3))"
