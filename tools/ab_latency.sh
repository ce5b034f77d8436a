for lib in build/variants/*.so; do
  PWN_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > gpurun_out/b.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$lib', round(d['value']), 'single pair ms', round(d['single_pair_latency_ms'],3))"
  PWN_HIP_LIB=$PWD/$lib timeout 300 python tools/bench_tracker.py --frames 100 --scale 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  tracker fps', round(d['value']))"
done
