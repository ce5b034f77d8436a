for sub in 64 32 16; do
for lib in build/variants/*.so; do
  PWN_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency --no-extras --sub-frames $sub > gpurun_out/b.json 2>gpurun_out/b.err || { echo "$lib FAILED"; tail -3 gpurun_out/b.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/b.json')); s=d['stage_ms_per_step']; print('sub', $sub, '$lib', round(d['value']), 'ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in s.items() if v})"
done
done
