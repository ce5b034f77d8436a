mkdir -p gpurun_out
for cfg in "--streams 2 --sub-pairs 64 --sub-frames 64" "--streams 1 --sub-pairs 128 --sub-frames 128" "--streams 2 --sub-pairs 32 --sub-frames 64" "--streams 3 --sub-pairs 32 --sub-frames 64" "--streams 4 --sub-pairs 32 --sub-frames 64" "--streams 2 --sub-pairs 64 --sub-frames 128"; do
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-latency --no-extras --no-profile $cfg > gpurun_out/b.json 2>gpurun_out/b.err || { echo "$cfg FAILED"; tail -3 gpurun_out/b.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/b.json')); print('$cfg', round(d['value']), 'ms/step', round(d['ms_per_step'],2))"
done
