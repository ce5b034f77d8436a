for cfg in "64 64 2" "128 64 2" "32 64 2" "64 128 1" "64 32 2" "128 128 1" "64 64 2"; do
  set -- $cfg
  timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-latency --no-profile --sub-frames $1 --sub-pairs $2 --streams $3 > gpurun_out/s.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/s.json')); print('sub_frames $1 sub_pairs $2 streams $3:', round(d['value']), round(d['ms_per_step'],2))"
done
