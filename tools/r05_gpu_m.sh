set -u
O=gpurun_out/r05m; mkdir -p $O
A="--steps 20 --warmup 3 --no-cpu-baseline --no-latency --no-extras"
for rep in 1 2; do
for v in base wgscope; do
  PWN_HIP_LIB=$PWD/build/variants/$v.so timeout -k 10 120 python bench.py $A > $O/${v}_$rep.json 2> $O/${v}_$rep.err; echo "$v rc $?"
  python -c "
import json; l=json.loads(open('$O/${v}_$rep.json').read().strip().splitlines()[-1]); print('$v rep $rep: %.0f/s %.3f ms integral %.3f stats %.3f' % (l['value'], l['ms_per_step'], l['stage_ms_per_step']['integral'], l['stage_ms_per_step']['stats']), l['gather']['records_vs_single_gpu_run']['equal'])" || tail -3 $O/${v}_$rep.err
done
done
