R=$(pwd); O=$R/gpurun_out
for m in regions frames frames_centre regions; do
HOMAN_POSE_ORDER=$m HOMAN_POSEINIT_LOOPS=fused python bench.py --pose-init 500 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', round(d['value']), {k:round(v['avg_launch_us']) for k,v in d['roofline']['kernels'].items()})"
done
