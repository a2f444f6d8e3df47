R=$(pwd); O=$R/gpurun_out
timeout 900 python -m pytest tests/test_poseinit.py tests/test_raster_gpu.py -q -m gpu > $O/g53.log 2>&1; tail -4 $O/g53.log | cut -c1-300
HOMAN_POSEINIT_LOOPS=fused python bench.py --pose-init 500 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('poseinit', round(d['value']), {k:round(v['avg_launch_us']) for k,v in d['roofline']['kernels'].items()})"
F="--steps 400 --warmup 20 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 1000 --freerun 0 --e2e-clips 0 --multi-clip 0"
python bench.py $F 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['value']), round(d['steady_state']['value']), {k:round(v['avg_launch_us'],1) for k,v in d['roofline']['kernels'].items()})"
