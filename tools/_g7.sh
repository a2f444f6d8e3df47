# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g7.sh')
R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests/test_raster_gpu.py tests/test_poseinit.py tests/test_handchain_gpu.py -q -m gpu -x -k "not tied and not two_hands" > $O/g67.log 2>&1; tail -3 $O/g67.log | cut -c1-300
F="--steps 400 --warmup 20 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 1000 --freerun 0 --e2e-clips 0 --multi-clip 0"
for i in 1 2; do
HOMAN_POSEINIT_LOOPS=fused python bench.py --pose-init 500 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pose', round(d['value']), {k:round(v['avg_launch_us']) for k,v in d['roofline']['kernels'].items()})"
python bench.py $F 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['value']), round(d['steady_state']['value']), {k:round(v['avg_launch_us'],1) for k,v in d['roofline']['kernels'].items()})"
done
python tools/bench_clips.py --clips 8 --steps 200 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch8', round(d['its_per_s']))"
