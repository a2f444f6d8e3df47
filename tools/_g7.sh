# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g7.sh')
R=$(pwd); O=$R/gpurun_out
F="--steps 400 --warmup 20 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --steady 1000 --freerun 0 --e2e-clips 0 --multi-clip 0"
for lib in "" variants/lib_nostore.so; do
env ${lib:+HOMAN_AMD_LIB=$lib} HOMAN_POSEINIT_LOOPS=fused python bench.py --pose-init 500 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pose $lib', round(d['value']), {k:round(v['avg_launch_us']) for k,v in d['roofline']['kernels'].items()})"
env ${lib:+HOMAN_AMD_LIB=$lib} python bench.py $F 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $lib', round(d['value']), round(d['steady_state']['value']), {k:round(v['avg_launch_us'],1) for k,v in d['roofline']['kernels'].items()})"
done
