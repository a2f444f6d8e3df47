R=$(pwd); O=$R/gpurun_out
timeout 900 python -m pytest tests/test_poseinit.py -q -m gpu > $O/g52.log 2>&1; tail -8 $O/g52.log | cut -c1-300
HOMAN_POSEINIT_LOOPS=fused python bench.py --pose-init 500 --no-cpu-baseline > $O/g52_bench.json 2>$O/g52b.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/g52_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['cold_fit'], {k:(v['avg_launch_us']) for k,v in d['roofline']['kernels'].items()})
PY
