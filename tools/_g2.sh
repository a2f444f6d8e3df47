R=$(pwd); O=$R/gpurun_out; mkdir -p $O
HOMAN_BENCH_DETAIL=$O/pi_new.detail.json python bench.py --pose-init 500 --no-cpu-baseline > $O/pi_new.json 2> $O/pi_new.err
HOMAN_AMD_LIB=$R/homan_amd/lib/lib_base.so HOMAN_BENCH_DETAIL=$O/pi_base.detail.json python bench.py --pose-init 500 --no-cpu-baseline > $O/pi_base.json 2> $O/pi_base.err
python - <<PY
import json
for n in ("base","new"):
    d=json.load(open("$O/pi_%s.detail.json" % n))
    print(n, "%.0f pose-steps/s" % d["value"], {k: round(v["avg_launch_us"],1) for k,v in d["roofline"]["kernels"].items()})
PY
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gputests.log 2>&1; tail -3 $O/gputests.log
