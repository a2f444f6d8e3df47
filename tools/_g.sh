# scratch GPU script of the round (A/B on one box): tools/_g.sh [test files...]
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
F="--no-cpu-baseline --steady 1000"
run() { # name, env...
  n=$1; shift
  env "$@" HOMAN_BENCH_DETAIL=$O/ab_$n.detail.json python bench.py $F > $O/ab_$n.json 2> $O/ab_$n.err
  env "$@" HOMAN_BENCH_DETAIL=$O/ab_${n}_drv.detail.json python bench.py $F --steps 20 --warmup 5 --multi-clip 0 > $O/ab_${n}_drv.json 2>> $O/ab_$n.err
  python - <<PY
import json
a=json.load(open("$O/ab_$n.detail.json")); d=json.load(open("$O/ab_${n}_drv.detail.json"))
ks=lambda r: " ".join("%s %.1f" % (k[2:], v["avg_launch_us"]) for k,v in r["kernels"].items())
print("$n: headline %.0f steady %.0f batch %.0f | drv %.0f | steady: %s | drv: %s" % (a["value"], a["steady_state"]["value"], (a["multi_clip"] or {}).get("value",0), d["value"], ks(a["steady_state"]["roofline"]), ks(d["roofline"])))
PY
}
if [ -n "$1" ]; then timeout 1200 python -m pytest "$@" -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3; fi
run base HOMAN_AMD_LIB=$R/homan_amd/lib/lib_base.so
run kb4 X=1
run kb2 HOMAN_AMD_LIB=$R/homan_amd/lib/lib_kb2.so
run base2 HOMAN_AMD_LIB=$R/homan_amd/lib/lib_base.so
run kb4b X=1
run kb2b HOMAN_AMD_LIB=$R/homan_amd/lib/lib_kb2.so
