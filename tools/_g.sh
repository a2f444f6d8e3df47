# scratch GPU script of the round (A/B on one box)
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
F="--no-cpu-baseline --steady 1000"
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py $F > $O/ab_$n.json 2> $O/ab_$n.err
  env "$@" python bench.py $F --steps 20 --warmup 5 --multi-clip 0 > $O/ab_${n}_drv.json 2>> $O/ab_$n.err
  python - <<PY
import json
a=json.load(open("$O/ab_$n.json")); d=json.load(open("$O/ab_${n}_drv.json"))
print("$n: headline %.0f steady %.0f batch %.0f | drv %.0f | sweep %.1f us" % (a["value"], a["steady_state"]["value"], a["multi_clip"]["value"], d["value"], a["steady_state"]["dominant_kernel_us"]))
PY
}
run base X=1
run k4 HOMAN_GRAPH_ITERS=4
run dyn HOMAN_SWEEP_DYN=1
run dyn_k4 HOMAN_SWEEP_DYN=1 HOMAN_GRAPH_ITERS=4
run base2 X=1
HOMAN_SWEEP_DYN=1 HOMAN_GRAPH_ITERS=4 python -m pytest tests/test_raster_gpu.py tests/test_parity_gpu.py tests/test_clip_batch_gpu.py -x -q -m gpu 2>&1 | tail -3
