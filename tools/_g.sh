# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g.sh')
R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_depth_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
dep() { env "$@" python bench.py --depth --multi-clip 0 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   $1 depth %.0f steady %.0f' % (d['value'], d['steady_state']['value']))"; }
dep HOMAN_DEPTH_SPARSE=0
dep HOMAN_DEPTH_SPARSE=1
dep HOMAN_DEPTH_SPARSE=0
dep HOMAN_DEPTH_SPARSE=1
