R=$(pwd); O=$R/gpurun_out; mkdir -p $O
if [ -n "$1" ]; then timeout 1500 python -m pytest "$@" -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -4; fi
one() { env "$@" python bench.py --no-cpu-baseline --steady 1000 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'head %.0f steady %.0f batch %.0f' % (d['value'], d['steady_state']['value'], d['multi_clip']['value']))"
 env "$@" python bench.py --no-cpu-baseline --steady 0 --multi-clip 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   drv %.0f' % d['value'])"; }
one HOMAN_DEFER_FINISH=0
one HOMAN_DEFER_FINISH=1
one HOMAN_DEFER_FINISH=0
one HOMAN_DEFER_FINISH=1
