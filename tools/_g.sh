R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_clip_batch_gpu.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 900 python -m pytest tests/test_handchain_gpu.py -x -q -m gpu -k "min" 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/measure_test_bars.py 2>/dev/null > $O/r06_test_bars.json; python -c "
import json; d=json.load(open('$O/r06_test_bars.json'))
for k,v in d.items(): print(k, {a: float('%.3g' % b) for a,b in v.items()})"
b() { env "$@" python tools/bench_clips.py --clips 8 --steps 200 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   $*', {k: d[k] for k in d if 'it' in k or 'ms' in k})"; }
b HOMAN_NN_SEED=0
b HOMAN_NN_SEED=1
b HOMAN_NN_SEED=1 HOMAN_MANO_BWD_RIGID=1
b HOMAN_NN_SEED=0
b HOMAN_NN_SEED=1
b HOMAN_NN_SEED=1 HOMAN_MANO_BWD_RIGID=1
for s in 0 1; do echo "seed $s"; HOMAN_NN_SEED=$s CHAIN_SKIP=1 python tools/chain_only.py cfg2 2>&1 | grep "shipped:"; done
