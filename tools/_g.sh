R=$(pwd); O=$R/gpurun_out; mkdir -p $O
r() { echo "== $*"; python tools/bench_clips.py --steps 400 --warmup 100 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_round %.4f  its/s %.0f' % (d['ms_per_round'], d['its_per_s']))"; }
r --clips 1 --frames 30
HOMAN_GRAPH_ITERS=1 r --clips 1 --frames 30
r --clips 2 --frames 15 --groups 2
r --clips 2 --frames 15
r --clips 3 --frames 10 --groups 3
r --clips 1 --frames 15
r --clips 2 --frames 30 --groups 2
r --clips 2 --frames 30
