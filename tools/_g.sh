R=$(pwd); O=$R/gpurun_out; mkdir -p $O
V=$R/variants
run() { echo "== $*"; env $1 python tools/bench_clips.py --clips 8 --steps 150 --warmup 30 --sweep-blocks $2 --stamps 20 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['its_per_s']), d.get('in_graph_us'))"; }
run X=1 1024
run X=1 1280
run HOMAN_AMD_LIB=$V/lib_slim.so 1280
run HOMAN_AMD_LIB=$V/lib_slim.so 1536
run HOMAN_AMD_LIB=$V/lib_slim6.so 1024
run HOMAN_AMD_LIB=$V/lib_slim6.so 1280
run HOMAN_AMD_LIB=$V/lib_slim6.so 1536
run X=1 1024
