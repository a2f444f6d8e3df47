R=$(pwd); O=$R/gpurun_out; mkdir -p $O
V=$R/variants
HOMAN_AMD_LIB=$V/lib_slim.so timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_poseinit.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -4
python tools/ab_state.py save /tmp/conv.pt
drv() { env "$@" python bench.py --no-cpu-baseline --steady 0 --multi-clip 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   drv %.0f' % d['value'])"; }
run() { echo "== $*"; env "$@" python tools/ab_state.py time /tmp/conv.pt 1 2>/dev/null | tail -1; drv "$@"; env "$@" python tools/sil_wave_balance.py --at 200 2>&1 | grep -E "iter" | head -3; }
run X=1
run HOMAN_SWEEP_BLOCKS=1024
run HOMAN_SWEEP_BLOCKS=1152
run HOMAN_AMD_LIB=$V/lib_slim.so
run HOMAN_AMD_LIB=$V/lib_slim6.so
run X=1
run HOMAN_AMD_LIB=$V/lib_slim.so
for c in 8; do for v in "" slim; do if [ -n "$v" ]; then L="HOMAN_AMD_LIB=$V/lib_$v.so"; else L="X=1"; fi; env $L python tools/ab_state.py time /tmp/conv.pt $c 2>/dev/null | tail -1; done; done
