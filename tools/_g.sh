R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 1200 python -m pytest tests/test_raster_gpu.py tests/test_poseinit.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -4
python tools/ab_state.py save /tmp/conv.pt
drv() { env "$@" python bench.py --no-cpu-baseline --steady 0 --multi-clip 0 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   $* drv %.0f' % d['value'])"; }
for v in nocomb "" p128 m16 nocomb "" p128 m16; do
  if [ -n "$v" ]; then L="HOMAN_AMD_LIB=$R/variants/lib_$v.so"; else L="X=1"; fi
  for c in 1 8; do env $L python tools/ab_state.py time /tmp/conv.pt $c 2>/dev/null | tail -1; done
  drv $L
done
