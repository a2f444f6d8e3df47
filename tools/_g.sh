R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_known_answers.py tests/test_poseinit.py -x -q -m gpu 2>&1 | grep -E "passed|failed|rror" | tail -3
for rep in 1 2; do for v in s1 s2; do echo "== $v"; HOMAN_AMD_LIB=variants/lib_$v.so CHAIN_SKIP=1 python tools/chain_only.py cfg2 2>&1 | grep "shipped:\|main_only:";
 HOMAN_AMD_LIB=variants/lib_$v.so python bench.py --depth --multi-clip 8 --no-cpu-baseline --legs '' --steady 0 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   depth %.0f  batch(depth) %.0f' % (d['value'], d['multi_clip']['value']))"
 HOMAN_AMD_LIB=variants/lib_$v.so python tools/bench_clips.py --clips 8 --steps 200 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   batch8 %.0f' % d['its_per_s'])"
 HOMAN_AMD_LIB=variants/lib_$v.so python bench.py --pose-init 500 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   poseinit %.0f' % d['value'])"
done; done
