R=$(pwd); O=$R/gpurun_out; mkdir -p $O
V=$R/variants
HOMAN_AMD_LIB=$V/lib_dbfp.so timeout 600 python -m pytest tests/test_depth_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python -m pytest tests/test_depth_gpu.py -x -q -m gpu 2>&1 | tail -2
dep() { env "$@" python bench.py --depth --multi-clip 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   $1 depth %.0f steady %.0f batch4 %.0f' % (d['value'], d['steady_state']['value'], d['multi_clip']['value']))"; }
dep X=1
dep HOMAN_AMD_LIB=$V/lib_dbfp.so
dep HOMAN_AMD_LIB=$V/lib_dbfp2.so
dep X=1
dep HOMAN_AMD_LIB=$V/lib_dbfp.so
dep HOMAN_AMD_LIB=$V/lib_dbfp2.so
