# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g7.sh')
R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests/test_handchain_gpu.py -q -k inter_type_min > $O/g64.log 2>&1; tail -25 $O/g64.log | cut -c1-500
