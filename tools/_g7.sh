# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g7.sh')
R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests/test_depth_gpu.py -q > $O/g62.log 2>&1; tail -30 $O/g62.log | cut -c1-400
