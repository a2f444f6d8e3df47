# scratch GPU script of the build sessions (gpurun -- 'bash tools/_g7.sh')
R=$(pwd); O=$R/gpurun_out
timeout 1500 python -m pytest tests/test_handchain_gpu.py -q -k "free_object_scale or tied_object_scale or two_hands" > $O/g61.log 2>&1; tail -4 $O/g61.log | cut -c1-300
