R=$(pwd); O=$R/gpurun_out; mkdir -p $O
V=$R/variants
timeout 600 python -m pytest tests/test_depth_gpu.py -x -q -m gpu 2>&1 | tail -2
dep() { env "$@" python bench.py --depth --multi-clip 4 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   $1 depth %.0f steady %.0f batch4 %.0f' % (d['value'], d['steady_state']['value'], d['multi_clip']['value']))"; }
dep HOMAN_AMD_LIB=$V/lib_dbf4.so
dep X=1
dep HOMAN_AMD_LIB=$V/lib_dbf4.so
dep X=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/pd -o pd -- python $R/bench.py --depth --multi-clip 0 --no-cpu-baseline --steady 0 > /dev/null 2>&1
cd $R
python tools/prof_summary.py $O/pd/pd_results.db "bench.py --depth" > $O/r05_p_cfg2_depth_kernel_stats.txt
python tools/prof_timeline.py $O/pd/pd_results.db > $O/r05_p_cfg2_depth_timeline.txt
rm -rf $O/pd
head -18 $O/r05_p_cfg2_depth_kernel_stats.txt | cut -c1-130
