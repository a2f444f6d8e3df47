R=$(pwd); O=$R/gpurun_out; mkdir -p $O; N=r05
timeout 1500 python -m pytest tests/test_depth_gpu.py tests/test_handchain_gpu.py tests/test_lockstep_gpu.py tests/test_clip_fitter_gpu.py tests/test_ortho.py tests/test_render_gpu.py -x -q -m gpu 2>&1 | tail -3
HOMAN_BENCH_DETAIL=$O/${N}_bench_cfg2_depth.json python bench.py --depth --multi-clip 4 > $O/${N}_bench_cfg2_depth_line.json 2> $O/${N}_bench_cfg2_depth.err
tail -c 600 $O/${N}_bench_cfg2_depth_line.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/pd -o pd -- python $R/bench.py --depth --multi-clip 0 --no-cpu-baseline --steady 0 > /dev/null 2>&1
cd $R
python tools/prof_summary.py $O/pd/pd_results.db "python bench.py --depth --multi-clip 0 --no-cpu-baseline --steady 0" > $O/${N}_p_cfg2_depth_kernel_stats.txt
python tools/prof_timeline.py $O/pd/pd_results.db > $O/${N}_p_cfg2_depth_timeline.txt
rm -rf $O/pd
