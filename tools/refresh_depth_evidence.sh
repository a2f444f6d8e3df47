R=/root/repo; N=r06; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/pmc_loop.sh --depth > $O/${N}_pmc_loop_depth.json 2>/dev/null
cp $O/${N}_pmc_loop_depth.json profiles/${N}_pmc_loop_depth.json
HOMAN_BENCH_DETAIL=$O/${N}_bench_cfg2_depth.json python bench.py --depth --multi-clip 4 > $O/${N}_bench_cfg2_depth_line.json 2> $O/${N}_bench_cfg2_depth.err
HOMAN_BENCH_DETAIL=$O/${N}_bench_cfg2.json python bench.py --parity > $O/${N}_bench_cfg2_line.json 2> $O/${N}_bench_cfg2.err
HOMAN_BENCH_DETAIL=$O/${N}_bench_cfg2_driver_flags.json python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${N}_bench_cfg2_driver_flags_line.json 2> $O/${N}_bench_cfg2_driver_flags.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/pd -o pd -- python $R/bench.py --depth --multi-clip 0 --no-cpu-baseline --steady 0 --legs '' > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/ph -o ph -- python $R/bench.py --multi-clip 0 --no-cpu-baseline --legs '' > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $O/p3 -o p3 -- python $R/bench.py --step2 --multi-clip 0 --no-cpu-baseline --steady 0 --legs '' > /dev/null 2>&1
cd $R
python tools/prof_summary.py $O/pd/pd_results.db "python bench.py --depth --multi-clip 0 --no-cpu-baseline --steady 0 --legs ''" > $O/${N}_p_cfg2_depth_kernel_stats.txt
python tools/prof_timeline.py $O/pd/pd_results.db > $O/${N}_p_cfg2_depth_timeline.txt
python tools/prof_summary.py $O/ph/ph_results.db "python bench.py --multi-clip 0 --no-cpu-baseline --legs ''" > $O/${N}_p_cfg2_headline_kernel_stats.txt
python tools/prof_timeline.py $O/ph/ph_results.db > $O/${N}_p_cfg2_headline_timeline.txt
python tools/prof_summary.py $O/p3/p3_results.db "python bench.py --step2 --multi-clip 0 --no-cpu-baseline --steady 0 --legs ''" > $O/${N}_p_cfg3_kernel_stats.txt
python tools/prof_timeline.py $O/p3/p3_results.db > $O/${N}_p_cfg3_timeline.txt
rm -rf $O/pd $O/ph $O/p3
