R=$(pwd); O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/ph -o ph -- python $R/bench.py --multi-clip 0 --parity-seeds 0 --lockstep 0 --no-cpu-baseline --freerun 0 --e2e-clips 0 > /dev/null 2>&1
cd $R
python tools/prof_summary.py $O/ph/ph_results.db new > $O/g6_kernel_stats.txt
python tools/prof_timeline.py $O/ph/ph_results.db > $O/g6_timeline.txt
rm -rf $O/ph
