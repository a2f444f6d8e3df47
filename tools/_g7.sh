R=$(pwd); O=$R/gpurun_out; N=r04
cd $R
bash tools/pmc_poseinit.sh > $O/${N}_pmc_poseinit.json 2>/dev/null
cp $O/${N}_pmc_poseinit.json profiles/${N}_pmc_poseinit.json
python bench.py --pose-init 500 > $O/${N}_bench_poseinit.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
POSE="HOMAN_POSEINIT_LOOPS=fused python bench.py --pose-init 500 --no-cpu-baseline"
HOMAN_POSEINIT_LOOPS=fused rocprofv3 --kernel-trace --stats -d $O/pp -o pp -- python $R/bench.py --pose-init 500 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/prof_summary.py $O/pp/pp_results.db "$POSE" > $O/${N}_p_poseinit_kernel_stats.txt
python tools/prof_timeline.py $O/pp/pp_results.db > $O/${N}_p_poseinit_timeline.txt
rm -rf $O/pp
python tools/poseinit_phases.py > $O/${N}_poseinit_phases.json 2>/dev/null
head -c 700 $O/${N}_bench_poseinit.json; echo; head -12 $O/${N}_p_poseinit_kernel_stats.txt | cut -c1-140
