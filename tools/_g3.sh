R=$(pwd); O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/pt -o pt -- python $R/tools/bench_clips.py --clips 1 --steps 400 --warmup 400 > $O/pt.json 2>/dev/null
cd $R
for back in 3 4 5 6 7; do python tools/prof_timeline.py $O/pt/pt_results.db k_adam $back; done > $O/timeline_k4.txt
cat $O/pt.json | tail -1
cat $O/timeline_k4.txt | head -45
rm -rf $O/pt
