R=$(pwd); O=$R/gpurun_out
python -m pytest tests/test_clip_batch_gpu.py tests/test_dist_gpu.py -x -q -m gpu > $O/g17_t.log 2>&1
for i in 1 2; do
HOMAN_SHARD_CONCURRENT=0 python tools/bench_clips.py --clips 8 --steps 200 --mixed > $O/g17_seq_$i.json 2>/dev/null
python tools/bench_clips.py --clips 8 --steps 200 --mixed > $O/g17_con_$i.json 2>/dev/null
done
