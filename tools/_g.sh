R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_ortho.py -x -q 2>&1 | tail -15
for c in 4 8 12 16 24 32; do python tools/bench_clips.py --clips $c --steps 150 --warmup 30 2>/dev/null | tail -1 | cut -c1-300; done
for c in 16 32; do for sb in 1280 768; do echo "sb $sb"; python tools/bench_clips.py --clips $c --steps 150 --warmup 30 --sweep-blocks $sb 2>/dev/null | tail -1 | cut -c1-200; done; done
