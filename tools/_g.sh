R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | cut -c1-400
