R=$(pwd); O=$R/gpurun_out; mkdir -p $O
timeout 1300 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/final_drv_line.json 2> $O/final_drv.err ) 2>&1 | grep real
tail -c 1800 $O/final_drv_line.json
( time python bench.py > $O/final_default_line.json 2> $O/final_default.err ) 2>&1 | grep real
tail -c 1800 $O/final_default_line.json
