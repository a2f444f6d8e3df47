for d in 0 4 5; do cp scratch/libs$d.so homan_amd/lib/libhoman_amd.so; echo "== variant $d"; python scratch/sw.py 2>&1 | grep -E "dur ticks|span"; done
