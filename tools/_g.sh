R=$(pwd); O=$R/gpurun_out; mkdir -p $O
FC=$O/fetch_calib; rm -rf $FC; mkdir -p $FC
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $FC -o f -- $R/tools/fetch_calib > $FC/requested.json 2>$FC/err1.txt
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $FC -o w -- $R/tools/fetch_calib > /dev/null 2>$FC/err2.txt )
ls -R $FC | head; python tools/fetch_calib_summary.py $FC | tee $O/r06_fetch_calib.json | python -c "
import json,sys; d=json.load(sys.stdin)['kernels']
for k,v in d.items(): print(k, {a:b for a,b in v.items() if 'over' in a or 'per_access' in a})"
