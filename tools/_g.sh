R=$(pwd); O=$R/gpurun_out; mkdir -p $O
dep() { env "$@" python bench.py --depth --multi-clip 0 --no-cpu-baseline --legs '' 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   $1 depth %.0f steady %.0f final %.6f' % (d['value'], d['steady_state']['value'], d['final_loss']))"; }
dep HOMAN_DEPTH_CALIBRATE=0
dep HOMAN_DEPTH_CALIBRATE=h
dep HOMAN_DEPTH_CALIBRATE=o
dep HOMAN_DEPTH_CALIBRATE=1
dep HOMAN_DEPTH_CALIBRATE=0
dep HOMAN_DEPTH_CALIBRATE=h
