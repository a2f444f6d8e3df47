R=$(pwd); O=$R/gpurun_out
python - > $O/g31_lock.json 2>$O/g31_lock.err <<'PY'
import json, sys
sys.path.insert(0, '.')
import bench
from homan_amd.mano_assets import synthetic_mano
mano = synthetic_mano(0)
res = {}
for name, kw in [("cfg2", dict(step2=False, steps=50)), ("cfg2_depth", dict(step2=False, steps=24, ordinal_depth=True)),
                 ("cfg3", dict(step2=True, steps=50))]:
    out = bench.lockstep_parity(mano, free_run=False, **kw)
    out.pop("per_step", None)
    res[name] = out
print(json.dumps(res, default=str))
PY
tail -3 $O/g31_lock.err
