#!/bin/bash
# HBM traffic and execution-unit counters of the heavy silhouette kernels in the FUSED loop of the object-pose initialisation
# (500 candidate poses x one 256^2 mask): separate rocprofv3 --pmc passes (kernel-trace only) over `bench.py --pose-init`,
# averaged over the last launches of every kernel (= the fused loop, which bench.py runs last).
# Usage (GPU box): bash tools/pmc_poseinit.sh > profiles/rNN_pmc_poseinit.json
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/pmc_pi; rm -rf $O; mkdir -p $O
N=${1:-500}; LAST=20
cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"; do
  i=$((i+1))
  HOMAN_POSEINIT_LOOPS=fused rocprofv3 --kernel-trace --pmc $set -d $O -o p$i -- python $R/bench.py --pose-init $N --steps 30 --no-cpu-baseline > $O/run$i.log 2>&1
done
cd $R
python - "$O" "$LAST" "$N" <<'PY'
import collections, glob, json, os, re, sqlite3, sys
out_dir, last, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
tab = collections.defaultdict(dict)
for db in sorted(glob.glob(os.path.join(out_dir, "*.db"))):
    c = sqlite3.connect(db)
    rows = c.execute("select E.name, E.counter_name, E.dispatch_id, sum(E.counter_value) from pmc_events E "
                     "group by E.dispatch_id, E.counter_name order by E.dispatch_id").fetchall()
    per = collections.defaultdict(list)
    for name, cn, _, v in rows:
        m = re.match(r"(?:void )?(\w+)", name)
        per[(m.group(1) if m else name, cn)].append(v)
    for (k, cn), vals in per.items():
        if k.startswith("k_"):
            vals = vals[-last:]
            tab[k][cn] = sum(vals) / len(vals)
            tab[k]["launches_averaged"] = len(vals)
for k, t in tab.items():
    if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
        t["traffic_bytes"] = int((2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024)
keep = ("k_raster_fwd", "k_bwd_lines", "k_bwd_sweep", "k_setup_faces", "k_rigid_bwd", "k_rigid_fwd")
print(json.dumps(dict(note="rocprofv3 --pmc passes over bench.py --pose-init (fused loop); traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 "
                           "per launch as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950", shape=dict(poses=n, size=256, faces=3000),
                      per_launch={k: tab[k] for k in keep if k in tab}), indent=1))
PY
rm -rf $O
