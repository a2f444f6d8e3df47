#!/bin/bash
# Execution-unit counters of the heavy silhouette kernels (SQ / LDS / TCP), one rocprofv3 --pmc pass per counter set over
# tools/bench_raster.py (kernel-trace only).  Usage (GPU box): bash tools/pmc_units.sh > profiles/rNN_pmc_units.txt
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/pmc_units; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_LDS_ADDR_CONFLICT TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $O -o p$i -- python $R/tools/bench_raster.py 5 both > /dev/null 2>&1
done
cd $R
python - <<PY
import glob, sqlite3, re, collections
tab = collections.defaultdict(dict)
for db in sorted(glob.glob("$O/*.db")):
    c = sqlite3.connect(db)
    rows = c.execute("select E.name, E.counter_name, E.dispatch_id, sum(E.counter_value) from pmc_events E group by E.dispatch_id, E.counter_name order by E.dispatch_id").fetchall()
    per = collections.defaultdict(list)
    for name, cn, _, v in rows:
        m = re.match(r"(?:void )?(\w+)", name)
        per[(m.group(1) if m else name, cn)].append(v)
    for (k, cn), vals in per.items():
        vals = vals[-5:]                      # the measured loop, not the launches that synthesise the clip
        tab[k][cn] = sum(vals) / len(vals)
print("# per-launch means over the 5 measured launches of tools/bench_raster.py (cfg2 clip, cold forward + backward);")
print("# SQ_* cycle counters are quad-cycles summed over all waves / SIMDs, see /opt/skills/guides/MI355X_MICROARCH.md")
for k in ("k_raster_fwd", "k_bwd_lines", "k_bwd_sweep"):
    print(k)
    for cn in sorted(tab[k]):
        print(f"   {cn:34s} {tab[k][cn]:16.1f}")
PY
rm -rf $O
