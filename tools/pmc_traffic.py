#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 PMC passes (rocpd SQLite): FETCH_SIZE and WRITE_SIZE, collected in
SEPARATE runs with --kernel-trace only, e.g. over `python tools/bench_raster.py 5 both`.
Counter unit = KB.  per_launch_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled per
/opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); WRITE_SIZE is used as reported.
Usage: python tools/pmc_traffic.py fetch.db write.db [last_n_launches] > profiles/rNN_pmc_traffic.json"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter, last=0):
    """kernel -> (mean counter value per launch, launches averaged); last > 0: only the last `last` launches of each
    kernel (the measured loop of tools/bench_raster.py, not the launches that synthesise the clip's target masks)."""
    c = sqlite3.connect(path)
    rows = c.execute("select E.name, E.counter_name, E.dispatch_id, sum(E.counter_value) from pmc_events E "
                     "group by E.dispatch_id, E.counter_name order by E.dispatch_id").fetchall()
    per = {}
    for name, cn, _, v in rows:
        if cn != counter:
            continue
        m = re.match(r"(?:void )?(\w+)", name)
        per.setdefault(m.group(1) if m else name, []).append(v)
    out = {}
    for k, vals in per.items():
        vals = vals[-last:] if last else vals
        out[k] = (sum(vals) / len(vals), len(vals))
    return out


def main(fetch_db, write_db, last=0):
    f = per_kernel(fetch_db, "FETCH_SIZE", last)
    w = per_kernel(write_db, "WRITE_SIZE", last)
    detail, per = {}, {}
    for k in sorted(set(f) & set(w)):
        if not k.startswith("k_"):
            continue
        fk, wk = f[k][0], w[k][0]
        detail[k] = dict(FETCH_SIZE_KB=fk, WRITE_SIZE_KB=wk, fetch_bytes_raw=int(fk * 1024), fetch_bytes_x2=int(2 * fk * 1024),
                         write_bytes=int(wk * 1024), launches=f[k][1])
        per[k] = int((2 * fk + wk) * 1024)
    print(json.dumps(dict(note=__doc__.strip(), per_launch_bytes=per, detail=detail), indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
