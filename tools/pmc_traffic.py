#!/usr/bin/env python
"""HBM traffic per launch from two rocprofv3 PMC passes (rocpd SQLite): FETCH_SIZE and WRITE_SIZE, collected in
SEPARATE runs with --kernel-trace only, e.g. over `python tools/bench_raster.py 5 both`.
Counter unit = KB.  per_launch_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE is doubled per
/opt/skills/guides/MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read); WRITE_SIZE is used as reported.
Usage: python tools/pmc_traffic.py fetch.db write.db > profiles/rNN_pmc_traffic.json"""
import json
import re
import sqlite3
import sys


def per_kernel(path, counter):
    c = sqlite3.connect(path)
    rows = c.execute("select name, counter_name, avg(value), count(*) from (select E.name as name, E.counter_name as "
                     "counter_name, sum(E.counter_value) as value from pmc_events E group by E.dispatch_id, E.counter_name) "
                     "group by name, counter_name").fetchall()
    out = {}
    for name, cn, v, n in rows:
        if cn != counter:
            continue
        m = re.match(r"(?:void )?(\w+)", name)
        short = m.group(1) if m else name
        out[short] = (v, n)
    return out


def main(fetch_db, write_db):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
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
    main(sys.argv[1], sys.argv[2])
