"""Joins tools/fetch_calib's requested bytes with the FETCH_SIZE / WRITE_SIZE counters of its rocprofv3 --pmc passes:
    python tools/fetch_calib_summary.py <out_dir with *.db and requested.json>  ->  JSON (profiles/r06_fetch_calib.json)
Counters are in KB.  `fetch_over_requested` = FETCH_SIZE * 1024 / requested bytes (the guide: 0.5 for wide coalesced streams);
`bytes_per_access` columns say what ONE access costs the memory side as the counters see it."""
import collections
import glob
import json
import os
import re
import sqlite3
import sys


def main(out_dir):
    req = json.load(open(os.path.join(out_dir, "requested.json")))
    tab = collections.defaultdict(dict)
    for db in sorted(glob.glob(os.path.join(out_dir, "**", "*.db"), recursive=True)):
        c = sqlite3.connect(db)
        for name, cn, v in c.execute("select E.name, E.counter_name, sum(E.counter_value) from pmc_events E "
                                     "group by E.dispatch_id, E.counter_name").fetchall():
            m = re.match(r"(?:void )?(\w+)", name)
            k = m.group(1) if m else name
            if k in req:
                tab[k][cn] = v
    out = {}
    for k, r in req.items():
        t = tab.get(k, {})
        rec = dict(r)
        if "FETCH_SIZE" in t:
            rec["FETCH_SIZE_bytes"] = t["FETCH_SIZE"] * 1024
            rec["fetch_over_requested"] = round(t["FETCH_SIZE"] * 1024 / r["requested_bytes"], 3)
            rec["fetch_bytes_per_access"] = round(t["FETCH_SIZE"] * 1024 / r["accesses"], 2)
        if "WRITE_SIZE" in t:
            rec["WRITE_SIZE_bytes"] = t["WRITE_SIZE"] * 1024
            rec["write_over_requested"] = round(t["WRITE_SIZE"] * 1024 / r["requested_bytes"], 3)
            rec["write_bytes_per_access"] = round(t["WRITE_SIZE"] * 1024 / r["accesses"], 2)
        out[k] = rec
    print(json.dumps(dict(note=__doc__.strip(), kernels=out), indent=1))


if __name__ == "__main__":
    main(sys.argv[1])
