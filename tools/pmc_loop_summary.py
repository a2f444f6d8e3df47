#!/usr/bin/env python
"""Summarise the rocprofv3 --pmc passes of tools/pmc_loop.sh into the JSON bench.py reads (profiles/rNN_pmc_loop.json).
traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch: the counters are in KB, FETCH_SIZE is doubled as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (it tallies 128-byte requests at 64 B), WRITE_SIZE is used
as reported (uncalibrated).  SQ_* cycle counters are quad-cycles summed over all waves / SIMDs."""
import collections
import glob
import json
import os
import re
import sqlite3
import sys


def main(out_dir, last):
    tab = collections.defaultdict(dict)
    shape = None
    for log in sorted(glob.glob(os.path.join(out_dir, "run*.log"))):
        for line in open(log):
            if line.startswith("{"):
                j = json.loads(line)
                shape = dict(frames=j["frames"], rend_size=j["rend_size"], faces=j["faces"], step2=j["step2"])
                if j.get("depth"):
                    shape["depth"] = True
                clips = j["clips"]
    for db in sorted(glob.glob(os.path.join(out_dir, "*.db"))):
        c = sqlite3.connect(db)
        rows = c.execute("select E.name, E.counter_name, E.dispatch_id, sum(E.counter_value) from pmc_events E "
                         "group by E.dispatch_id, E.counter_name order by E.dispatch_id").fetchall()
        per = collections.defaultdict(list)
        for name, cn, _, v in rows:
            m = re.match(r"(?:void )?(\w+)", name)
            per[(m.group(1) if m else name, cn)].append(v)
        merged = any(k == "k_raster_fwd_multi" for k, _ in per)      # round 6: silhouette + object depth render as ONE launch pair
        for (k, cn), vals in per.items():
            if k.startswith("k_"):
                if shape and shape.get("depth") and merged and k in ("k_raster_fwd", "k_setup_faces", "k_raster_fwd_multi",
                                                                     "k_setup_faces_multi"):
                    # hm_sil_fwd_multi: the multi kernels hold the silhouette render AND the object's depth render (reported under
                    # the plain kernel's name + "#sil+obj_depth"), the plain kernels are the hand's depth render
                    nm = k[:-6] + "#sil+obj_depth" if k.endswith("_multi") else k + "#hand_depth"
                    vals = vals[-last:]
                    tab[nm][cn] = sum(vals) / len(vals)
                    tab[nm]["launches_averaged"] = len(vals)
                    continue
                if shape and shape.get("depth") and k in ("k_raster_fwd", "k_setup_faces", "k_depth_bwd_faces", "k_depth_bwd_gather"):
                    # with the ordinal depth term an iteration holds this kernel once per render: the silhouette's (first in
                    # dispatch order), the object's depth render, the hand's - averaged apart (k#obj_depth, k#hand_depth)
                    per_it = 3 if k in ("k_raster_fwd", "k_setup_faces") and not merged else 2
                    names = ([k, k + "#obj_depth", k + "#hand_depth"] if per_it == 3 else [k + "#hand_depth", k + "#obj_depth"])
                    vals = vals[len(vals) % per_it:]
                    for r, nm in enumerate(names):
                        sub = vals[r::per_it][-last:]
                        if sub:
                            tab[nm][cn] = sum(sub) / len(sub)
                            tab[nm]["launches_averaged"] = len(sub)
                    continue
                vals = vals[-last:]
                tab[k][cn] = sum(vals) / len(vals)
                tab[k]["launches_averaged"] = len(vals)
    for k, t in tab.items():
        if "FETCH_SIZE" in t and "WRITE_SIZE" in t:
            t["traffic_bytes"] = int((2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024)
    keep = ("k_raster_fwd", "k_bwd_lines", "k_bwd_sweep", "k_setup_faces", "k_mano_fwd", "k_mano_bwd", "k_nn", "k_rigid_bwd",
            "k_rigid_bwd_x", "k_pair_terms", "k_ordinal_depth", "k_ordinal_depth_bwd", "k_depth_bwd_faces", "k_depth_bwd_gather")
    print(json.dumps(dict(note=__doc__.strip(), shape=shape, clips=clips,
                          per_launch={k: tab[k] for k in sorted(tab) if k.split("#")[0] in keep}), indent=1))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
