#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into the per-kernel table committed under profiles/.
Usage: python tools/prof_summary.py gpurun_out/prof_x/x_results.db ["profiled command"] [> profiles/rNN_x.txt]"""
import sqlite3
import sys


def main(path, cmd=None):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), min(d.end-d.start), max(d.end-d.start), "
        "max(s.arch_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), "
        "max(d.grid_size_x*d.grid_size_y*d.grid_size_z/(d.workgroup_size_x*d.workgroup_size_y*d.workgroup_size_z)) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
        "group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    span = c.execute("select min(start), max(end) from rocpd_kernel_dispatch").fetchone()
    print(f"# source: {path}")
    if cmd:
        print(f"# command: rocprofv3 --kernel-trace --stats -- {cmd}")
    print(f"# kernels: {len(rows)}  dispatches: {sum(r[1] for r in rows)}  sum(kernel time) {total/1e6:.3f} ms  "
          f"trace span {(span[1]-span[0])/1e6:.3f} ms")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>9s} {'min_us':>8s} {'max_us':>8s} {'%':>6s} "
          f"{'vgpr':>5s} {'sgpr':>5s} {'lds':>6s} {'wgs':>6s}")
    for name, n, tot, mn, mx, vg, sg, lds, wgs in rows:
        short = name if len(name) <= 70 else name[:67] + "..."
        print(f"{short:70s} {n:7d} {tot/1e6:10.3f} {tot/n/1e3:9.2f} {mn/1e3:8.2f} {mx/1e3:8.2f} {100*tot/total:6.2f} "
              f"{vg or 0:5d} {sg or 0:5d} {lds or 0:6d} {wgs or 0:6d}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
