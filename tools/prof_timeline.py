#!/usr/bin/env python
"""Timeline of ONE optimisation iteration out of a rocprofv3 kernel trace (rocpd SQLite): kernels between two
consecutive launches of a marker kernel (default k_adam), with start offsets, durations and the HIP queue they ran on.
Usage: python tools/prof_timeline.py gpurun_out/prof_x/x_results.db [marker] [iteration index from the end]
Default: an iteration from the MIDDLE of the run (the timed region of bench.py) whose span is the median of its neighbours' - the
last iterations of a bench run belong to the roofline-stamp replays, which carry two memset nodes and a copy node each."""
import sqlite3
import sys


def main(path, marker="k_adam", back=None):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select s.kernel_name, d.start, d.end, d.queue_id, d.stream_id from rocpd_kernel_dispatch d "
        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if back is None:
        # the median-length iteration among the 21 around the middle of the run
        mid = len(marks) // 2
        cand = [k for k in range(max(1, mid - 10), min(len(marks) - 1, mid + 11))]
        span = lambda k: rows[marks[k]][2] - rows[marks[k - 1]][2]
        k = sorted(cand, key=span)[len(cand) // 2]
        back = len(marks) - k
    a, b = marks[-back - 1], marks[-back]
    t0 = rows[a][2]
    print(f"# iteration between {marker} launches #{len(marks)-back-1} and #{len(marks)-back}: "
          f"{(rows[b][2]-t0)/1e3:.1f} us end to end")
    print(f"{'start_us':>9s} {'dur_us':>8s} {'end_us':>8s} {'q':>3s} {'s':>3s}  kernel")
    for name, st, en, q, sid in rows[a + 1:b + 1]:
        short = name.split("(")[0]
        if short.startswith("_Z"):
            import re
            m = re.match(r"_Z(\d+)", short)
            if m:                                    # (plain names only; templates / namespaces are printed mangled)
                n = int(m.group(1))
                short = short[2 + len(m.group(1)):2 + len(m.group(1)) + n]
        print(f"{(st-t0)/1e3:9.1f} {(en-st)/1e3:8.1f} {(en-t0)/1e3:8.1f} {q:3d} {sid:3d}  {short[:60]}")


if __name__ == "__main__":
    main(sys.argv[1], *(sys.argv[2:3] or ["k_adam"]), *([int(sys.argv[3])] if len(sys.argv) > 3 else []))
