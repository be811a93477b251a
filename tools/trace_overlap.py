#!/usr/bin/env python3
"""Dev tool: how well do the kernels of several in-flight MSMs overlap?  Reads a rocprofv3 `--kernel-trace
--output-format csv` kernel trace and reports, over the steady-state part of the run (the last `--tail` fraction of
the dispatch time range): wall time, time with 0 / 1 / 2 / ... `k_accumulate` kernels resident, the time no kernel at
all is running, and per kernel: calls, average duration, share of the wall time in which it is the ONLY kernel running.
  python tools/trace_overlap.py <dir-or-csv> [--tail 0.6] [--focus k_accumulate]"""
import argparse
import csv
import glob
import os
import re


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0][-40:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--tail", type=float, default=0.6)
    ap.add_argument("--focus", default="k_accumulate")
    ap.add_argument("--first", type=int, default=-1, help="window = from the start of the N-th k_prepare (0-based) ...")
    ap.add_argument("--count", type=int, default=20, help="... to the end of the (N+count)-th k_final")
    ap.add_argument("--until-prepare", type=int, default=-1,
                    help="... or (batch mode: one k_final per batch) to the end of the last kernel started before the M-th k_prepare")
    ap.add_argument("--gaps", type=float, default=0.0, help="also list every interval of at least this many us with NO kernel running")
    ap.add_argument("--edges", type=int, default=0, help="also list the first / last N dispatches of the window (relative times)")
    a = ap.parse_args()
    files = [a.path] if os.path.isfile(a.path) else glob.glob(os.path.join(a.path, "**", "*kernel_trace.csv"), recursive=True)
    ev = []
    for f in files:
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    ev.sort()
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    if a.first >= 0:
        preps = [e for e in ev if e[2] == "k_prepare"]
        fins = sorted(e[1] for e in ev if e[2] == "k_final")
        if a.until_prepare >= 0:
            stop = preps[a.until_prepare][0] if a.until_prepare < len(preps) else max(e[1] for e in ev) + 1
            lo, hi = preps[a.first][0], max(e[1] for e in ev if e[0] < stop)
        else:
            lo, hi = preps[a.first][0], fins[a.first + a.count - 1]
        ev = [e for e in ev if e[0] >= lo and e[1] <= hi]
    else:
        lo = t1 - (t1 - t0) * a.tail
        ev = [e for e in ev if e[0] >= lo]
    t0, t1 = ev[0][0], max(e[1] for e in ev)
    pts = []
    for s, e, n in ev:
        pts.append((s, 1, n))
        pts.append((e, -1, n))
    pts.sort()
    live = {}
    hist, only, idle = {}, {}, 0
    prev = pts[0][0]
    for t, d, n in pts:
        dt = t - prev
        if dt > 0:
            k = live.get(a.focus, 0)
            hist[k] = hist.get(k, 0) + dt
            running = [x for x, c in live.items() if c > 0]
            if not running:
                idle += dt
            elif len(running) == 1 and live[running[0]] == 1:
                only[running[0]] = only.get(running[0], 0) + dt
        live[n] = live.get(n, 0) + d
        prev = t
    if a.gaps > 0:
        cur_end, last_name = ev[0][0], "(window start)"
        print("idle intervals >= %.0f us (offset from the window start, length, kernel that ended -> kernel that starts):" % a.gaps)
        for s_, e_, n_ in ev:
            if s_ - cur_end >= a.gaps * 1e3:
                print("  +%9.1f us  %7.1f us   %s -> %s" % ((cur_end - t0) / 1e3, (s_ - cur_end) / 1e3, last_name, n_))
            if e_ > cur_end:
                cur_end, last_name = e_, n_
    if a.edges > 0:
        for label, part in (("first", ev[:a.edges]), ("last", ev[-a.edges:])):
            print("%s %d dispatches (start, end in us from the window start):" % (label, len(part)))
            for s_, e_, n_ in part:
                print("  %9.1f %9.1f  %s" % ((s_ - t0) / 1e3, (e_ - t0) / 1e3, n_))
    wall = (t1 - t0) / 1e6
    print("steady-state window: %.3f ms, %d dispatches, %d %s" % (wall, len(ev), sum(1 for e in ev if e[2] == a.focus), a.focus))
    print("time with k %s resident: " % a.focus + "  ".join("%d: %.1f%%" % (k, 100 * v / (t1 - t0)) for k, v in sorted(hist.items())))
    print("no kernel running: %.1f%%" % (100 * idle / (t1 - t0)))
    tot, cnt = {}, {}
    for s, e, n in ev:
        tot[n] = tot.get(n, 0) + (e - s)
        cnt[n] = cnt.get(n, 0) + 1
    print("%-28s %6s %10s %10s %10s" % ("kernel", "calls", "avg_us", "sum_ms", "alone_%wall"))
    for n in sorted(tot, key=lambda x: -tot[x]):
        print("%-28s %6d %10.1f %10.3f %10.1f" % (n, cnt[n], tot[n] / cnt[n] / 1e3, tot[n] / 1e6, 100 * only.get(n, 0) / (t1 - t0)))


if __name__ == "__main__":
    main()
