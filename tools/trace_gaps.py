"""Where is the GPU idle?  From a rocprofv3 kernel trace of bench.py: the steady-state window (as tools/trace_summary.py),
every gap between consecutive kernels (end -> next start) longer than `mingap` us, grouped by the pair (kernel before, kernel
after), plus the split of the window into busy / short gaps (< mingap: back-to-back launch latency) / long gaps.
usage: trace_gaps.py <kernel_trace.csv> <warmup> [mingap_us] [top]"""
import collections
import csv
import sys

path, warm = sys.argv[1], int(sys.argv[2])
mingap = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
top = int(sys.argv[4]) if len(sys.argv) > 4 else 30
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
g = [r for r in rows if "gagm_kernel" in r["Kernel_Name"]]
# the window opens 15 ms before the first timed solver launch (the TTA forward up to the solver is ~12 ms), never before the last
# warm-up solver launch: the warm-up Dice pass with its first-use kernel builds stays out
t0 = max(g[warm - 1]["e"], g[warm]["s"] - 15_000_000) if warm > 0 and len(g) > warm else rows[0]["s"]
win = [r for r in rows if r["s"] >= t0]
span = (win[-1]["e"] - win[0]["s"]) / 1e3
busy = 0.0
cur_end = win[0]["s"]
gaps = collections.defaultdict(lambda: [0.0, 0])
short = long_ = 0.0
nshort = nlong = 0
prev = None
for r in win:
    if r["s"] > cur_end:
        gap = (r["s"] - cur_end) / 1e3
        if gap >= mingap:
            key = ((prev["Kernel_Name"].split("(")[0][:48] if prev else "-"), r["Kernel_Name"].split("(")[0][:48])
            gaps[key][0] += gap
            gaps[key][1] += 1
            long_ += gap
            nlong += 1
        else:
            short += gap
            nshort += 1
    busy += max(0, r["e"] - max(r["s"], cur_end)) / 1e3
    if r["e"] > cur_end:
        cur_end, prev = r["e"], r
nsteps = max(1, len(g) - warm)
print("window %.1f ms over %d adapted batches: busy %.1f ms (%.1f %%), %d gaps < %.0f us = %.1f ms (%.1f %%), %d gaps >= %.0f us = %.1f ms (%.1f %%); %d kernels"
      % (span / 1e3, nsteps, busy / 1e3, 100 * busy / span, nshort, mingap, short / 1e3, 100 * short / span, nlong, mingap, long_ / 1e3, 100 * long_ / span, len(win)))
print("%-50s -> %-50s %6s %10s %9s" % ("kernel before the gap", "kernel after", "count", "total us", "avg us"))
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%-50s -> %-50s %6d %10.1f %9.1f" % (a, b, c, t, t / c))
