"""Summarise a rocprofv3 kernel trace over the steady-state window of bench.py: from the (warmup+1)-th
gagm_kernel dispatch to the end.  Usage: python tools/trace_summary.py <kernel_trace.csv> <warmup> [top]"""
import csv, sys, collections
path, warm = sys.argv[1], int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
rows = list(csv.DictReader(open(path)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
g = [r for r in rows if "gagm_kernel" in r["Kernel_Name"]]
t0 = g[warm]["s"] - 60_000_000 if len(g) > warm else rows[0]["s"]     # window opens ~60 ms before the first timed solver launch
# refine: start at the first backbone kernel after the previous solver launch ended
prev_end = g[warm - 1]["e"] if warm > 0 else rows[0]["s"]
t0 = max(t0, prev_end)
win = [r for r in rows if r["s"] >= t0]
span = (win[-1]["e"] - win[0]["s"]) / 1e6
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    n = r["Kernel_Name"]
    short = n.split("(")[0][:90]
    agg[short][0] += r["e"] - r["s"]; agg[short][1] += 1
busy = sum(v[0] for v in agg.values()) / 1e6
nsteps = len(g) - warm
print("window %.1f ms, kernel-busy %.1f ms, %d timed solver launches (%.1f ms / TTA step incl. eval share)" % (span, busy, nsteps, span / max(nsteps, 1)))
print("%-92s %8s %10s %9s %6s" % ("kernel", "calls", "total ms", "avg us", "%busy"))
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:top]:
    print("%-92s %8d %10.3f %9.1f %6.2f" % (k, c, t / 1e6, t / c / 1e3, 100 * t / 1e6 / busy))
