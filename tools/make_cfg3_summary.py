"""Condense a rocprofv3 --kernel-trace --stats run of tools/bench_core.py at BASELINE cfg-3 (8 graphs x 256 nodes) into the
text summary committed under profiles/: usage  make_cfg3_summary.py <kernel_stats.csv> <bench_core stdout> <out.txt>"""
import csv, sys
stats, log, out = sys.argv[1], sys.argv[2], sys.argv[3]
rows = list(csv.DictReader(open(stats)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python tools/bench_core.py rand256x256x256x256x256x256x256x256 5   (MI355X, gfx950)\n")
    f.write("# BASELINE cfg-3: MGM3_unsup forward+backward on 8 graphs x 256 nodes (M = 2048), fp32; 13 forwards + 8 backwards in the run\n")
    for line in open(log):
        if line.startswith(("sizes", "fwd")):
            f.write("# " + line)
    f.write("%-58s %6s %12s %10s %6s\n" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for r in rows:
        t = float(r["TotalDurationNs"])
        if t / tot < 0.002:
            continue
        f.write("%-58s %6s %12.0f %10.1f %6.1f\n" % (r["Name"].split("(")[0][:58], r["Calls"], t / 1e3, float(r["AverageNs"]) / 1e3, 100 * t / tot))
    f.write("# total kernel time %.2f ms\n" % (tot / 1e6))
