"""compact a rocprofv3 --stats kernel_stats.csv into a short, committable summary"""
import csv, sys
src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(src)))
with open(dst, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary of: {' '.join(sys.argv[3:])}\n")
    f.write(f"{'kernel':70s} {'calls':>7s} {'avg_us':>9s} {'min_us':>9s} {'max_us':>9s} {'pct':>7s}\n")
    for r in rows[:12]:
        name = r["Name"]
        name = name if len(name) <= 68 else name[:65] + "..."
        f.write(f"{name:70s} {int(r['Calls']):7d} {float(r['AverageNs'])/1e3:9.3f} {float(r['MinNs'])/1e3:9.3f} "
                f"{float(r['MaxNs'])/1e3:9.3f} {float(r['Percentage']):7.2f}\n")
print(open(dst).read())
