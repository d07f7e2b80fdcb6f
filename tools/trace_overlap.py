"""From a rocprofv3 --kernel-trace CSV: for every launch of kernel A (substring), which launches of other kernels ran while it ran, and how
their durations compare with the same kernels' launches outside any A window.  Usage: trace_overlap.py <dir> <A-substring>"""
import csv, glob, sys
from collections import defaultdict

d, a = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
rows.sort(key=lambda r: r[1])
A = [(s, e) for n, s, e in rows if a in n]
print(f"{len(A)} launches of *{a}*, mean {sum(e - s for s, e in A) / max(len(A), 1) / 1e3:.1f} us")
inside, outside = defaultdict(list), defaultdict(list)
j = 0
for n, s, e in rows:
    if a in n:
        continue
    while j < len(A) and A[j][1] < s:
        j += 1
    k = n.split("(")[0].split("<")[0][-40:]
    over = any(s < ae and e > as_ for as_, ae in A[max(0, j - 1):j + 1])
    (inside if over else outside)[k].append((e - s) / 1e3)
print(f"{'kernel':42} {'n under A':>9} {'us under A':>11} {'n alone':>8} {'us alone':>9}")
for k in sorted(outside, key=lambda k: -sum(outside[k]))[:8]:
    i, o = inside.get(k, []), outside[k]
    print(f"{k:42} {len(i):9d} {sum(i) / max(len(i), 1):11.2f} {len(o):8d} {sum(o) / len(o):9.2f}")
