"""Dev tool: idle gaps of a rocprofv3 kernel trace (csv): the largest gaps, the kernel before and behind each, and the
distribution of gap lengths.   python tools/gap_report.py <kernel_trace.csv> [first fraction to skip, default 0.5]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("snout::", "").replace("void ", "")[:26]) for r in rows)
ev = ev[int(len(ev) * skip):]
t_end, last = ev[0][1], ev[0][2]
gaps = []
for s, e, n in ev[1:]:
    if s > t_end:
        gaps.append((s - t_end, last, n))
    if e > t_end:
        t_end, last = e, n
span = t_end - ev[0][0]
tot = sum(g[0] for g in gaps)
print(f"window {span / 1e6:.2f} ms, idle {tot / 1e6:.2f} ms = {100 * tot / span:.1f} % in {len(gaps)} gaps")
hist = collections.Counter(min(int(g[0] / 1e3) // 10 * 10, 200) for g in gaps)
print("gap length (us, bucket of 10) -> count, idle ms:", {k: (hist[k], round(sum(g[0] for g in gaps if min(int(g[0] / 1e3) // 10 * 10, 200) == k) / 1e6, 2)) for k in sorted(hist)})
by = collections.Counter()
for g in gaps:
    by[(g[1], g[2])] += g[0]
print("idle by (kernel before -> kernel behind), ms:")
for (a, b), v in by.most_common(12):
    print(f"  {a:26s} -> {b:26s} {v / 1e6:6.2f}")
