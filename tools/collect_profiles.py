"""Copy the snout:: rows of a rocprofv3 kernel_stats.csv (under gpurun_out/) into profiles/."""
import csv, glob, os, sys

src_dir, dst = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(src_dir, "*", "*kernel_stats.csv")), key=os.path.getmtime)[-1]
rows = list(csv.DictReader(open(f)))
keep = [r for r in rows if "snout::" in r["Name"] or "rocclr" in r["Name"]]
with open(dst, "w", newline="") as o:
    wr = csv.DictWriter(o, fieldnames=rows[0].keys())
    wr.writeheader()
    for r in keep:
        r["Name"] = r["Name"].split("(")[0].replace("void ", "")
        wr.writerow(r)
print(f, "->", dst, len(keep), "rows")
