"""Everything under profiles/r1_* from the rocprofv3 outputs of tools/refresh_profiles.sh
(gpurun_out/prof_*): kernel stats, PMC rows of the snout:: kernels, r1_traffic.json."""
import csv, glob, json, os, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def short(name):
    return name.split("(")[0].replace("void ", "")


def stats(src, dst):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "collect_profiles.py"),
                    os.path.join(G, src), os.path.join(P, dst)], check=True)


def pmc(src):
    """-> {kernel: [values in KB per launch]} of the one counter collected in that pass"""
    f = sorted(glob.glob(os.path.join(G, src, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1]
    out = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "snout::" in r["Kernel_Name"]:
            out[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return out


stats("prof_bench", "r1_bench_kernel_stats.csv")
stats("prof_bench_sync", "r1_bench_sync_kernel_stats.csv")
stats("prof_wide", "r1_zigbee_wideband_kernel_stats.csv")
stats("prof_zbbig", "r1_zigbee_1e9_kernel_stats.csv")

fetch, write = pmc("prof_pmc_fetch"), pmc("prof_pmc_write")
with open(os.path.join(P, "r1_bench_pmc_snout_kernels.csv"), "w", newline="") as o:
    w = csv.writer(o)
    w.writerow(["Kernel", "Counter", "Launches", "Mean_KB_per_launch", "Min", "Max"])
    for k in sorted(set(fetch) | set(write)):
        for name, d in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            if d.get(k):
                v = d[k]
                w.writerow([k, name, len(v), sum(v) / len(v), min(v), max(v)])
k1 = [k for k in fetch if "btle_demod_corr" in k][0]
f_kb, w_kb = sum(fetch[k1]) / len(fetch[k1]), sum(write[k1]) / len(write[k1])
json.dump({
    "kernel": "btle_demod_corr",
    "rocprof_kernel_name": k1,
    "workload_samples": 1000000000,
    "FETCH_SIZE_KB": f_kb,
    "WRITE_SIZE_KB": w_kb,
    "correction": "gfx950: FETCH_SIZE reports 1/2 of wide coalesced streaming reads (MI355X_MICROARCH.md HBM) -> doubled; WRITE_SIZE exact",
    "traffic_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
    "source": "profiles/r1_bench_pmc_snout_kernels.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, python3 bench.py --steps 2 --warmup 1 --no-cpu)",
}, open(os.path.join(P, "r1_traffic.json"), "w"), indent=1)

if glob.glob(os.path.join(G, "prof_zb_pmc_fetch")):
    zf, zw = pmc("prof_zb_pmc_fetch"), pmc("prof_zb_pmc_write")
    with open(os.path.join(P, "r1_zigbee_1e9_pmc.csv"), "w", newline="") as o:
        w = csv.writer(o)
        w.writerow(["Kernel", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch"])
        tot = 0.0
        for k in sorted(set(zf) | set(zw)):
            a = sum(zf.get(k, [0])) / max(1, len(zf.get(k, [0])))
            b = sum(zw.get(k, [0])) / max(1, len(zw.get(k, [0])))
            w.writerow([k, a, b])
            tot += (2 * a + b) * 1024
    print("zigbee 1e9: %.2f B/sample of HBM traffic" % (tot / 1e9))
print("traffic: %.4f GB per %s launch" % ((2.0 * f_kb + w_kb) * 1024.0 / 1e9, k1))
