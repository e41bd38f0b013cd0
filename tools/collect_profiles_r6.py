"""Everything under profiles/r6_* from the rocprofv3 outputs of tools/refresh_profiles_r6.sh
(gpurun_out/r6prof/*): kernel stats of every workload, HBM traffic of the dominant kernels
(r6_traffic.json, read by bench.py), SQ counters of the channelizer."""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out", "r6prof")
P = os.path.join(ROOT, "profiles")


def short(name):
    return name.split("(")[0].replace("void ", "")


def stats(src, dst):
    fs = sorted(glob.glob(os.path.join(G, src, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        print("missing", src)
        return
    # only traces of the steady state are collected (tools/check_trace.py ran on the box, before the
    # per-dispatch trace was deleted: the dominant kernels' max within 10 x their median)
    st = os.path.join(G, src, "steady.txt")
    if not os.path.exists(st) or not open(st).read().startswith("STEADY"):
        print("REFUSED", src, "(no steady.txt, or NOT_STEADY: some kernel's max exceeds 10 x its median)")
        return
    rows = list(csv.DictReader(open(fs[-1])))
    keep = [r for r in rows if "snout::" in r["Name"] or "rocclr" in r["Name"]]
    with open(os.path.join(P, dst), "w", newline="") as o:
        wr = csv.DictWriter(o, fieldnames=rows[0].keys())
        wr.writeheader()
        for r in keep:
            r["Name"] = short(r["Name"])
            wr.writerow(r)
    print(src, "->", dst, len(keep), "rows")


def pmc(src):
    """-> {kernel: {counter: [values per launch]}}"""
    fs = sorted(glob.glob(os.path.join(G, src, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    out = defaultdict(lambda: defaultdict(list))
    if not fs:
        return out
    for r in csv.DictReader(open(fs[-1])):
        if "snout::" in r["Kernel_Name"]:
            out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


for src, dst in (("bench", "r6_bench_kernel_stats.csv"), ("bench_sync", "r6_bench_sync_kernel_stats.csv"),
                 ("cfg2", "r6_cfg2_kernel_stats.csv"), ("cfg4", "r6_cfg4_kernel_stats.csv"),
                 ("zigbee1", "r6_zigbee_1e9_kernel_stats.csv"), ("cfg5", "r6_cfg5_kernel_stats.csv"), ("cfg4_pipe", "r6_cfg4_pipelined_kernel_stats.csv")):
    stats(src, dst)

# HBM traffic per launch: (2 FETCH_SIZE + WRITE_SIZE) KB (gfx950: FETCH_SIZE reports half of wide streaming reads)
SIZES = {"bench": ("cfg3", 800000000), "cfg2": ("cfg2", 1000000000), "cfg4": ("cfg4", 320000000),
         "zigbee1": ("zigbee1", 1000000000)}
DOM = {"cfg3": ("pfb_spec40", "pfb_spec<40, 1, 0, 16>"), "cfg2": ("btle_demod_corr", "btle_demod_corr<1, 0>"),
       "cfg4": ("pfb_spec16", "pfb_spec<16, 2, 0, 16>"), "zigbee1": ("zb_discrim..zb_walk", None)}
traffic = []
with open(os.path.join(P, "r6_pmc_hbm_bytes.csv"), "w", newline="") as o:
    w = csv.writer(o)
    w.writerow(["Workload", "Kernel", "Launches", "FETCH_SIZE_KB_per_launch", "WRITE_SIZE_KB_per_launch", "HBM_bytes_per_launch"])
    for src, (wl, n) in SIZES.items():
        f = pmc("pmc_fetch" if src == "bench" else src + "_fetch")
        wr = pmc("pmc_write" if src == "bench" else src + "_write")
        tot = 0.0
        for k in sorted(set(f) | set(wr)):
            a = f[k].get("FETCH_SIZE", [0.0]); b = wr[k].get("WRITE_SIZE", [0.0])
            fa, wb = sum(a) / max(1, len(a)), sum(b) / max(1, len(b))
            by = (2.0 * fa + wb) * 1024.0
            w.writerow([wl, k, len(a), fa, wb, by])
            name, full = DOM[wl]
            if full and k.endswith(full):
                traffic.append({"workload": wl, "kernel": name, "rocprof_kernel_name": k, "workload_samples": n,
                                "FETCH_SIZE_KB": fa, "WRITE_SIZE_KB": wb, "traffic_bytes_per_launch": by})
            if wl == "zigbee1" and k.split("::")[-1].startswith("zb_"):
                tot += by
        if wl == "zigbee1" and tot:
            traffic.append({"workload": wl, "kernel": "zb_discrim..zb_walk", "workload_samples": n,
                            "traffic_bytes_per_launch": tot, "note": "sum over the chain's kernels, per segment"})
for e in traffic:
    e["correction"] = "gfx950: FETCH_SIZE reports 1/2 of wide coalesced streaming reads (MI355X_MICROARCH.md HBM) -> doubled; WRITE_SIZE exact"
    e["source"] = "profiles/r6_pmc_hbm_bytes.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, tools/refresh_profiles_r6.sh)"
json.dump(traffic, open(os.path.join(P, "r6_traffic.json"), "w"), indent=1)
for e in traffic:
    print("%-8s %-24s %.3f GB per launch = %.2f B/sample" % (e["workload"], e["kernel"], e["traffic_bytes_per_launch"] / 1e9,
                                                             e["traffic_bytes_per_launch"] / e["workload_samples"]))

# SQ counters of the channelizer
with open(os.path.join(P, "r6_pfb40_sq_counters.csv"), "w", newline="") as o:
    w = csv.writer(o)
    w.writerow(["Kernel", "Counter", "Launches", "Mean_per_launch"])
    for tag in "abc":
        for k, d in pmc("sq_" + tag).items():
            if "pfb_" in k:
                for c, v in sorted(d.items()):
                    w.writerow([k, c, len(v), sum(v) / len(v)])
import shutil
for src, dst in (("cfg5_timeline.txt", "r6_cfg5_timeline.txt"), ("cfg4_timeline.txt", "r6_cfg4_timeline.txt"), ("fake_world.txt", "r6_fake_world.txt"),
                 ("bench_timeline.txt", "r6_bench_timeline.txt"), ("spec40_stamps.txt", "r6_spec40_stamps.txt"), ("cfg5/steady.txt", "r6_cfg5_steady.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
        print(src, "->", dst)
for f in ("bench.log", "bench_sync.log", "bench_plain.log"):
    p = os.path.join(G, f)
    if os.path.exists(p):
        lines = [ln for ln in open(p) if ln.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            print(f, "value %.0f" % d["value"], "ms/step %.3f" % d["ms_per_step"], "kernel_ms %.3f" % d["roofline"]["kernel_ms"],
                  "frac %.4f" % d["roofline"]["frac"])
