"""Dev tool: the default bench line (every workload, no CPU legs) under two or more builds of the library
(SNOUT_RX_LIB), alternating, one row per workload: python tools/ab_bench.py <lib.so> <lib.so> ..."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for rep in range(2):
    for lib in sys.argv[1:]:
        env = dict(os.environ, SNOUT_RX_LIB=os.path.abspath(lib))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "10", "--warmup", "3", "--no-cpu"],
                             capture_output=True, env=env).stdout.decode().splitlines()[-1]
        d = json.loads(out)
        row = ["cfg3 %.3f/%.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"])]
        for k, v in d["other_workloads"].items():
            row.append("%s %.3f/%s" % (k, v["ms_per_step"], ("%.3f" % v["kernel_ms"]) if "kernel_ms" in v else "-"))
        print(os.path.basename(lib), " | ".join(row), flush=True)
