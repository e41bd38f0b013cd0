"""Dev tool: cfg #4 (one segment at a time) under variants of the library built by tools/pfb_variants.sh:
python tools/ab_cfg4.py <repo root> name[@blocks] ...   (blocks -> SNOUT_PFB_BLOCKS)"""
import json, os, subprocess, sys
root = sys.argv[1]
for rep in range(2):
    for spec in sys.argv[2:]:
        v, _, blocks = spec.partition("@")
        env = dict(os.environ, SNOUT_RX_LIB=os.path.join(root, "build", "variants", "libsnout_rx_%s.so" % v))
        if blocks:
            env["SNOUT_PFB_BLOCKS"] = blocks
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "cfg4", "--steps", "10", "--warmup", "3", "--no-cpu", "--sync"],
                             capture_output=True, env=env).stdout.decode().splitlines()[-1]
        d = json.loads(out)
        print("%-12s cfg4 step %.3f ms kernel %.3f ms ok %d" % (spec, d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["decoded_crc_ok_per_gpu"]), flush=True)
