import os, sys, subprocess, json
root = sys.argv[1]
for rep in range(2):
    for v in sys.argv[2:]:
        env = dict(os.environ, SNOUT_RX_LIB=os.path.join(root, "build", "variants", "libsnout_rx_%s.so" % v))
        out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "cfg4", "--steps", "10", "--warmup", "3", "--no-cpu", "--sync"], capture_output=True, env=env).stdout.decode().splitlines()[-1]
        d = json.loads(out)
        print(v, "cfg4 step %.3f ms kernel %.3f ms ok %d" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["config"]["decoded_crc_ok_per_gpu"]))
