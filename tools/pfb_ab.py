"""Dev tool: A/B timing of libsnout_rx.so variants (tools/pfb_variants.sh) on the wideband paths.

    python tools/pfb_ab.py [--proto 0|1] [--samples 8e8] name1 name2 ...     (driver: one child per variant)

Each child loads build/variants/libsnout_rx_<name>.so through SNOUT_RX_LIB, checks the records of a
small capture bit for bit against the CPU oracle, then times the channelizer kernel (HIP events,
snout_rx_profile_history) and the whole step on a capture resident in HBM."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(proto, n_samples):
    import numpy as np, torch
    from snout_amd import synth
    from snout_amd.rx import SnoutRx
    from oracle import oracle_py
    M = 40 if proto == 0 else 16
    tile, truth = synth.wideband_capture(proto, M * (1 << (16 if proto == 0 else 17)), seed=3 + proto, sigma=0.0)
    rng = np.random.default_rng(5)
    small = (tile[:M * 30000] + 0.05 * (rng.standard_normal(M * 30000) + 1j * rng.standard_normal(M * 30000))).astype(np.complex64)
    rx = SnoutRx(proto=proto, n_channels=M)
    got = rx.process(torch.from_numpy(small.view(np.float32)).cuda())
    want = oracle_py.wideband_segment(small, proto=proto)
    same = len(got) == len(want) and all(np.array_equal(got[f], want[f]) for f in got.dtype.names)
    t = torch.from_numpy(tile.view(np.float32)).cuda()
    reps = max(1, int(n_samples) // tile.size)
    x = t.repeat(reps)
    torch.manual_seed(11)
    x += 0.05 * torch.randn_like(x)
    n = x.numel() // 2
    for _ in range(3):
        pk = rx.process(x, copy=False)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); pk = rx.process(x, copy=False); ts.append(time.perf_counter() - t0)
    k = rx.profile_history()[-8:]
    print(f"parity={'OK' if same else 'FAIL'} ({len(got)} recs) n={n:.3g} kernel avg {k.mean():.3f} min {k.min():.3f} ms "
          f"step min {min(ts)*1e3:.3f} ms pkts={len(pk)} ok={int(pk['crc_ok'].sum())}", flush=True)


if __name__ == "__main__":
    args = sys.argv[1:]
    proto, ns = 0, 8e8
    if "--child" in args:
        child(int(args[args.index("--proto") + 1]), float(args[args.index("--samples") + 1]))
        sys.exit(0)
    if "--proto" in args:
        i = args.index("--proto"); proto = int(args[i + 1]); del args[i:i + 2]
    if "--samples" in args:
        i = args.index("--samples"); ns = float(args[i + 1]); del args[i:i + 2]
    for name in args:
        blocks = None
        if "@" in name:                                     # name@1024: persistent grid of the channelizer
            name, blocks = name.split("@")
        lib = os.path.join(ROOT, "build", "variants", f"libsnout_rx_{name}.so")
        env = dict(os.environ, SNOUT_RX_LIB=lib)
        if blocks:
            env["SNOUT_PFB_BLOCKS"] = blocks
            name = f"{name}@{blocks}"
        r = subprocess.run([sys.executable, __file__, "--child", "--proto", str(proto), "--samples", str(ns)],
                           env=env, capture_output=True, text=True, timeout=900)
        out = (r.stdout.strip().splitlines() or ["(no output)"])[-1]
        print(f"{name:24s} {out}" + ("" if r.returncode == 0 else f"  rc={r.returncode} {r.stderr[-400:]}"), flush=True)
