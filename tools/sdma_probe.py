import os, sys, time, torch
x = torch.empty(12_000_000, dtype=torch.uint8, device="cuda")
h = torch.empty(12_000_000, dtype=torch.uint8).pin_memory()
s = torch.cuda.Stream()
for _ in range(3):
    with torch.cuda.stream(s): h.copy_(x, non_blocking=True)
s.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    with torch.cuda.stream(s): h.copy_(x, non_blocking=True)
s.synchronize()
print("D2H 12 MB: %.1f us each" % ((time.perf_counter() - t0) / 20 * 1e6), {k: v for k, v in os.environ.items() if "SDMA" in k or "BLIT" in k})
