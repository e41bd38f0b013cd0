"""Dev probe: device -> pinned-host copy of the first 96 bytes of 75 000 records of 160 bytes (hipMemcpy2DAsync) against the
contiguous 12 MB copy (hipMemcpyAsync), on the box's HIP runtime."""
import ctypes as C, time, torch
hip = C.CDLL("libamdhip64.so")
n, pitch, width = 75000, 160, 96
d = torch.zeros(n * pitch, dtype=torch.uint8, device="cuda")
h = torch.zeros(n * pitch, dtype=torch.uint8).pin_memory()
st = torch.cuda.Stream()
s = C.c_void_p(st.cuda_stream)
D2H = 2
def t(fn, k=20):
    for _ in range(3): fn()
    st.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    st.synchronize(); return (time.perf_counter() - t0) / k * 1e6
flat = lambda: hip.hipMemcpyAsync(C.c_void_p(h.data_ptr()), C.c_void_p(d.data_ptr()), C.c_size_t(n * pitch), D2H, s)
rect = lambda: hip.hipMemcpy2DAsync(C.c_void_p(h.data_ptr()), C.c_size_t(pitch), C.c_void_p(d.data_ptr()), C.c_size_t(pitch),
                                    C.c_size_t(width), C.c_size_t(n), D2H, s)
print(f"contiguous 12 MB: {t(flat):.0f} us; 2D 96 of 160 bytes x 75 000: {t(rect):.0f} us")
