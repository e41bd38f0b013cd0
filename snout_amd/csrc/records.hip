// records.hip — packing of decoded records for the multi-GPU gather: one launch per
// collected segment copies the device copy of its records into the exchange buffer in the wire format
// (the first `width` bytes of each 160-byte snout_pkt: BTLE records use 24 + <= 42 bytes), marking the
// records a segment found in its pre-roll as disowned.  Replaces five torch operations per segment.
#include "common.h"
#include <rocprim/rocprim.hpp>

namespace snout {

__global__ __launch_bounds__(256) void pack_records(const uint4* __restrict__ src, uint64_t n, uint4* __restrict__ dst,
                                                    uint32_t w16, uint64_t own_from, unsigned long long* __restrict__ longest)
{
    __shared__ uint32_t mx;
    if (longest) { if (threadIdx.x == 0) mx = 0u; __syncthreads(); }
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;       // one 16-byte piece
    if (i < n * w16) {
        const uint64_t rec = i / w16;
        const uint32_t piece = (uint32_t)(i % w16);
        uint4 v = src[rec * (sizeof(snout_pkt) / 16u) + piece];
        if (piece == 0u) {
            if (own_from) {
                const uint64_t si = (uint64_t)v.x | ((uint64_t)v.y << 32);
                if (si < own_from) { v.x = 0u; v.y = 1u << 30; }             // sample_index = 2^62
            }
            if (longest) atomicMax(&mx, v.w >> 16);                          // snout_pkt.len
        }
        dst[i] = v;
    }
    if (longest) {
        __syncthreads();
        if (threadIdx.x == 0 && mx) atomicMax(longest, (unsigned long long)mx);
    }
}

int launch_pack_records(const snout_pkt* src, uint64_t n, void* dst, uint32_t width, uint64_t own_from, void* longest_dev, hipStream_t st)
{
    const uint32_t w16 = width / 16u;
    const uint64_t pieces = n * w16;
    hipLaunchKernelGGL(pack_records, dim3((uint32_t)((pieces + 255u) / 256u)), dim3(256), 0, st,
                       reinterpret_cast<const uint4*>(src), n, reinterpret_cast<uint4*>(dst), w16, own_from,
                       reinterpret_cast<unsigned long long*>(longest_dev));
    SNOUT_HIP(hipGetLastError());
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Rank 0 of the gather: fixed-shape sort + duplicate removal of the gathered wire records on the device (what
// snout_amd/dist.py::dedup_records states on the host).  Three kernels of ours around rocPRIM's radix sort and scan --
// the torch formulation of the same rule was ~65 launches per exchange from the Python thread that also feeds the scans.
// ---------------------------------------------------------------------------------------------
constexpr unsigned long long kDedupNoKey = ~0ull >> 2;          // above every valid key (proto < 4), 62 bits

__global__ __launch_bounds__(256) void dedup_keys(const unsigned char* __restrict__ rows, uint32_t width, uint64_t cap, uint64_t R,
                                                  const long long* __restrict__ counts, uint32_t counts_stride,
                                                  unsigned long long* __restrict__ keys, uint32_t* __restrict__ idx)
{
    const uint64_t r = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (r >= R) return;
    const uint64_t b = r / cap, slot = r - b * cap;
    long long cnt = counts[b * counts_stride];
    if (cnt > (long long)cap) cnt = (long long)cap;
    const uint4 h = *reinterpret_cast<const uint4*>(rows + r * width);      // sample_index | proto | channel, len
    const unsigned long long si = (unsigned long long)h.x | ((unsigned long long)h.y << 32);
    const bool valid = (long long)slot < cnt && si < (1ull << 62);
    keys[r] = valid ? ((unsigned long long)(h.z & 0xFu) << 60) | ((unsigned long long)(h.w & 0xFFFFu) << 44) | (si & ((1ull << 44) - 1ull))
                    : kDedupNoKey;
    idx[r] = (uint32_t)r;
}

__global__ __launch_bounds__(256) void dedup_flags(const unsigned char* __restrict__ rows, uint32_t width, uint64_t R, uint32_t tol,
                                                   const unsigned long long* __restrict__ sk, const uint32_t* __restrict__ sidx,
                                                   uint32_t* __restrict__ keep)
{
    const uint64_t p = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (p >= R) return;
    const unsigned long long k = sk[p];
    bool kp = k != kDedupNoKey;
    if (kp && p > 0) {
        const unsigned long long kb = sk[p - 1];
        bool dup = (k >> 44) == (kb >> 44) && (k - kb) <= (unsigned long long)tol;
        if (dup && tol > 0u) {
            // the same frame only if length and bytes agree too (each run may first recognise another preamble symbol)
            const unsigned char* a = rows + (uint64_t)sidx[p] * width;
            const unsigned char* b = rows + (uint64_t)sidx[p - 1] * width;
            dup = *reinterpret_cast<const unsigned long long*>(a + 8) == *reinterpret_cast<const unsigned long long*>(b + 8);
            for (uint32_t o = 24; dup && o < width; o += 8)
                dup = *reinterpret_cast<const unsigned long long*>(a + o) == *reinterpret_cast<const unsigned long long*>(b + o);
        }
        kp = !dup;
    }
    keep[p] = kp ? 1u : 0u;
}

__global__ __launch_bounds__(256) void dedup_scatter(const uint4* __restrict__ rows, uint32_t w16, uint64_t R, const uint32_t* __restrict__ sidx,
                                                     const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos,
                                                     uint4* __restrict__ out, unsigned long long* __restrict__ n_keep)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;       // one 16-byte piece of sorted position p
    if (i >= R * w16) return;
    const uint64_t p = i / w16;
    const uint32_t piece = (uint32_t)(i - p * w16);
    if (keep[p]) out[(uint64_t)pos[p] * w16 + piece] = rows[(uint64_t)sidx[p] * w16 + piece];
    if (i == R * w16 - 1u) *n_keep = (unsigned long long)pos[R - 1] + keep[R - 1];
}

struct DedupPlan { size_t keys_in, keys_out, idx_in, idx_out, keep, pos, tmp, tmp_bytes, total; };

static int dedup_plan(uint64_t R, DedupPlan* pl)
{
    size_t sort_b = 0, scan_b = 0;
    SNOUT_HIP(rocprim::radix_sort_pairs(nullptr, sort_b, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                        (uint32_t*)nullptr, (size_t)R, 0u, 62u, (hipStream_t) nullptr));
    SNOUT_HIP(rocprim::exclusive_scan(nullptr, scan_b, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, (size_t)R, rocprim::plus<uint32_t>(),
                                      (hipStream_t) nullptr));
    auto up = [](size_t v) { return (v + 255u) & ~(size_t)255u; };
    size_t o = 0;
    pl->keys_in = o; o += up(R * 8u);
    pl->keys_out = o; o += up(R * 8u);
    pl->idx_in = o; o += up(R * 4u);
    pl->idx_out = o; o += up(R * 4u);
    pl->keep = o; o += up(R * 4u);
    pl->pos = o; o += up(R * 4u);
    pl->tmp = o; pl->tmp_bytes = std::max(sort_b, scan_b); o += up(pl->tmp_bytes);
    pl->total = o;
    return 0;
}

}  // namespace snout

using namespace snout;

extern "C" {

size_t snout_records_dedup_workspace(uint32_t blocks, uint64_t cap)
{
    DedupPlan pl{};
    if (blocks == 0 || cap == 0 || dedup_plan((uint64_t)blocks * cap, &pl)) return 0;
    return pl.total;
}

int snout_records_dedup(const void* rows_dev, uint32_t width, uint32_t blocks, uint64_t cap, const int64_t* counts_dev,
                        uint32_t counts_stride, uint32_t tol, void* out_dev, uint64_t* n_keep_dev, void* work_dev,
                        size_t work_bytes, void* hip_stream)
{
    if (!rows_dev || !counts_dev || !out_dev || !n_keep_dev || !work_dev || blocks == 0 || cap == 0 || width < 32u ||
        width > sizeof(snout_pkt) || (width & 15u) || counts_stride == 0)
        return SNOUT_EINVAL;
    const uint64_t R = (uint64_t)blocks * cap;
    if (R >= (1ull << 32)) { set_last_error("%llu record slots: at most 2^32 - 1", (unsigned long long)R); return SNOUT_ERANGE; }
    DedupPlan pl{};
    if (int rc = dedup_plan(R, &pl)) return rc;
    if (work_bytes < pl.total) { set_last_error("dedup workspace of %zu bytes, %zu needed", work_bytes, pl.total); return SNOUT_EINVAL; }
    hipStream_t st = (hipStream_t)hip_stream;
    unsigned char* w = reinterpret_cast<unsigned char*>(work_dev);
    auto* keys_in = reinterpret_cast<unsigned long long*>(w + pl.keys_in);
    auto* keys_out = reinterpret_cast<unsigned long long*>(w + pl.keys_out);
    auto* idx_in = reinterpret_cast<uint32_t*>(w + pl.idx_in);
    auto* idx_out = reinterpret_cast<uint32_t*>(w + pl.idx_out);
    auto* keep = reinterpret_cast<uint32_t*>(w + pl.keep);
    auto* pos = reinterpret_cast<uint32_t*>(w + pl.pos);
    const uint32_t g = (uint32_t)((R + 255u) / 256u);
    hipLaunchKernelGGL(dedup_keys, dim3(g), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(rows_dev), width, cap, R,
                       reinterpret_cast<const long long*>(counts_dev), counts_stride, keys_in, idx_in);
    size_t tb = pl.tmp_bytes;
    SNOUT_HIP(rocprim::radix_sort_pairs(w + pl.tmp, tb, keys_in, keys_out, idx_in, idx_out, (size_t)R, 0u, 62u, st));
    hipLaunchKernelGGL(dedup_flags, dim3(g), dim3(256), 0, st, reinterpret_cast<const unsigned char*>(rows_dev), width, R, tol,
                       keys_out, idx_out, keep);
    tb = pl.tmp_bytes;
    SNOUT_HIP(rocprim::exclusive_scan(w + pl.tmp, tb, keep, pos, 0u, (size_t)R, rocprim::plus<uint32_t>(), st));
    const uint32_t w16 = width / 16u;
    hipLaunchKernelGGL(dedup_scatter, dim3((uint32_t)((R * w16 + 255u) / 256u)), dim3(256), 0, st, reinterpret_cast<const uint4*>(rows_dev),
                       w16, R, idx_out, keep, pos, reinterpret_cast<uint4*>(out_dev), reinterpret_cast<unsigned long long*>(n_keep_dev));
    SNOUT_HIP(hipGetLastError());
    return SNOUT_OK;
}

}  // extern "C"
