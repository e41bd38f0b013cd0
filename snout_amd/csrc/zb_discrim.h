// zb_discrim.h — the FM discriminator arithmetic of the 802.15.4 path, shared by zb_discrim
// (zigbee.hip) and the fused M = 16 channelizer epilogue (pfb.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace snout {

// The 257-entry atan table as the kernels hold it in LDS: plain (two 4-byte reads per value: tab[k], tab[k + 1]) or as 256
// pairs (tab[k], tab[k + 1]) read with ONE 8-byte read -- half the LDS instructions and half the bank-conflict cycles of the
// lane-random index (SNOUT_ATAN_PAIR in pfb_spec.hip).
struct AtanPlain {
    const float* t;
    __device__ __forceinline__ void get(int k, float& t0, float& t1) const { t0 = t[k]; t1 = t[k + 1]; }
};
struct AtanPairs {
    const float2* p;
    __device__ __forceinline__ void get(int k, float& t0, float& t1) const { const float2 v = p[k]; t0 = v.x; t1 = v.y; }
};

// a4: d[t] = fast_atan2f(Im(x[t] conj x[t-1]), Re(...)),  x[-1] = 0.  GNU Radio's table-driven
// fast_atan2f (257-entry atan table, linear interpolation, octant fix-up).
template <class Tab>
__device__ __forceinline__ float fast_atan2f_with(float y, float x, const Tab tab)
{
    // Straight-line form of the oracle's branches (same operations on the taken path, selected).
    const float ya = fabsf(y), xa = fabsf(x);
    const bool lt = ya < xa;
    const float z = (lt ? ya : xa) / (lt ? xa : ya);         // 0/0 -> NaN, see the end
    const float a = z * 255.0f;
    // the oracle's "& 0xff" is the identity here: z is a minimum over a maximum, in [0, 1] or NaN, so a is in
    // [0, 255] or NaN and v_cvt_i32_f32 gives 0..255 (NaN -> 0)
    const int k = (int)a;
    float t0, t1;
    tab.get(k, t0, t1);
    // (t1 - t0 kept apart from a - k: the compiler otherwise packs the two subtractions into one v_pk_add_f32 that costs two
    //  register moves to set up)
    float dt;
    asm("v_sub_f32 %0, %1, %2" : "=v"(dt) : "v"(t1), "v"(t0));
    const float interp = t0 + dt * (a - (float)k);
    const float base = z < 0.003921569f ? z : interp;
    const float PI = 3.14159265358979323846f, H = 1.57079632679489661923f;
    const bool xp = x >= 0.0f, yp = y >= 0.0f;
    // The oracle's eight branch values are +-m with m = base | PI - base (|y| < |x|) or
    // H - base | H + base (otherwise): base - PI == -(PI - base), -H + base == -(H - base) and
    // -H - base == -(H + base) exactly (rounding is symmetric), so one magnitude and the sign of
    // "y >= 0" (a compare, not the sign bit: -0.0 counts as non-negative) give the same bits.
    // The magnitude itself is c + s base with c = 0 | PI | H and s = -1 where exactly one of
    // (|y| < |x|), (x >= 0) holds: a - b and a + (-b) round alike and 0 + base is base.
    const float c = lt ? (xp ? 0.0f : PI) : H;
    const float sb = __uint_as_float(__float_as_uint(base) ^ ((lt != xp) ? 0x80000000u : 0u));
    const float m = c + sb;
    // x == y == 0 gives z = 0/0 = NaN and a NaN angle here: the caller's finiteness test turns it into
    // the 0 the oracle returns for that case
    return __uint_as_float(__float_as_uint(m) ^ (yp ? 0u : 0x80000000u));
}


// d = fast_atan2f(Im(a conj p), Re(a conj p)) with the products rounded first (contraction is off);
// a non-finite result is defined as 0 (as the oracle).
template <class Tab>
__device__ __forceinline__ float zb_discriminate_with(float2 a, float2 p, const Tab tab)
{
    const float re = a.x * p.x + a.y * p.y;
    const float im = a.y * p.x - a.x * p.y;
    float v = fast_atan2f_with(im, re, tab);
    if (!(fabsf(v) <= 4.0f)) v = 0.0f;
    return v;
}

__device__ __forceinline__ float fast_atan2f_tab(float y, float x, const float* __restrict__ tab)
{
    return fast_atan2f_with(y, x, AtanPlain{tab});
}
__device__ __forceinline__ float zb_discriminate(float2 a, float2 p, const float* __restrict__ tab)
{
    return zb_discriminate_with(a, p, AtanPlain{tab});
}

}  // namespace snout
