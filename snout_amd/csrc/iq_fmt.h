// iq_fmt.h — input sample formats of the receive path and their device-side loads.
//
//   cf32 : interleaved float32 I,Q (numpy.complex64 / GNU Radio gr_complex), 8 B per sample
//   sc8  : interleaved int8  I,Q — what a HackRF delivers and what upstream `btle_rx` consumes
//          ("IQ = interleaved int8"), 2 B per sample
//   sc16 : interleaved int16 I,Q (USRP / `.sc16` captures), 4 B per sample
//
// Integer samples are DEFINED as the cf32 samples  v * 2^-7  (sc8) /  v * 2^-15  (sc16): both the
// conversion and the scale are exact in fp32, so a receive path on integer input is bit-identical
// to the cf32 path on the converted capture (which is how the oracle checks it).  Where the
// arithmetic that follows is invariant under an exact power-of-two scale (the sign of a cross
// product) the scale is skipped.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace snout {

constexpr int kFmtCf32 = 0, kFmtSc8 = 1, kFmtSc16 = 2;

__host__ __device__ constexpr uint32_t fmt_bytes(int fmt) { return fmt == kFmtSc8 ? 2u : (fmt == kFmtSc16 ? 4u : 8u); }
__host__ __device__ constexpr float fmt_scale(int fmt)
{
    return fmt == kFmtSc8 ? 0.0078125f : (fmt == kFmtSc16 ? 0.000030517578125f : 1.0f);
}

__device__ __forceinline__ float sc8_lo(uint32_t w, int byte) { return (float)(int)(int8_t)(w >> (8 * byte)); }
__device__ __forceinline__ float sc16_lo(uint32_t w, int half) { return (float)(int)(int16_t)(w >> (16 * half)); }

// one sample
template <int FMT>
__device__ __forceinline__ float2 iq_sample(const void* __restrict__ base, uint64_t i)
{
    if constexpr (FMT == kFmtSc8) {
        const uint16_t w = reinterpret_cast<const uint16_t*>(base)[i];
        return make_float2(sc8_lo(w, 0) * fmt_scale(FMT), sc8_lo(w, 1) * fmt_scale(FMT));
    } else if constexpr (FMT == kFmtSc16) {
        const uint32_t w = reinterpret_cast<const uint32_t*>(base)[i];
        return make_float2(sc16_lo(w, 0) * fmt_scale(FMT), sc16_lo(w, 1) * fmt_scale(FMT));
    } else {
        return reinterpret_cast<const float2*>(base)[i];
    }
}

// samples i, i+1 (i even) as (re0, im0, re1, im1)
template <int FMT>
__device__ __forceinline__ float4 iq_pair(const void* __restrict__ base, uint64_t i)
{
    constexpr float s = fmt_scale(FMT);
    if constexpr (FMT == kFmtSc8) {
        const uint32_t w = reinterpret_cast<const uint32_t*>(base)[i >> 1];
        return make_float4(sc8_lo(w, 0) * s, sc8_lo(w, 1) * s, sc8_lo(w, 2) * s, sc8_lo(w, 3) * s);
    } else if constexpr (FMT == kFmtSc16) {
        const uint2 w = reinterpret_cast<const uint2*>(base)[i >> 1];
        return make_float4(sc16_lo(w.x, 0) * s, sc16_lo(w.x, 1) * s, sc16_lo(w.y, 0) * s, sc16_lo(w.y, 1) * s);
    } else {
        return reinterpret_cast<const float4*>(base)[i >> 1];
    }
}

// The same pair kept raw (as loaded) and converted later: lets a kernel issue the load early and pay
// the conversion where the data is consumed.
template <int FMT> struct IqRaw { using pair = float4; };
template <> struct IqRaw<kFmtSc8> { using pair = uint32_t; };
template <> struct IqRaw<kFmtSc16> { using pair = uint2; };

template <int FMT>
__device__ __forceinline__ typename IqRaw<FMT>::pair iq_pair_raw(const void* __restrict__ base, uint64_t i)
{
    return reinterpret_cast<const typename IqRaw<FMT>::pair*>(base)[i >> 1];
}

// sample i alone in the low half of a raw pair, the other half zero
template <int FMT>
__device__ __forceinline__ typename IqRaw<FMT>::pair iq_single_raw(const void* __restrict__ base, uint64_t i)
{
    if constexpr (FMT == kFmtSc8) {
        return (uint32_t)reinterpret_cast<const uint16_t*>(base)[i];
    } else if constexpr (FMT == kFmtSc16) {
        return make_uint2(reinterpret_cast<const uint32_t*>(base)[i], 0u);
    } else {
        const float2 a = reinterpret_cast<const float2*>(base)[i];
        return make_float4(a.x, a.y, 0.0f, 0.0f);
    }
}

template <int FMT>
__device__ __forceinline__ float4 iq_pair_cvt(typename IqRaw<FMT>::pair w)
{
    constexpr float s = fmt_scale(FMT);
    if constexpr (FMT == kFmtSc8) {
        return make_float4(sc8_lo(w, 0) * s, sc8_lo(w, 1) * s, sc8_lo(w, 2) * s, sc8_lo(w, 3) * s);
    } else if constexpr (FMT == kFmtSc16) {
        return make_float4(sc16_lo(w.x, 0) * s, sc16_lo(w.x, 1) * s, sc16_lo(w.y, 0) * s, sc16_lo(w.y, 1) * s);
    } else {
        return w;
    }
}

// samples i .. i+3 (i a multiple of 4)
template <int FMT>
__device__ __forceinline__ void iq_quad(const void* __restrict__ base, uint64_t i, float2 out[4])
{
    constexpr float s = fmt_scale(FMT);
    if constexpr (FMT == kFmtSc8) {
        const uint2 w = reinterpret_cast<const uint2*>(base)[i >> 2];
        out[0] = make_float2(sc8_lo(w.x, 0) * s, sc8_lo(w.x, 1) * s);
        out[1] = make_float2(sc8_lo(w.x, 2) * s, sc8_lo(w.x, 3) * s);
        out[2] = make_float2(sc8_lo(w.y, 0) * s, sc8_lo(w.y, 1) * s);
        out[3] = make_float2(sc8_lo(w.y, 2) * s, sc8_lo(w.y, 3) * s);
    } else if constexpr (FMT == kFmtSc16) {
        const uint4 w = reinterpret_cast<const uint4*>(base)[i >> 2];
        out[0] = make_float2(sc16_lo(w.x, 0) * s, sc16_lo(w.x, 1) * s);
        out[1] = make_float2(sc16_lo(w.y, 0) * s, sc16_lo(w.y, 1) * s);
        out[2] = make_float2(sc16_lo(w.z, 0) * s, sc16_lo(w.z, 1) * s);
        out[3] = make_float2(sc16_lo(w.w, 0) * s, sc16_lo(w.w, 1) * s);
    } else {
        const float4 a = reinterpret_cast<const float4*>(base)[i >> 1];
        const float4 b = reinterpret_cast<const float4*>(base)[(i >> 1) + 1];
        out[0] = make_float2(a.x, a.y); out[1] = make_float2(a.z, a.w);
        out[2] = make_float2(b.x, b.y); out[3] = make_float2(b.z, b.w);
    }
}

}  // namespace snout
