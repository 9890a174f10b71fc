// fft_wave.h -- the one-wavefront 1024-point transform (device only; shared by the batched FFT
// kernel and the fused FIR->FFT chain kernel).  Index maps and butterflies: fft_core.h.
#pragma once
#include "fft_core.h"

namespace redio {

// Orders this wave's LDS accesses as written.  LDS operations of one wavefront execute in issue
// order, so cross-lane exchange inside a wave needs no s_barrier -- only a compiler fence.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Stages m = 1, 4, 16, 64 of the 1024-point flow by the calling wavefront: v[t] holds x[lane + 64 t] on
// entry; on return v[4 q + j] holds leaf-block j (positions 256 j .. 256 j + 255) at index lane + 64 q,
// i.e. the four 256-point transforms of x[4 i + j] -- the input of the last stage.
template <bool INV, typename TwPtr>
__device__ __forceinline__ void fft1k_wave_stages0to3(float2 (&v)[16], float2 *ex, TwPtr tw, const Fft1kTw &t, int lane)
{
    fft1k_passA<INV>(v, tw);
    wave_lds_fence(); // every lane has read its inputs before anyone overwrites ex
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int k3 = 0; k3 < 4; ++k3) ex[fft1k_A_store(lane, k3, k4)] = v[k3 + 4 * k4];
    wave_lds_fence();
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = ex[fft1k_B_load(lane, e)];
    wave_lds_fence();
    fft1k_passB<INV>(v, t);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2)
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) ex[fft1k_B_store(lane, k1, k2)] = v[k1 + 4 * k2];
    wave_lds_fence();
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[4 * q + j] = ex[fft1k_C_load(lane, q, j)];
}

// Everything after the input load of a 1024-point transform by the calling wavefront: v[t] holds
// x[lane + 64 t] on entry.  ex: FFT1K_LDS float2 of LDS owned by this wave; dst: natural-order
// output (global memory, or LDS -- it may alias ex: LDS operations of a wave complete in order).
template <bool INV, typename DstPtr, typename TwPtr>
__device__ __forceinline__ void fft1k_wave_regs(float2 (&v)[16], DstPtr dst, float2 *ex, TwPtr tw, const Fft1kTw &t, int lane)
{
    fft1k_wave_stages0to3<INV>(v, ex, tw, t, lane);
    fft1k_passC<INV>(v, t);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[lane + 64 * q + 256 * j] = v[4 * q + j];
}

// One 1024-point transform by the calling wavefront.  src: natural-order input (global or LDS, may
// alias ex).
template <bool INV, typename SrcPtr>
__device__ __forceinline__ void fft1k_wave(SrcPtr src, float2 *dst, float2 *ex,
                                           const float2 *__restrict__ tw, int lane)
{
    float2 v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = src[lane + 64 * t];
    Fft1kTw t;
    fft1k_load_tw(t, lane, tw);
    fft1k_wave_regs<INV>(v, dst, ex, tw, t, lane);
}

} // namespace redio

namespace redio {

// The "native" 1024-point forward transform of fft_core.h by the calling wavefront: a[4 s + r] holds
// y[256 s + 4 lane + r] on entry (the FIR's own register layout), ex is FFT1KN_LDS float2 of LDS owned
// by this wave, dst the block's spectrum in global memory (natural order).
template <bool INV>
__device__ __forceinline__ void fft1kn_wave_tw(float2 (&a)[16], float2 *ex, const float2 *__restrict__ tw, const Fft1knTw12 &t12,
                                               const Fft1knTw34 &t34, float2 *dst, int ln)
{
    fft1kn_stage0<INV>(a, tw);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int d0 = 0; d0 < 4; ++d0) ex[fft1kn_x1_store(ln, k4, d0)] = a[4 * k4 + d0];
    wave_lds_fence();
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = ex[fft1kn_x1_load(ln, e)];
    wave_lds_fence();
    fft1kn_pass12<INV>(a, t12);
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) ex[fft1kn_x2_store(ln, k2, k3)] = a[k2 + 4 * k3];
    wave_lds_fence();
#pragma unroll
    for (int f = 0; f < 16; ++f) a[f] = ex[fft1kn_x2_load(ln, f)];
    wave_lds_fence();
    fft1kn_pass34<INV>(a, t34);
#pragma unroll
    for (int k0 = 0; k0 < 4; ++k0)
#pragma unroll
#if defined(REDIO_EXP_CHAIN_NT) && (REDIO_EXP_CHAIN_NT & 2) // the fused chain writes its spectra once (chain_v4.hip sets this before including the header)
        for (int k1 = 0; k1 < 4; ++k1) {
            typedef float nt_v2f __attribute__((ext_vector_type(2)));
            __builtin_nontemporal_store(nt_v2f{a[k1 + 4 * k0].x, a[k1 + 4 * k0].y}, reinterpret_cast<nt_v2f *>(dst + ln + 64 * k1 + 256 * k0));
        }
#else
        for (int k1 = 0; k1 < 4; ++k1) dst[ln + 64 * k1 + 256 * k0] = a[k1 + 4 * k0];
#endif
}

// the same with the 30 lane-dependent twiddles loaded here (callers that transform many blocks load them
// once and call fft1kn_wave_tw)
template <bool INV>
__device__ __forceinline__ void fft1kn_wave(float2 (&a)[16], float2 *ex, const float2 *__restrict__ tw, float2 *dst, int ln)
{
    Fft1knTw12 t12;
    Fft1knTw34 t34;
    fft1kn_load_tw12(t12, ln, tw);
    fft1kn_load_tw34(t34, ln, tw);
    fft1kn_wave_tw<INV>(a, ex, tw, t12, t34, dst, ln);
}

} // namespace redio
