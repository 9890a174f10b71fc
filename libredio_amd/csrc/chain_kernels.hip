// chain_kernels.hip -- dispatch of the north-star chain: cf32 IQ -> K-tap FIR, keep every D-th output
// (dsputils::convolve semantics, src/dsputils/src/dsputils.rs:30-32) -> 1024-point forward transform of
// consecutive decimated blocks (kissfft::fft semantics, src/kissfft/src/kissfft.rs:18-31).
//
// One kernel (chain_v4.hip) for the shapes listed in chain_supported() on a 16-byte aligned stream: 8 B in +
// 8/D B out per input sample, the decimated stream never leaves the CU.  Everything else (and an 8-byte
// aligned stream) returns hipErrorNotSupported and the C-ABI layer runs the FIR and FFT kernels back to back
// through a plan-owned intermediate: same bits.  The earlier kernel generations (workgroup per tile, wave per
// block without the carried halo, dynamic chunk queue) are in the repository history, not in the library.
#include "redio_internal.h"
#include <stdio.h>

namespace redio {

hipError_t launch_chain_v4(const float2 *x, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused,
                           hipStream_t s, unsigned long long *dbg, long dbg_cap); // chain_v4.hip
hipError_t launch_chain_v4_u8(const void *bytes, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused, hipStream_t s);
hipError_t launch_chain_v4_shape_u8(int K, int D, const void *bytes, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused,
                                    hipStream_t s);
hipError_t launch_chain_v4_shape(int K, int D, const float2 *x, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused,
                                 hipStream_t s); // chain_v4.hip

bool chain_supported(int K, long D, int nfft)
{
    if (nfft != 1024) return false;
    return (K == 127 && (D == 5 || D == 1 || D == 3)) || (K == 63 && (D == 5 || D == 1));
}

hipError_t launch_chain(const FftPlanDev &p, const float2 *x, long n_in, const float *taps, int K, long D,
                        float2 *out, long nblocks, bool fused, hipStream_t s, unsigned long long *dbg, long dbg_cap)
{
    (void)n_in;
    if (nblocks <= 0) return hipSuccess;
    if (p.nfft != 1024 || p.inverse || !chain_supported(K, D, p.nfft)) return hipErrorNotSupported;
    if ((reinterpret_cast<uintptr_t>(x) & 15) != 0) return hipErrorNotSupported; // every sub-tile starts on an even sample
    if (K == 127 && D == 5) return launch_chain_v4(x, taps, p.tw, out, nblocks, fused, s, dbg, dbg_cap);
    return launch_chain_v4_shape(K, (int)D, x, taps, p.tw, out, nblocks, fused, s);
}

// chain_v4_kernel<K, D, FUSED, WPS = 2, CH = 8, FIR_ONLY = false, TWP = true, IN_U8 = false> (launch_chain_v4 / launch_chain_v4_shape)
const char *chain_kernel_name(int K, long D, bool fused_math, char *buf, size_t cap)
{
    if (!chain_supported(K, D, 1024)) return nullptr;
    snprintf(buf, cap, "chain_v4_kernel<%d,%ld,%s,2,8,false,true,false,false>", K, D, fused_math ? "true" : "false");
    return buf;
}

// u8 I/Q input: the shapes with a one-kernel form are those of the cf32 chain
hipError_t launch_chain_u8(const FftPlanDev &p, const void *bytes, const float *taps, int K, long D, float2 *out, long nblocks, bool fused,
                           hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    if (p.nfft != 1024 || p.inverse || !chain_supported(K, D, p.nfft)) return hipErrorNotSupported;
    if ((reinterpret_cast<uintptr_t>(bytes) & 3) != 0) return hipErrorNotSupported; // every sub-tile starts on an even sample
    if (K == 127 && D == 5) return launch_chain_v4_u8(bytes, taps, p.tw, out, nblocks, fused, s);
    return launch_chain_v4_shape_u8(K, (int)D, bytes, taps, p.tw, out, nblocks, fused, s);
}

} // namespace redio
