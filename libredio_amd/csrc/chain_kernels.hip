// chain_kernels.hip -- the north-star chain as ONE kernel: cf32 IQ -> K-tap FIR, keep every D-th
// output (dsputils::convolve semantics, src/dsputils/src/dsputils.rs:30-32) -> 1024-point forward
// transform of consecutive decimated blocks (kissfft::fft semantics, src/kissfft/src/kissfft.rs:18-31).
//
// Algorithmic HBM traffic: 8 B in + 8/D B out per input sample (9.6 B at D = 5); the decimated
// stream never leaves the CU.  A 256-thread workgroup produces four consecutive 1024-sample
// decimated blocks (all four waves share each FIR tile, R = 4 outputs per lane), parks them in LDS,
// then every wave transforms one block on its own (fft_wave.h), so the FFT phase needs no barrier.
#include "fir_core.h"
#include "fir_tile.h"
#include "fft_wave.h"
#include "redio_internal.h"

namespace redio {

template <int K, int D, bool FUSED>
__global__ __launch_bounds__(256) void chain_fir_fft1k_kernel(const float2 *__restrict__ x, long n_in,
                                                              const float *__restrict__ taps,
                                                              const float2 *__restrict__ tw,
                                                              float2 *__restrict__ out, long nblocks, int vec_ok)
{
    constexpr int R = 4, NT = 256, TILE_OUT = NT * R; // = 1024 = one transform
    using G = FirGeom<K, D, R>;
    constexpr int TILE_IN = G::tile_in(TILE_OUT);
    constexpr int XS = (G::lds_elems(TILE_OUT) + 1) & ~1; // keep the block buffers 16-B aligned
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *xs = reinterpret_cast<float2 *>(smem);
    float2 *yb = xs + XS; // 4 x FFT1K_LDS

    const int tid = threadIdx.x;
    const long blk0 = (long)blockIdx.x * 4;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
        const long blk = blk0 + t;
        if (blk >= nblocks) break; // workgroup-uniform
        load_tile<float2, G, NT, TILE_IN>(x, n_in, blk * (long)TILE_OUT * D, xs, vec_ok != 0);
        __syncthreads();
        float2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);
        fir_lane<float2, K, D, R, FUSED>(xs, tid, taps, acc);
        float4 *yt = reinterpret_cast<float4 *>(yb + t * FFT1K_LDS + tid * R);
        yt[0] = make_float4(acc[0].x, acc[0].y, acc[1].x, acc[1].y);
        yt[1] = make_float4(acc[2].x, acc[2].y, acc[3].x, acc[3].y);
        __syncthreads(); // xs free for the next tile; yb[t] complete
    }
    const int wave = tid >> 6, lane = tid & 63;
    const long blk = blk0 + wave;
    if (blk < nblocks) {
        float2 *mine = yb + wave * FFT1K_LDS;
        fft1k_wave<false>(mine, out + blk * 1024, mine, tw, lane);
    }
}

bool chain_supported(int K, long D, int nfft) { return nfft == 1024 && K == 127 && D == 5; }

template <int K, int D>
static hipError_t launch_chain_t(const FftPlanDev &p, const float2 *x, long n_in, const float *taps, float2 *out,
                                 long nblocks, bool fused, hipStream_t s)
{
    using G = FirGeom<K, D, 4>;
    constexpr int XS = (G::lds_elems(1024) + 1) & ~1;
    constexpr size_t LDS = (size_t)(XS + 4 * FFT1K_LDS) * sizeof(float2);
    static_assert(LDS <= 160 * 1024, "chain tile does not fit LDS");
    auto kf = chain_fir_fft1k_kernel<K, D, true>;
    auto ke = chain_fir_fft1k_kernel<K, D, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fused ? kf : ke),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((nblocks + 3) / 4);
    const int vec_ok = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    if (fused) hipLaunchKernelGGL(kf, dim3(grid), dim3(256), LDS, s, x, n_in, taps, p.tw, out, nblocks, vec_ok);
    else hipLaunchKernelGGL(ke, dim3(grid), dim3(256), LDS, s, x, n_in, taps, p.tw, out, nblocks, vec_ok);
    return hipGetLastError();
}

hipError_t launch_chain(const FftPlanDev &p, const float2 *x, long n_in, const float *taps, int K, long D,
                        float2 *out, long nblocks, bool fused, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    if (p.nfft == 1024 && !p.inverse && K == 127 && D == 5)
        return launch_chain_t<127, 5>(p, x, n_in, taps, out, nblocks, fused, s);
    return hipErrorNotSupported;
}

} // namespace redio
