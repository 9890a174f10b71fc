// fir_kernels.hip -- gfx950 kernels for dsputils::convolve (src/dsputils/src/dsputils.rs:30-32)
// and its decimating / complex-input extensions.
//
// Bound: HBM for D>=1 at the north-star sizes, with a VALU floor close behind (127 taps / 5 on cf32
// is 50.8 FMA per input sample).  Design: one 256-thread workgroup stages a contiguous input tile
// in LDS with coalesced 16-byte loads; each lane then produces R consecutive outputs from registers
// (fir_core.h).  Taps are wave-uniform: read through the scalar cache into SGPRs, never LDS/VGPR.
#include "fir_core.h"
#include "fir_tile.h"
#include "redio_internal.h"

namespace redio {

// ---- specialised tiled kernel ------------------------------------------------------------------
template <typename T, int K, int D, int R, bool FUSED, int NT>
__global__ __launch_bounds__(NT) void fir_tiled_kernel(const T *__restrict__ x, long n_in,
                                                       const float *__restrict__ taps,
                                                       T *__restrict__ y, long n_out, int vec_ok)
{
    using G = FirGeom<K, D, R>;
    constexpr int TILE_OUT = NT * R;
    constexpr int TILE_IN = G::tile_in(TILE_OUT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *xs = reinterpret_cast<T *>(smem);

    const long tile = blockIdx.x;
    load_tile<T, G, NT, TILE_IN>(x, n_in, tile * (long)TILE_OUT * D, xs, vec_ok != 0);
    __syncthreads();

    T acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = T{};
    fir_lane<T, K, D, R, FUSED>(xs, (int)threadIdx.x, taps, acc);

    const long o0 = tile * (long)TILE_OUT + (long)threadIdx.x * R;
    if (o0 + R <= n_out) {
        constexpr int BYTES = R * sizeof(T);
        if constexpr (BYTES % 16 == 0) {
            if (vec_ok) { // y base 16-B aligned and o0*sizeof(T) a multiple of 16
                float4 *y4 = reinterpret_cast<float4 *>(y + o0);
                const float *a = reinterpret_cast<const float *>(acc);
#pragma unroll
                for (int q = 0; q < BYTES / 16; ++q) y4[q] = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
                return;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) y[o0 + r] = acc[r];
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (o0 + r < n_out) y[o0 + r] = acc[r];
    }
}

// ---- direct kernel: any K and D, no tile (L1/L2 serve the overlap); one output per thread -------
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void fir_direct_kernel(const T *__restrict__ x, const float *__restrict__ taps,
                                                         int K, long D, T *__restrict__ y, long n_out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const T *w = x + i * D;
    T acc{};
    for (int j = 0; j < K; ++j) acc = mac<FUSED>(w[j], taps[j], acc);
    y[i] = acc;
}

template <typename T, int K, int D, int R, bool FUSED>
static hipError_t launch_tiled(const T *x, long n_in, const float *taps, T *y, long n_out, hipStream_t s)
{
    constexpr int NT = 256;
    using G = FirGeom<K, D, R>;
    constexpr int TILE_OUT = NT * R;
    constexpr size_t LDS = (size_t)G::lds_elems(TILE_OUT) * sizeof(T);
    static_assert(LDS <= 160 * 1024, "tile does not fit LDS");
    auto kern = fir_tiled_kernel<T, K, D, R, FUSED, NT>;
    if (LDS > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return e;
    }
    const long ntiles = (n_out + TILE_OUT - 1) / TILE_OUT;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(NT), LDS, s, x, n_in, taps, y, n_out, vec_ok);
    return hipGetLastError();
}

template <typename T, bool FUSED>
static hipError_t launch_fir_t(const T *x, long n_in, const float *taps, int K, long D, T *y, long n_out, hipStream_t s)
{
    if (n_out <= 0) return hipSuccess;
    // specialisations for the configurations BASELINE.json names (63 / 127 taps, decimate 1 / 5)
    if (K == 127 && D == 5) return launch_tiled<T, 127, 5, 4, FUSED>(x, n_in, taps, y, n_out, s);
    if (K == 127 && D == 1) return launch_tiled<T, 127, 1, 8, FUSED>(x, n_in, taps, y, n_out, s);
    if (K == 63 && D == 1) return launch_tiled<T, 63, 1, 8, FUSED>(x, n_in, taps, y, n_out, s);
    if (K == 63 && D == 5) return launch_tiled<T, 63, 5, 4, FUSED>(x, n_in, taps, y, n_out, s);
    const long nb = (n_out + 255) / 256;
    hipLaunchKernelGGL((fir_direct_kernel<T, FUSED>), dim3((unsigned)nb), dim3(256), 0, s, x, taps, K, D, y, n_out);
    return hipGetLastError();
}

hipError_t launch_fir_v4_127_5(const float2 *x, const float *taps, float2 *y, long nblocks, bool fused, hipStream_t s); // chain_v4.hip

hipError_t launch_fir(const void *x, long n_in, const float *taps, int K, long D, void *y, long n_out,
                      bool cplx, bool fused, hipStream_t s)
{
    // the north-star shape runs on the chain kernel's data path (wave-private images, halo carried in
    // LDS, register prefetch): whole 1024-output blocks there, the remainder on the tiled kernel
    if (cplx && K == 127 && D == 5 && n_out >= 1024 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
        const long nblocks = n_out / 1024;
        hipError_t e = launch_fir_v4_127_5((const float2 *)x, taps, (float2 *)y, nblocks, fused, s);
        if (e != hipSuccess) return e;
        const long done = nblocks * 1024;
        if (done == n_out) return hipSuccess;
        x = (const float2 *)x + done * 5;
        y = (float2 *)y + done;
        n_in -= done * 5;
        n_out -= done;
    }
    if (cplx) {
        if (fused) return launch_fir_t<float2, true>((const float2 *)x, n_in, taps, K, D, (float2 *)y, n_out, s);
        return launch_fir_t<float2, false>((const float2 *)x, n_in, taps, K, D, (float2 *)y, n_out, s);
    }
    if (fused) return launch_fir_t<float, true>((const float *)x, n_in, taps, K, D, (float *)y, n_out, s);
    return launch_fir_t<float, false>((const float *)x, n_in, taps, K, D, (float *)y, n_out, s);
}

} // namespace redio
