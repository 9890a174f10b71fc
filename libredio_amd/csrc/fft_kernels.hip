// fft_kernels.hip -- gfx950 kernels behind kissfft::fft (src/kissfft/src/kissfft.rs:18-31) and the
// kiss_fft_* C symbols it binds (:11-16).  Unnormalised in both directions, interleaved cf32,
// natural-order output; arithmetic order of the published kissfft butterflies (fft_core.h).
//
// Bound: HBM (16 B per sample, 5 log2 N flop per sample).  Three data-movement strategies:
//   fft1k_wave_kernel   N = 1024: one wavefront per transform, 16 points per lane in registers,
//                       two padded LDS exchanges, no workgroup barrier at all.
//   fft_lds_kernel      any N that fits LDS: one workgroup per transform, in-place stages in LDS.
//   fft_global_*        larger N: digit-reversal copy + one launch per stage in global memory.
#include "redio_internal.h"
#include "fft_wave.h"

namespace redio {

template <bool INV>
__global__ __launch_bounds__(256) void fft1k_wave_kernel(const float2 *in, float2 *out,
                                                         const float2 *__restrict__ tw, long nbatch, long in_stride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float2 *ex = reinterpret_cast<float2 *>(smem) + wave * FFT1K_LDS;
    const long b = (long)blockIdx.x * 4 + wave;
    if (b >= nbatch) return; // wave-uniform
    fft1k_wave<INV>(in + b * in_stride, out + b * 1024, ex, tw, lane);
}

// ---- any N that fits LDS: one workgroup per transform ----------------------------------------
template <bool INV>
__global__ __launch_bounds__(256) void fft_lds_kernel(FftPlanDev p, const float2 *in,
                                                      float2 *out, long in_stride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *A = reinterpret_cast<float2 *>(smem);
    float2 *B = A + p.nfft;
    const int n = p.nfft, tid = threadIdx.x, nt = blockDim.x;
    const float2 *src = in + (long)blockIdx.x * in_stride;
    float2 *dst = out + (long)blockIdx.x * n;
    for (int P = tid; P < n; P += nt) A[P] = src[p.leaf_src[P]];
    __syncthreads();
    for (int s = p.nstages - 1; s >= 0; --s) {
        const FftStage st = p.st[s];
        if (st.p <= 5) {
            const int nb = n / st.p;
            for (int b = tid; b < nb; b += nt) fft_stage_butterfly<INV>(A, p.tw, st, b);
            __syncthreads();
        } else {
            // one output element per thread iteration: e = g*p*m + u + q1*m
            const int pm = st.p * st.m;
            for (int e = tid; e < n; e += nt) {
                const int g = e / pm, r = e - g * pm, q1 = r / st.m, u = r - q1 * st.m;
                B[e] = fft_generic_output(A, p.tw, st, n, g, u, q1);
            }
            __syncthreads();
            float2 *t = A; A = B; B = t;
        }
    }
    for (int P = tid; P < n; P += nt) dst[P] = A[P];
}

// ---- large N: global-memory stages -----------------------------------------------------------
__global__ __launch_bounds__(256) void fft_global_leaf_kernel(FftPlanDev p, const float2 *__restrict__ in,
                                                              float2 *__restrict__ out, long total, long in_stride)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long b = i / p.nfft;
    const int P = (int)(i - b * p.nfft);
    out[i] = in[b * in_stride + p.leaf_src[P]];
}

template <bool INV>
__global__ __launch_bounds__(256) void fft_global_stage_kernel(FftPlanDev p, int s, float2 *data, long total_bfly)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total_bfly) return;
    const FftStage st = p.st[s];
    const int nb = p.nfft / st.p;
    const long b = i / nb;
    const int bf = (int)(i - b * nb);
    fft_stage_butterfly<INV>(data + b * p.nfft, p.tw, st, bf);
}

// ---- N = 65536: two passes of four in-LDS radix-4 stages (fft_core.h, "65536-point transform") ----
// One workgroup per 256 x 16 tile; rows are 128 contiguous bytes in memory.  PASS 0 gathers the
// digit-reversed input (in -> out), PASS 1 works in place on out.  32 B of HBM traffic per sample
// (twice the one-pass minimum; a 512 KiB transform does not fit LDS).
template <bool INV, int PASS>
__global__ __launch_bounds__(256) void fft64k_pass_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw,
                                                          long in_stride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *L = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    const long xf = blockIdx.x >> 4; // transform
    const int c = blockIdx.x & 15;   // tile
    const float2 *src = PASS == 0 ? in + xf * in_stride : out + xf * F64K_N;
    float2 *dst = out + xf * F64K_N;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, row = e >> 4, col = e & 15;
        if (PASS == 0) L[rev4_of_8bit(row) * F64K_LD + col] = src[f64k_p0_src(c, row, col)];
        else L[row * F64K_LD + col] = src[f64k_p1_pos(c, row, col)];
    }
    __syncthreads();
    const int col = tid & 15;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) f64k_tile_butterfly<INV>(L, tw, PASS, t, col, (tid >> 4) + 16 * u, F64K_COLS * c + col);
        __syncthreads();
    }
    if (PASS == 0) {
#pragma unroll 4
        for (int it = 0; it < 16; ++it) dst[f64k_p0_dst(c, tid, it)] = L[tid * F64K_LD + it]; // 2 KiB runs per column
    } else {
#pragma unroll 4
        for (int it = 0; it < 16; ++it) {
            const int e = tid + 256 * it, row = e >> 4, cc = e & 15;
            dst[f64k_p1_pos(c, row, cc)] = L[row * F64K_LD + cc];
        }
    }
}

template <bool INV>
static hipError_t launch_fft64k(const float2 *in, float2 *out, const float2 *tw, long nbatch, long in_stride, hipStream_t s)
{
    const size_t lds = 256 * F64K_LD * sizeof(float2);
    const unsigned grid = (unsigned)(nbatch * 16);
    hipLaunchKernelGGL((fft64k_pass_kernel<INV, 0>), dim3(grid), dim3(256), lds, s, in, out, tw, in_stride);
    hipLaunchKernelGGL((fft64k_pass_kernel<INV, 1>), dim3(grid), dim3(256), lds, s, in, out, tw, in_stride);
    return hipGetLastError();
}

// ---- overlap-save at N = 65536 in three passes instead of six ------------------------------------
// forward pass 1, the spectrum product and inverse pass 0 touch the same 256 x 16 tile (column k0 of the
// forward output IS the stride-256 leaf set of the inverse transform: f64k_p1_pos == f64k_p0_src), so
// they run back to back in LDS; inverse pass 1 scales and writes only the `hop` valid outputs.  Same
// butterflies, same C_MUL, same scale as transform -> multiply -> transform -> copy: bit-identical.
__global__ __launch_bounds__(256) void ovsave64k_mid_kernel(const float2 *__restrict__ a, float2 *__restrict__ b,
                                                            const float2 *__restrict__ tw_f, const float2 *__restrict__ tw_i,
                                                            const float2 *__restrict__ Hc)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *L = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    const long xf = blockIdx.x >> 4;
    const int c = blockIdx.x & 15;
    const float2 *src = a + xf * F64K_N;
    float2 *dst = b + xf * F64K_N;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, row = e >> 4, col = e & 15;
        L[row * F64K_LD + col] = src[f64k_p1_pos(c, row, col)];
    }
    __syncthreads();
    const int col = tid & 15;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) { // forward stages m = 256 .. 16384
#pragma unroll
        for (int u = 0; u < 4; ++u) f64k_tile_butterfly<false>(L, tw_f, 1, t, col, (tid >> 4) + 16 * u, F64K_COLS * c + col);
        __syncthreads();
    }
    float2 v[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, row = e >> 4, cc = e & 15;
        v[it] = cmul_rn(L[row * F64K_LD + cc], Hc[f64k_p1_pos(c, row, cc)]);
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) { // the inverse transform's leaf order along the row index
        const int e = tid + 256 * it, row = e >> 4, cc = e & 15;
        L[rev4_of_8bit(row) * F64K_LD + cc] = v[it];
    }
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < 4; ++t) { // inverse stages m = 1 .. 64
#pragma unroll
        for (int u = 0; u < 4; ++u) f64k_tile_butterfly<true>(L, tw_i, 0, t, col, (tid >> 4) + 16 * u, F64K_COLS * c + col);
        __syncthreads();
    }
#pragma unroll 4
    for (int it = 0; it < 16; ++it) dst[f64k_p0_dst(c, tid, it)] = L[tid * F64K_LD + it];
}

__global__ __launch_bounds__(256) void ovsave64k_last_kernel(const float2 *__restrict__ b, float2 *__restrict__ out,
                                                             const float2 *__restrict__ tw_i, long hop, float scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *L = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    const long xf = blockIdx.x >> 4;
    const int c = blockIdx.x & 15;
    const float2 *src = b + xf * F64K_N;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, row = e >> 4, col = e & 15;
        L[row * F64K_LD + col] = src[f64k_p1_pos(c, row, col)];
    }
    __syncthreads();
    const int col = tid & 15;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int u = 0; u < 4; ++u) f64k_tile_butterfly<true>(L, tw_i, 1, t, col, (tid >> 4) + 16 * u, F64K_COLS * c + col);
        __syncthreads();
    }
    float2 *dst = out + xf * hop;
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int e = tid + 256 * it, row = e >> 4, cc = e & 15;
        const int pos = f64k_p1_pos(c, row, cc);
        const float2 y = L[row * F64K_LD + cc];
        if (pos < hop) dst[pos] = make_float2(mul_rn(y.x, scale), mul_rn(y.y, scale));
    }
}

hipError_t launch_ovsave64k(const float2 *x, long hop, float2 *a, float2 *b, const float2 *tw_f, const float2 *tw_i, const float2 *Hc,
                            float2 *out, long nblk, float scale, hipStream_t s)
{
    const size_t lds = 256 * F64K_LD * sizeof(float2);
    const unsigned grid = (unsigned)(nblk * 16);
    hipLaunchKernelGGL((fft64k_pass_kernel<false, 0>), dim3(grid), dim3(256), lds, s, x, a, tw_f, hop);
    hipLaunchKernelGGL(ovsave64k_mid_kernel, dim3(grid), dim3(256), lds, s, a, b, tw_f, tw_i, Hc);
    hipLaunchKernelGGL(ovsave64k_last_kernel, dim3(grid), dim3(256), lds, s, b, out, tw_i, hop, scale);
    return hipGetLastError();
}

hipError_t launch_fft(const FftPlanDev &p, const float2 *in, float2 *out, long nbatch, hipStream_t s, long in_stride)
{
    if (in_stride <= 0) in_stride = p.nfft; // consecutive messages; smaller strides give overlapping blocks (overlap-save)
    if (nbatch <= 0) return hipSuccess;
    const bool inv = p.inverse != 0;
    if (p.nfft == 1024) {
        const size_t lds = 4 * FFT1K_LDS * sizeof(float2);
        const unsigned grid = (unsigned)((nbatch + 3) / 4);
        if (inv) hipLaunchKernelGGL(fft1k_wave_kernel<true>, dim3(grid), dim3(256), lds, s, in, out, p.tw, nbatch, in_stride);
        else hipLaunchKernelGGL(fft1k_wave_kernel<false>, dim3(grid), dim3(256), lds, s, in, out, p.tw, nbatch, in_stride);
        return hipGetLastError();
    }
    if (p.nfft == F64K_N) {
        if (in == out) return hipErrorNotSupported; // pass 0 is a global transposition: the C-ABI layer stages in-place calls
        return inv ? launch_fft64k<true>(in, out, p.tw, nbatch, in_stride, s) : launch_fft64k<false>(in, out, p.tw, nbatch, in_stride, s);
    }
    bool generic = false;
    for (int i = 0; i < p.nstages; ++i) generic |= p.st[i].p > 5;
    const size_t lds = (size_t)p.nfft * sizeof(float2) * (generic ? 2 : 1);
    if (lds <= 128 * 1024) {
        auto kf = fft_lds_kernel<false>;
        auto ki = fft_lds_kernel<true>;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(inv ? ki : kf),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const int nt = p.nfft >= 1024 ? 256 : (p.nfft >= 256 ? 128 : 64);
        if (inv) hipLaunchKernelGGL(ki, dim3((unsigned)nbatch), dim3(nt), lds, s, p, in, out, in_stride);
        else hipLaunchKernelGGL(kf, dim3((unsigned)nbatch), dim3(nt), lds, s, p, in, out, in_stride);
        return hipGetLastError();
    }
    if (generic || in == out) return hipErrorNotSupported; // the C-ABI layer routes in-place calls through a temporary
    const long total = nbatch * p.nfft;
    hipLaunchKernelGGL(fft_global_leaf_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, p, in, out, total, in_stride);
    for (int st = p.nstages - 1; st >= 0; --st) {
        const long nb = nbatch * (p.nfft / p.st[st].p);
        const unsigned grid = (unsigned)((nb + 255) / 256);
        if (inv) hipLaunchKernelGGL(fft_global_stage_kernel<true>, dim3(grid), dim3(256), 0, s, p, st, out, nb);
        else hipLaunchKernelGGL(fft_global_stage_kernel<false>, dim3(grid), dim3(256), 0, s, p, st, out, nb);
    }
    return hipGetLastError();
}

} // namespace redio
