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
#include "fft_big_core.h"

namespace redio {

// Cache policy of the batched one-wave transforms' two streams (bit 0: non-temporal loads, bit 1: non-temporal stores).  Measured in round 6
// (-DREDIO_EXP_FFT_NT=3 against this default, alternating processes on one box, profiles/r06_fft_nt.txt): 1024 points 16 % SLOWER non-temporal
// (0.76 -> 0.91 ms per 2^28 points: a wavefront stores a block as sixteen 512-byte pieces 2 KiB apart, which the L2 merges into whole lines),
// 256 and 64 points within 1 %.  The default policy stays.
#ifndef REDIO_EXP_FFT_NT
#define REDIO_EXP_FFT_NT 0
#endif
typedef float fft_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 fft_ld_once(const float2 *p)
{
#if REDIO_EXP_FFT_NT & 1
    const fft_v2f v = __builtin_nontemporal_load(reinterpret_cast<const fft_v2f *>(p));
    return make_float2(v.x, v.y);
#else
    return *p;
#endif
}
// a global destination whose element stores are non-temporal: dst[i] = v (the one-wave programs store through a templated pointer)
struct FftOnceOut {
    float2 *p;
    struct Ref {
        float2 *q;
        __device__ __forceinline__ void operator=(float2 v) const
        {
#if REDIO_EXP_FFT_NT & 2
            __builtin_nontemporal_store(fft_v2f{v.x, v.y}, reinterpret_cast<fft_v2f *>(q));
#else
            *q = v;
#endif
        }
    };
    __device__ __forceinline__ Ref operator[](long i) const { return Ref{p + i}; }
};

constexpr int FFT1K_PER_WAVE = 8; // blocks per wavefront of the 1024-point overlap-save kernel: its 54 lane-dependent twiddles are loaded once and reused
constexpr int FFT1K_RUN = 4;      // transforms per wavefront of the plain transform (27 twiddles; round 3: 2 / 4 / 8 / 16 per wave 0.808 / 0.757 / 0.782 / 0.797 ms per 2^28 points)
template <bool INV>
__global__ __launch_bounds__(256) void fft1k_wave_kernel(const float2 *in, float2 *out,
                                                         const float2 *__restrict__ tw, long nbatch, long in_stride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *ex = reinterpret_cast<float2 *>(smem) + wave * FFT1K_LDS;
    const long b0 = ((long)blockIdx.x * 4 + wave) * FFT1K_RUN;
    if (b0 >= nbatch) return; // wave-uniform
    const long b1 = (b0 + FFT1K_RUN < nbatch) ? b0 + FFT1K_RUN : nbatch;
    Fft1kTw t;
    fft1k_load_tw(t, lane, tw);
    float2 v[16], nx[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = fft_ld_once(in + b0 * in_stride + lane + 64 * i);
    for (long b = b0; b < b1; ++b) {
        const long bn = (b + 1 < b1) ? b + 1 : b; // prefetch the next transform's input under this one's arithmetic
#pragma unroll
        for (int i = 0; i < 16; ++i) nx[i] = fft_ld_once(in + bn * in_stride + lane + 64 * i);
        fft1k_wave_regs<INV>(v, FftOnceOut{out + b * 1024}, ex, tw, t, lane);
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = nx[i];
    }
}

// ---- overlap-save with 1024-point blocks: one wavefront per block, nothing but registers in between -------
// The forward transform leaves X[lane + 64 q + 256 j] in v[4 q + j]; the inverse transform wants
// v'[t] = Y[lane + 64 t], t = q + 4 j -- a register renaming.  Load, transform, product with conj(H), inverse
// transform, scale and the store of the hop valid outputs, eight blocks per wave with the 54 lane-dependent
// twiddles of both directions held in registers.
__global__ __launch_bounds__(256) void ovsave1k_kernel(const float2 *__restrict__ x, long hop, const float2 *__restrict__ tw_f,
                                                       const float2 *__restrict__ tw_i, const float2 *__restrict__ Hc,
                                                       float2 *__restrict__ out, long nblk, float scale)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *ex = reinterpret_cast<float2 *>(smem) + wave * FFT1K_LDS;
    const long b0 = ((long)blockIdx.x * 4 + wave) * FFT1K_PER_WAVE;
    if (b0 >= nblk) return; // wave-uniform
    const long b1 = (b0 + FFT1K_PER_WAVE < nblk) ? b0 + FFT1K_PER_WAVE : nblk;
    Fft1kTw tf, ti;
    fft1k_load_tw(tf, lane, tw_f);
    fft1k_load_tw(ti, lane, tw_i);
    for (long b = b0; b < b1; ++b) {
        float2 v[16], w[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) v[t] = x[b * hop + lane + 64 * t];
        fft1k_wave_stages0to3<false>(v, ex, tw_f, tf, lane);
        fft1k_passC<false>(v, tf);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) w[q + 4 * j] = cmul_rn(v[4 * q + j], Hc[lane + 64 * (q + 4 * j)]); // 8 KiB table: L1-resident
        wave_lds_fence();
        fft1k_wave_stages0to3<true>(w, ex, tw_i, ti, lane);
        fft1k_passC<true>(w, ti);
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pos = lane + 64 * q + 256 * j;
                if (pos < hop) out[b * hop + pos] = make_float2(mul_rn(w[4 * q + j].x, scale), mul_rn(w[4 * q + j].y, scale));
            }
    }
}

hipError_t launch_ovsave1k(const float2 *x, long hop, const float2 *tw_f, const float2 *tw_i, const float2 *Hc, float2 *out, long nblk,
                           float scale, hipStream_t s)
{
    const size_t lds = 4 * FFT1K_LDS * sizeof(float2);
    const unsigned grid = (unsigned)((nblk + 4 * FFT1K_PER_WAVE - 1) / (4 * FFT1K_PER_WAVE));
    hipLaunchKernelGGL(ovsave1k_kernel, dim3(grid), dim3(256), lds, s, x, hop, tw_f, tw_i, Hc, out, nblk, scale);
    return hipGetLastError();
}

// ---- N = 256: four transforms per wavefront -------------------------------------------------------
// Four independent 256-point transforms are exactly the four leaf blocks of the 1024-point flow without
// its last stage: block j works on the virtual input x[4 i + j] = X_j[i] with every fourth twiddle of the
// 1024 table, which is the 256 table entry for entry (exact scalings of the phase).  So a wave loads
// v[t] = X_{lane & 3}[(lane >> 2) + 16 t] (128-byte runs), runs stages m = 1 .. 64 of fft_wave.h and
// stores block j as transform j.
struct TwShift {
    const float2 *p; int sh;
    __device__ __forceinline__ float2 operator[](int i) const { return p[i >> sh]; }
};

template <bool INV>
__global__ __launch_bounds__(256) void fft256_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, long nbatch,
                                                     long in_stride)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *ex = reinterpret_cast<float2 *>(smem) + wave * FFT1K_LDS;
    const long b0 = ((long)blockIdx.x * 4 + wave) * 4;
    if (b0 >= nbatch) return; // wave-uniform
    const long bj = (b0 + (lane & 3) < nbatch) ? b0 + (lane & 3) : nbatch - 1;
    float2 v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = fft_ld_once(in + bj * in_stride + (lane >> 2) + 16 * t);
    const TwShift tw1k = {tw, 2};
    Fft1kTw t;
    fft1k_load_tw(t, lane, tw1k); // the last-stage entries are loaded but unused (indices stay inside the table)
    fft1k_wave_stages0to3<INV>(v, ex, tw1k, t, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (b0 + j >= nbatch) break;
#pragma unroll
        for (int q = 0; q < 4; ++q) FftOnceOut{out}[(b0 + j) * 256 + lane + 64 * q] = v[4 * q + j];
    }
}

// ---- N = 64: sixteen transforms per wavefront -----------------------------------------------------
// The same embedding one level down: sixteen 64-point transforms are the sixteen 64-position leaf blocks
// of the 1024-point flow after its first three stages (m = 1, 4, 16); block d1 + 4 d0 works on the
// virtual input x[d0 + 4 d1 + 16 i] with every sixteenth twiddle of the 1024 table (= the 64 table).
// The wave stages its 16 x 64 inputs through LDS (512-byte coalesced loads; row stride 66 float2 makes
// the per-lane gather conflict-free) and stores block b as transform b in 128-byte runs.
constexpr int FFT64_ROW = 66;
template <bool INV>
__global__ __launch_bounds__(256) void fft64_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, long nbatch,
                                                    long in_stride)
{
    static_assert(16 * FFT64_ROW <= FFT1K_LDS, "the staging image reuses the wave's exchange area");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *ex = reinterpret_cast<float2 *>(smem) + wave * FFT1K_LDS;
    const long b0 = ((long)blockIdx.x * 4 + wave) * 16;
    if (b0 >= nbatch) return; // wave-uniform
    float2 v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const long b = (b0 + t < nbatch) ? b0 + t : nbatch - 1;
        v[t] = fft_ld_once(in + b * in_stride + lane);
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) ex[t * FFT64_ROW + lane] = v[t];
    wave_lds_fence();
    const int blk = (lane & 3) * 4 + ((lane >> 2) & 3); // transform of virtual sample n = lane + 64 t
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = ex[blk * FFT64_ROW + (lane >> 4) + 4 * t];
    const TwShift tw1k = {tw, 4};
    fft1k_passA<INV>(v, tw1k);
    wave_lds_fence();
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
        for (int k3 = 0; k3 < 4; ++k3) ex[fft1k_A_store(lane, k3, k4)] = v[k3 + 4 * k4];
    wave_lds_fence();
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = ex[fft1k_B_load(lane, e)];
    const int k = lane >> 2; // stage m = 16: k = k4 + 4 k3
    const float2 w1 = tw[k], w2 = tw[2 * k], w3 = tw[3 * k];
#pragma unroll
    for (int d1 = 0; d1 < 4; ++d1) bfly4<INV>(v[d1], v[d1 + 4], v[d1 + 8], v[d1 + 12], w1, w2, w3);
    // v[d1 + 4 k2] is output k + 16 k2 of transform d1 + 4 d0, d0 = lane & 3
#pragma unroll
    for (int d1 = 0; d1 < 4; ++d1) {
        const long b = b0 + d1 + 4 * (lane & 3);
        if (b < nbatch) {
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) FftOnceOut{out}[b * 64 + k + 16 * k2] = v[d1 + 4 * k2];
        }
    }
}

// the one-wave programs (2048 and 4096 points, also as the quarters of 8192 and 16384) read their twiddles from a
// stage-ordered copy of the table: the stage with sub-length m = NS / (4 fs) starts at m - ML (ML = 1 for 4096 = 4^6,
// 2 for 2048 = 2 * 4^5) and holds T[(n - 1) m + k] = tw[n k fs] (fftbig_tables_build), so that lanes with neighbouring k
// read neighbouring entries (in table order the 64 twiddles of a wave's stage-4 load are spread over 16 to 64 cache lines)
template <int NS, int ML>
struct TwProgram { const float2 *T; };
template <int NS, int ML>
__device__ __forceinline__ float2 tw_get(TwProgram<NS, ML> p, unsigned k, unsigned fs, unsigned n)
{
    const unsigned m = NS / (4 * fs);
    return p.T[(m - ML) + (n - 1) * m + k];
}
template <typename TwPtr>
__device__ __forceinline__ float2 tw_get(TwPtr tw, unsigned k, unsigned fs, unsigned n) { return tw[n * k * fs]; }

// ---- small powers of two and N = 2 * 4^L up to 512 (2, 4, 8, 16, 32, 128, 512; 2048 and 8192 have their own kernels below): compile-time stages in LDS
// kissfft factors 2 * 4^L as 4, 4, ..., 4, 2 with the radix-2 stage innermost.  One 256-thread workgroup
// handles 4096 points (8192 for the largest size): max(1, 4096 / N) transforms.  Coalesced load with the
// digit reversal applied on the LDS side, then register passes over LDS (one pad float2 per 8 keeps the
// 8-point first pass conflict-free): [radix-2 + radix-4 on 8 consecutive positions], then pairs of radix-4
// stages on 16 points per thread, a single radix-4 stage if one is left, coalesced store.  Every index is
// a compile-time shift; butterflies, twiddle indices and stage order are kissfft's (bit-identical).
template <int LOG2N>
struct FftP2 {
    static constexpr int N = 1 << LOG2N, L4 = LOG2N / 2;
    static constexpr bool ODD = (LOG2N & 1) != 0; // a radix-2 stage innermost
    static constexpr int E = N >= 4096 ? N : 4096; // points per workgroup
    static constexpr int T = E / N;                // transforms per workgroup
    // one pad float2 per 8.  Round 6 modelled every LDS access of the kernels on this image (tools/p2_lds_model.py: 50 % of the LDS-array cycles at 256
    // points are bank conflicts, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE measures 47 %) and tried one pad per 16 for 128 points and more, which halves
    // them in the model -- measured: 256 channels + 0.8 %, 1024 channels - 3.7 %, 512 - 1.5 %, the 128-point transform - 2.3 %
    // (profiles/r06_p2_lds_padding.txt): the LDS array is not what these kernels wait for.  Kept at 8.
    __device__ static __forceinline__ int phys(int e) { return e + (e >> 3); }
    static constexpr int LDS_ELEMS = E + (E >> 3) + 8;
    // leaf position of input index n: the top bit is the radix-2 digit, base-4 digits reverse onto N/4, N/16, ...
    __device__ static __forceinline__ int leaf_pos(int n)
    {
        int P = ODD ? n >> (2 * L4) : 0;
#pragma unroll
        for (int i = 0; i < L4; ++i) P += ((n >> (2 * i)) & 3) * (N >> (2 * i + 2));
        return P;
    }
};

// LB: the workgroup barriers between the stages order LDS traffic only and the caller keeps global requests in flight across them
// (pfb_p2_kernel): `s_waitcnt lgkmcnt(0); s_barrier` instead of __syncthreads(), whose fence also waits for those requests (vmcnt(0))
template <bool LB>
__device__ __forceinline__ void fftp2_barrier()
{
    if constexpr (LB) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else __syncthreads();
}
template <int LOG2N, bool INV, int M, typename TwPtr, bool LB = false>
__device__ __forceinline__ void fftp2_rest(float2 *Ls, TwPtr tw, int tid)
{
    using F = FftP2<LOG2N>;
    constexpr int N = F::N, E = F::E;
    if constexpr (M * 4 <= N / 4) { // two stages: sub-lengths M and 4M on 16 points base + j*M
        constexpr int FS = N / (4 * M), FS2 = N / (16 * M);
#pragma unroll 1
        for (int g = tid; g < E / 16; g += 256) {
            const int xf = g / (N / 16), gl = g % (N / 16);
            const int blk = gl / M, kk = gl % M;
            const int base = xf * N + blk * 16 * M + kk;
            float2 a[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) a[j] = Ls[F::phys(base + j * M)];
            const float2 t1 = tw_get(tw, (unsigned)kk, (unsigned)FS, 1), t2 = tw_get(tw, (unsigned)kk, (unsigned)FS, 2), t3 = tw_get(tw, (unsigned)kk, (unsigned)FS, 3);
#pragma unroll
            for (int q = 0; q < 4; ++q) bfly4<INV>(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3], t1, t2, t3);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k2 = kk + u * M;
                bfly4<INV>(a[u], a[u + 4], a[u + 8], a[u + 12], tw_get(tw, (unsigned)k2, (unsigned)FS2, 1), tw_get(tw, (unsigned)k2, (unsigned)FS2, 2), tw_get(tw, (unsigned)k2, (unsigned)FS2, 3));
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) Ls[F::phys(base + j * M)] = a[j];
        }
        fftp2_barrier<LB>();
        fftp2_rest<LOG2N, INV, 16 * M, TwPtr, LB>(Ls, tw, tid);
    } else if constexpr (M <= N / 4) { // one stage left
        constexpr int FS = N / (4 * M);
#pragma unroll 1
        for (int g = tid; g < E / 4; g += 256) {
            const int xf = g / (N / 4), gl = g % (N / 4);
            const int blk = gl / M, kk = gl % M;
            const int base = xf * N + blk * 4 * M + kk;
            float2 a0 = Ls[F::phys(base)], a1 = Ls[F::phys(base + M)], a2 = Ls[F::phys(base + 2 * M)], a3 = Ls[F::phys(base + 3 * M)];
            bfly4<INV>(a0, a1, a2, a3, tw_get(tw, (unsigned)kk, (unsigned)FS, 1), tw_get(tw, (unsigned)kk, (unsigned)FS, 2), tw_get(tw, (unsigned)kk, (unsigned)FS, 3));
            Ls[F::phys(base)] = a0; Ls[F::phys(base + M)] = a1; Ls[F::phys(base + 2 * M)] = a2; Ls[F::phys(base + 3 * M)] = a3;
        }
        fftp2_barrier<LB>();
    }
}

// every stage of the transforms of one workgroup image in LDS (leaf order in, natural order out); ends with a workgroup barrier
template <int LOG2N, bool INV, bool LB = false>
__device__ __forceinline__ void fftp2_lds_stages(float2 *Ls, const float2 *__restrict__ tw, const float2 *__restrict__ Tord, int tid)
{
    using F = FftP2<LOG2N>;
    constexpr int N = F::N, E = F::E;
    if constexpr (!F::ODD) {
        fftp2_rest<LOG2N, INV, 1, const float2 *, LB>(Ls, tw, tid); // powers of four: stages m = 1, 4, ... straight away
    } else if constexpr (N == 2) {
        for (int g = tid; g < E / 2; g += 256) {
            float2 a0 = Ls[F::phys(2 * g)], a1 = Ls[F::phys(2 * g + 1)];
            bfly2(a0, a1, tw[0]);
            Ls[F::phys(2 * g)] = a0; Ls[F::phys(2 * g + 1)] = a1;
        }
        fftp2_barrier<LB>();
    } else {
        // first pass: radix-2 (m = 1) then radix-4 (m = 2) on 8 consecutive positions
        constexpr int FS = N / 8;
        const float2 one = tw[0], w1 = tw[FS], w2 = tw[2 * FS], w3 = tw[3 * FS];
#pragma unroll 1
        for (int g = tid; g < E / 8; g += 256) {
            float2 a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = Ls[F::phys(8 * g + j)];
#pragma unroll
            for (int q = 0; q < 4; ++q) bfly2(a[2 * q], a[2 * q + 1], one);
            bfly4<INV>(a[0], a[2], a[4], a[6], one, one, one);
            bfly4<INV>(a[1], a[3], a[5], a[7], w1, w2, w3);
#pragma unroll
            for (int j = 0; j < 8; ++j) Ls[F::phys(8 * g + j)] = a[j];
        }
        fftp2_barrier<LB>();
        if constexpr (LOG2N == 9) fftp2_rest<LOG2N, INV, 8, TwProgram<512, 2>, LB>(Ls, TwProgram<512, 2>{Tord}, tid); // 512: the stage-ordered copy (+6 %; nothing below)
        else fftp2_rest<LOG2N, INV, 8, const float2 *, LB>(Ls, tw, tid);
    }
}

template <int LOG2N, bool INV>
__global__ __launch_bounds__(256) void fft_p2_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, const float2 *__restrict__ Tord,
                                                     long nbatch, long in_stride)
{
    using F = FftP2<LOG2N>;
    constexpr int N = F::N, E = F::E, T = F::T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *Ls = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    const long b0 = (long)blockIdx.x * T;
#pragma unroll 4
    for (int e = tid; e < E; e += 256) {
        const int xf = e / N, n = e % N;
        const long b = (b0 + xf < nbatch) ? b0 + xf : nbatch - 1;
        Ls[F::phys(xf * N + F::leaf_pos(n))] = in[b * in_stride + n];
    }
    __syncthreads();
    fftp2_lds_stages<LOG2N, INV>(Ls, tw, Tord, tid);
#pragma unroll 4
    for (int e = tid; e < E; e += 256) {
        const int xf = e / N;
        if (b0 + xf < nbatch) out[(b0 + xf) * N + (e % N)] = Ls[F::phys(e)];
    }
}

template <int LOG2N>
static hipError_t launch_fft_p2(const float2 *in, float2 *out, const float2 *tw, const float2 *Tord, long nbatch, long in_stride, bool inv,
                                hipStream_t s)
{
    using F = FftP2<LOG2N>;
    const size_t lds = (size_t)F::LDS_ELEMS * sizeof(float2);
    auto kf = fft_p2_kernel<LOG2N, false>;
    auto ki = fft_p2_kernel<LOG2N, true>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(inv ? ki : kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const unsigned grid = (unsigned)((nbatch + F::T - 1) / F::T);
    if (LOG2N == 9 && !Tord) return hipErrorInvalidValue;
    if (inv) hipLaunchKernelGGL(ki, dim3(grid), dim3(256), lds, s, in, out, tw, Tord, nbatch, in_stride);
    else hipLaunchKernelGGL(kf, dim3(grid), dim3(256), lds, s, in, out, tw, Tord, nbatch, in_stride);
    return hipGetLastError();
}

// ---- polyphase channelizer with M = 32, 128, 256, 512 or 1024 channels in ONE kernel (round 4): branch filters into the LDS image, then the
// M-point transform of fft_p2_kernel on the image, then the rows out -- 16 bytes per sample through HBM instead of the 32 of the two-pass
// form (pfb_api.hip: pfb_branch_kernel + the plan's transform).  v[t][m] = fold_p x[(t + p) M + m] * h[M p + m] (ascending p: dsputils.rs:31),
// kissfft's M-point forward transform across the branches of each row: the bits of oracle orc_pfb_channelizer.
// A workgroup iteration is 4096 points.  Up to 256 channels: 16 rows of each of its G = 256 / M row streams, thread (m, g) walks stream g;
// above: 4096 / M rows of ONE stream, a thread owns M / 256 channels.  A stream is a contiguous range of rows whose P - 1 rows of filter
// history are carried in registers, so an input row is loaded once (M * 8 contiguous bytes), and the next iteration's rows are requested
// before this one's arithmetic.
// PAIR: a thread owns two NEIGHBOURING channels (2 m, 2 m + 1) and loads them with one 16-byte access (M / 2 threads per row, so more
// row streams per workgroup and 8 rows per iteration); otherwise one channel per thread (or M / 256 channels, 256 apart), 8-byte loads.
#ifndef REDIO_EXP_PFB_NT
#define REDIO_EXP_PFB_NT 3 // bit 0: non-temporal row loads, bit 1: non-temporal row stores (pfb_kernels.hip: why)
#endif
typedef float p2_v2f __attribute__((ext_vector_type(2)));
typedef float p2_v4f __attribute__((ext_vector_type(4)));
template <int LOG2M, int P, bool FUSED, bool PAIR>
__global__ __launch_bounds__(256) void pfb_p2_kernel(const float2 *__restrict__ x, const float *__restrict__ h, const float2 *__restrict__ tw,
                                                     const float2 *__restrict__ Tord, float2 *__restrict__ out, long rows, long rps, int ngroups)
{
    using F = FftP2<LOG2M>;
    constexpr int M = F::N;
    constexpr int NP = PAIR ? (M <= 512 ? 1 : M / 512) : (M <= 256 ? 1 : M / 256); // loads per thread and row
    constexpr int CPT = PAIR ? 2 * NP : NP, MT = M / CPT, G = 256 / MT, TR = 16 / CPT; // channels per thread, threads per row, streams, rows per iteration
    static_assert(F::E == 4096 && G * TR * M == 4096 && P >= 2 && P <= 16 && MT <= 256, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *Ls = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x, m = tid % MT, g = tid / MT;
    auto chan = [&](int c) { return PAIR ? 2 * m + (c & 1) + 512 * (c >> 1) : m + 256 * c; }; // channel of slot c
    const long s0 = (long)blockIdx.x * G;                 // first stream of this workgroup: the one with the most rows
    const long t0 = (s0 + g) * rps, last_in_row = rows + P - 2;
    const long rows0 = (s0 * rps + rps < rows ? rps : rows - s0 * rps);
    const int iters = (int)((rows0 + TR - 1) / TR);       // workgroup-uniform
    float gt[CPT][P];
#pragma unroll
    for (int c = 0; c < CPT; ++c)
#pragma unroll
        for (int p = 0; p < P; ++p) gt[c][p] = h[M * p + chan(c)];
    float2 hist[CPT][P - 1], ra[CPT][TR], rb[CPT][TR];
    // The transform's twiddles live in LDS for the life of the workgroup (round 5): read from global memory inside the stages they are vector
    // loads whose wait (vmcnt: loads return in order) also waits for the NEXT iteration's rows requested at the top of this one -- and every
    // barrier below is an LDS-only barrier for the same reason (__syncthreads() waits for all requests in flight).  M <= 1024 entries.
    float2 *Ltw = Ls + F::LDS_ELEMS, *Ltord = Ltw + M; // the M-entry table; for 512 channels also the stage-ordered copy (510 entries) behind it
    for (int i = tid; i < M; i += 256) Ltw[i] = tw[i];
    if constexpr (LOG2M == 9)
        for (int i = tid; i < 510; i += 256) Ltord[i] = Tord[i];
    // rows past the stream's end are clamped (their outputs are never stored); ONE path, never skipped: the compiler can then count the
    // requests in flight at every use instead of waiting for all of them
    auto ld_row = [&](long r, float2 *dst /* [CPT], stride given by `step` */, int step) {
        const float2 *rowp = x + (long)M * (r < last_in_row ? r : last_in_row);
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            if (PAIR) {
#if REDIO_EXP_PFB_NT & 1
                const p2_v4f v = __builtin_nontemporal_load(reinterpret_cast<const p2_v4f *>(rowp + 2 * m + 512 * q));
#else
                const float4 v = *reinterpret_cast<const float4 *>(rowp + 2 * m + 512 * q);
#endif
                dst[(2 * q) * step] = make_float2(v.x, v.y); dst[(2 * q + 1) * step] = make_float2(v.z, v.w);
            } else {
#if REDIO_EXP_PFB_NT & 1
                const p2_v2f v = __builtin_nontemporal_load(reinterpret_cast<const p2_v2f *>(rowp + m + 256 * q));
                dst[q * step] = make_float2(v.x, v.y);
#else
                dst[q * step] = rowp[m + 256 * q];
#endif
            }
        }
    };
#pragma unroll
    for (int p = 0; p < P - 1; ++p) ld_row(t0 + p, &hist[0][p], P - 1);
#pragma unroll
    for (int ti = 0; ti < TR; ++ti) ld_row(t0 + P - 1 + ti, &ra[0][ti], TR);
    const int cpg = M / ngroups;
    fftp2_barrier<true>(); // the twiddle copy is complete
    // one iteration: rows tb .. tb + TR - 1 from `cur`, the next iteration's rows requested into `nx` (the loop below is unrolled by two with
    // the two register sets swapping roles: a copy nx -> cur of a loop-carried array lands behind the iteration's stores and waits for them)
    auto iteration = [&](int it, float2(&cur)[CPT][TR], float2(&nx)[CPT][TR]) {
        const long tb = t0 + (long)TR * it;
#pragma unroll
        for (int ti = 0; ti < TR; ++ti) ld_row(tb + TR + P - 1 + ti, &nx[0][ti], TR);
#pragma unroll
        for (int c = 0; c < CPT; ++c) {
            const int lp = (TR * g) * M + F::leaf_pos(chan(c));
#pragma unroll
            for (int ti = 0; ti < TR; ++ti) { // output row tb + ti: input rows tb + ti + p, p = 0 .. P - 1 (the window is [hist | cur])
                float2 acc = make_float2(0.f, 0.f);
#pragma unroll
                for (int p = 0; p < P; ++p) acc = mac<FUSED>(ti + p < P - 1 ? hist[c][ti + p] : cur[c][ti + p - (P - 1)], gt[c][p], acc);
                Ls[F::phys(lp + ti * M)] = acc;
            }
            float2 hn[P - 1]; // the last P - 1 rows of [hist | cur]
#pragma unroll
            for (int p = 0; p < P - 1; ++p) hn[p] = p + TR < P - 1 ? hist[c][p + TR] : cur[c][p + TR - (P - 1)];
#pragma unroll
            for (int p = 0; p < P - 1; ++p) hist[c][p] = hn[p];
        }
        fftp2_barrier<true>();
        fftp2_lds_stages<LOG2M, false, true>(Ls, Ltw, Ltord, tid);
#pragma unroll 2
        for (int e = 2 * tid; e < 4096; e += 512) { // two neighbouring channels per thread: one 16-byte store
            const int xf = e / M, n = e % M, gg = xf / TR, ti = xf % TR;
            const long sbase = (s0 + gg) * rps, row = sbase + (long)TR * it + ti, rend = sbase + rps < rows ? sbase + rps : rows;
            if (row < rend) {
                const float2 v0 = Ls[F::phys(e)], v1 = Ls[F::phys(e + 1)];
                float2 *o16 = ngroups == 1 ? out + row * M + n : out + (long)(n / cpg) * rows * cpg + row * cpg + (n % cpg);
                if (ngroups == 1 || cpg >= 2) {
#if REDIO_EXP_PFB_NT & 2
                    __builtin_nontemporal_store(p2_v4f{v0.x, v0.y, v1.x, v1.y}, reinterpret_cast<p2_v4f *>(o16));
#else
                    *reinterpret_cast<float4 *>(o16) = make_float4(v0.x, v0.y, v1.x, v1.y);
#endif
                }
                else { out[(long)n * rows + row] = v0; out[(long)(n + 1) * rows + row] = v1; }
            }
        }
        fftp2_barrier<true>();
    };
    for (int it = 0; it < iters; it += 2) {
        iteration(it, ra, rb);
        if (it + 1 < iters) iteration(it + 1, rb, ra); // workgroup-uniform
    }
}

bool pfb_p2_supported(int nchan, int taps_per_branch)
{
    return (nchan == 32 || nchan == 128 || nchan == 256 || nchan == 512 || nchan == 1024) && (taps_per_branch == 4 || taps_per_branch == 8 || taps_per_branch == 16);
}
template <int LOG2M, int P, bool PAIR>
static hipError_t launch_pfb_p2_t(const float2 *x, const float *h, const float2 *tw, const float2 *Tord, float2 *out, long rows, int ngroups, bool fused,
                                  hipStream_t s)
{
    using F = FftP2<LOG2M>;
    constexpr int M = F::N, NP = PAIR ? (M <= 512 ? 1 : M / 512) : (M <= 256 ? 1 : M / 256), CPT = PAIR ? 2 * NP : NP, G = 256 / (M / CPT), TR = 16 / CPT;
    if (LOG2M == 9 && !Tord) return hipErrorInvalidValue;
    // the image + the twiddles the shape needs (M entries; 512 channels: + the 510-entry stage-ordered copy).  Round 5 reserved 1024 entries
    // for every shape: 45.1 KB, three workgroups per CU where four were launched (advisor, round 5); now 37.2-39 KB up to 256 channels
    const size_t lds = (size_t)(F::LDS_ELEMS + (LOG2M == 9 ? 1022 : M)) * sizeof(float2);
    // contiguous row ranges per stream, a multiple of the iteration's rows; about four workgroups per CU -- also for 512 / 1024 channels,
    // where three are resident: sized for three the launch is 16 % SLOWER (0.866 -> 1.027 ms, profiles/r06_c4gen_ab.txt: the fourth
    // quarter of the streams is what evens out the tail); at least 64 rows (the P - 1 row prologue)
    long streams = 4L * num_cus() * G;
    long rps = (rows + streams - 1) / streams;
    rps = ((rps + TR - 1) / TR) * TR;
    if (rps < 64) rps = 64;
    const long nstreams = (rows + rps - 1) / rps;
    const unsigned grid = (unsigned)((nstreams + G - 1) / G);
    if (fused) hipLaunchKernelGGL((pfb_p2_kernel<LOG2M, P, true, PAIR>), dim3(grid), dim3(256), lds, s, x, h, tw, Tord, out, rows, rps, ngroups);
    else hipLaunchKernelGGL((pfb_p2_kernel<LOG2M, P, false, PAIR>), dim3(grid), dim3(256), lds, s, x, h, tw, Tord, out, rows, rps, ngroups);
    return hipGetLastError();
}
// out: [row][M] (ngroups == 1) or [group][row][M / ngroups]; tw: the M-entry forward table; Tord: the 512-point plan's stage-ordered copy (512 channels only)
hipError_t launch_pfb_p2(const float2 *x, const float *h, const float2 *tw, const float2 *Tord, float2 *out, long rows, int nchan, int taps_per_branch,
                         int ngroups, bool fused, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    if (!pfb_p2_supported(nchan, taps_per_branch) || ngroups < 1 || nchan % ngroups) return hipErrorNotSupported;
    if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) return hipErrorNotSupported; // 16-byte stores
    // 16-byte row loads (two neighbouring channels per thread, 16-byte aligned input) where a thread owns several channels anyway: 512 channels
    // 0.993 -> 0.972 ms, 1024 channels 1.031 -> 0.957 ms per 2^28 samples; up to 256 channels the second channel's window costs more registers
    // than the wider load saves (256 channels, 16 taps: 0.96 -> 1.57 ms), so those keep one channel per thread (profiles/r04_channelizer_pair_loads_ab.txt).
    // Measurement builds: REDIO_PFB_NO_PAIR forces the 8-byte form, REDIO_PFB_PAIR the 16-byte form
    const bool pair = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (nchan >= 512 || measure_env("REDIO_PFB_PAIR")) && !measure_env("REDIO_PFB_NO_PAIR");
#define REDIO_PFB_P2(L, Q)                                                                                                    \
    if (nchan == (1 << L) && taps_per_branch == Q)                                                                            \
        return pair ? launch_pfb_p2_t<L, Q, true>(x, h, tw, Tord, out, rows, ngroups, fused, s) : launch_pfb_p2_t<L, Q, false>(x, h, tw, Tord, out, rows, ngroups, fused, s);
    REDIO_PFB_P2(5, 4) REDIO_PFB_P2(5, 8) REDIO_PFB_P2(5, 16) REDIO_PFB_P2(7, 4) REDIO_PFB_P2(7, 8) REDIO_PFB_P2(7, 16) REDIO_PFB_P2(8, 4) REDIO_PFB_P2(8, 8) REDIO_PFB_P2(8, 16)
    REDIO_PFB_P2(9, 4) REDIO_PFB_P2(9, 8) REDIO_PFB_P2(9, 16) REDIO_PFB_P2(10, 4) REDIO_PFB_P2(10, 8) REDIO_PFB_P2(10, 16)
#undef REDIO_PFB_P2
    return hipErrorNotSupported;
}

// ---- every 2^a 3^b 5^c size up to 8192 without a kernel of its own (the sizes kiss_fft_next_fast_size returns; see the
// dispatch list): the same scheme with a compile-time factor list ----
// kissfft's factor order (4s, then 2, then 3, 5; fft_plan_stages) evaluated at compile time; radix-4 neighbours run
// as register pairs -- as do any two neighbours of up to 25 points (3x2, 5x5, 5x2, 5x4 ...) -- and a stage left
// over as one pass over padded LDS; 4096 / N (at least one) transforms per
// workgroup.  Butterflies, twiddle indices and stage order are the table-driven kernel's, so results are the same bits.
template <int N>
struct FftCt {
    static constexpr int isqrt() { int r = 0; while ((r + 1) * (r + 1) <= N) ++r; return r; }
    static constexpr int MAXS = 16;
    struct List { int n; int p[MAXS], m[MAXS], fs[MAXS]; };
    static constexpr List make()
    {
        List l{};
        int p = 4, n = N, fstride = 1;
        const int fsq = isqrt();
        do {
            while (n % p) {
                switch (p) {
                case 4: p = 2; break;
                case 2: p = 3; break;
                default: p += 2; break;
                }
                if (p > fsq) p = n;
            }
            n /= p;
            l.p[l.n] = p; l.m[l.n] = n; l.fs[l.n] = fstride;
            fstride *= p;
            ++l.n;
        } while (n > 1);
        return l;
    }
    static constexpr List L = make();
    static constexpr int T = N >= 4096 ? 1 : 4096 / N;
    static constexpr int E = T * N;
    static constexpr int LDS_ELEMS = E + (E >> 3) + 8;
    __device__ static __forceinline__ int phys(int e) { return e + (e >> 3); }
    __device__ static __forceinline__ int leaf_pos(int n)
    {
        int P = 0;
#pragma unroll
        for (int s = 0; s < L.n; ++s) P += ((n / L.fs[s]) % L.p[s]) * L.m[s];
        return P;
    }
    static constexpr bool supported()
    {
        for (int s = 0; s < L.n; ++s)
            if (L.p[s] > 5) return false;
        return true;
    }
    // the stage-ordered twiddle copy of the plan (redio_api.hip): stage s holds T[toff(s) + (n - 1) m + k] = tw[n k fstride],
    // n = 1 .. p - 1, k < m, so lanes with neighbouring k read neighbouring entries
    static constexpr int toff(int s)
    {
        int o = 0;
        for (int u = 0; u < s; ++u) o += (L.p[u] - 1) * L.m[u];
        return o;
    }
};

struct CtView { // one transform inside the padded batch image
    float2 *p; int off;
    __device__ __forceinline__ float2 &operator[](int i) const { const int e = off + i; return p[e + (e >> 3)]; }
};

// one radix-P butterfly on P contiguous register values: index k inside the sub-length m, twiddle stride fs
// (the argument lists of fft_stage_butterfly_gk)
template <int P, bool INV>
__device__ __forceinline__ void fftct_bfly(float2 (&a)[P], const float2 *__restrict__ Ts, const float2 *__restrict__ tw, int k, int fs, int m)
{
    if constexpr (P == 2) bfly2(a[0], a[1], Ts[k]);
    else if constexpr (P == 3) bfly3(a[0], a[1], a[2], Ts[k], Ts[m + k], tw[fs * m]);
    else if constexpr (P == 4) bfly4<INV>(a[0], a[1], a[2], a[3], Ts[k], Ts[m + k], Ts[2 * m + k]);
    else bfly5(a[0], a[1], a[2], a[3], a[4], Ts[k], Ts[m + k], Ts[2 * m + k], Ts[3 * m + k], tw[fs * m], tw[fs * 2 * m]);
}

template <int NTH>
__device__ __forceinline__ void fftct_sync()
{
    if constexpr (NTH == 64) wave_lds_fence(); // the image belongs to one wave: LDS operations of a wave complete in order
    else __syncthreads();
}

template <int N, bool INV, int S, int NTH = 256, int EPTS = FftCt<N>::E>
__device__ __forceinline__ void fftct_stages(float2 *Ls, const float2 *__restrict__ tw, const float2 *__restrict__ T, int tid)
{
    using F = FftCt<N>;
    if constexpr (S >= 0) {
        constexpr int P = F::L.p[S], M = F::L.m[S], FS = F::L.fs[S];
        constexpr int PO = S >= 1 ? F::L.p[S >= 1 ? S - 1 : 0] : 0; // the next stage out
        if constexpr (S >= 1 && P * PO <= 25) {
            // two stages in registers: P*PO points base + j*M; inner radix P (sub-length M), outer radix PO (sub-length P*M)
            constexpr int FS2 = F::L.fs[S - 1], G = P * PO;
#pragma unroll 1
            for (int g = tid; g < EPTS / G; g += NTH) {
                const int xf = g / (N / G), gl = g % (N / G);
                const int blk = gl / M, kk = gl % M;
                const int base = xf * N + blk * G * M + kk;
                float2 a[G];
#pragma unroll
                for (int j = 0; j < G; ++j) a[j] = Ls[F::phys(base + j * M)];
#pragma unroll
                for (int q = 0; q < PO; ++q) {
                    float2 b[P];
#pragma unroll
                    for (int i = 0; i < P; ++i) b[i] = a[q * P + i];
                    fftct_bfly<P, INV>(b, T + F::toff(S), tw, kk, FS, M);
#pragma unroll
                    for (int i = 0; i < P; ++i) a[q * P + i] = b[i];
                }
#pragma unroll
                for (int u = 0; u < P; ++u) {
                    float2 b[PO];
#pragma unroll
                    for (int i = 0; i < PO; ++i) b[i] = a[u + P * i];
                    fftct_bfly<PO, INV>(b, T + F::toff(S - 1), tw, kk + u * M, FS2, P * M);
#pragma unroll
                    for (int i = 0; i < PO; ++i) a[u + P * i] = b[i];
                }
#pragma unroll
                for (int j = 0; j < G; ++j) Ls[F::phys(base + j * M)] = a[j];
            }
            fftct_sync<NTH>();
            fftct_stages<N, INV, S - 2, NTH, EPTS>(Ls, tw, T, tid);
        } else {
#pragma unroll 1
            for (int bb = tid; bb < EPTS / P; bb += NTH) {
                const int xf = bb / (N / P), b = bb % (N / P);
                const int base = xf * N + (b / M) * P * M + (b % M);
                float2 a[P];
#pragma unroll
                for (int j = 0; j < P; ++j) a[j] = Ls[F::phys(base + j * M)];
                fftct_bfly<P, INV>(a, T + F::toff(S), tw, b % M, FS, M);
#pragma unroll
                for (int j = 0; j < P; ++j) Ls[F::phys(base + j * M)] = a[j];
            }
            fftct_sync<NTH>();
            fftct_stages<N, INV, S - 1, NTH, EPTS>(Ls, tw, T, tid);
        }
    }
}

// 600 ... 1280 points (launch_fft_ct): every wave owns its own transforms (about 1024 points, at least one transform) in its
// own LDS image, so the passes are separated by compiler fences instead of workgroup barriers
template <int N>
struct FftCtW {
    static constexpr int TW = N >= 1024 ? 1 : 1024 / N;      // transforms per wave
    static constexpr int EW = TW * N;
    static constexpr int LDS_W = (EW + (EW >> 3) + 8 + 1) & ~1; // float2 per wave
};
template <int N, bool INV>
__global__ __launch_bounds__(256) void fft_ct_wave_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, const float2 *__restrict__ T, long nbatch, long in_stride)
{
    using F = FftCt<N>;
    using W = FftCtW<N>;
    static_assert(F::supported(), "radices up to 5 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *Ls = reinterpret_cast<float2 *>(smem) + w * W::LDS_W;
    const long b0 = ((long)blockIdx.x * 4 + w) * W::TW;
    if (b0 >= nbatch) return; // wave-uniform; no workgroup barrier in this kernel
#pragma unroll 4
    for (int e = lane; e < W::EW; e += 64) {
        const int xf = e / N, n = e % N;
        const long b = (b0 + xf < nbatch) ? b0 + xf : nbatch - 1;
        Ls[F::phys(xf * N + F::leaf_pos(n))] = in[b * in_stride + n];
    }
    wave_lds_fence();
    fftct_stages<N, INV, F::L.n - 1, 64, W::EW>(Ls, tw, T, lane);
#pragma unroll 4
    for (int e = lane; e < W::EW; e += 64) {
        const int xf = e / N;
        if (b0 + xf < nbatch) out[(b0 + xf) * N + (e % N)] = Ls[F::phys(e)];
    }
}

template <int N, bool INV>
__global__ __launch_bounds__(256) void fft_ct_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, const float2 *__restrict__ T, long nbatch, long in_stride)
{
    using F = FftCt<N>;
    static_assert(F::supported(), "radices up to 5 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *Ls = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    const long b0 = (long)blockIdx.x * F::T;
#pragma unroll 4
    for (int e = tid; e < F::E; e += 256) {
        const int xf = e / N, n = e % N;
        const long b = (b0 + xf < nbatch) ? b0 + xf : nbatch - 1;
        Ls[F::phys(xf * N + F::leaf_pos(n))] = in[b * in_stride + n];
    }
    __syncthreads();
    fftct_stages<N, INV, F::L.n - 1>(Ls, tw, T, tid);
#pragma unroll 4
    for (int e = tid; e < F::E; e += 256) {
        const int xf = e / N;
        if (b0 + xf < nbatch) out[(b0 + xf) * N + (e % N)] = Ls[F::phys(e)];
    }
}

// one transform per NTH-thread workgroup: 1281 ... 2048 points with 128 threads (two waves meet at the barriers instead of
// four), more than 5120 points with 512 (more waves to hide the LDS round trips of a 50-70 KiB image)
template <int N, bool INV, int NTH = 128>
__global__ __launch_bounds__(NTH) void fft_ct_pair_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, const float2 *__restrict__ T, long in_stride)
{
    using F = FftCt<N>;
    static_assert(F::supported(), "radices up to 5 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *Ls = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x;
    const float2 *src = in + (long)blockIdx.x * in_stride;
#pragma unroll 4
    for (int n = tid; n < N; n += NTH) Ls[F::phys(F::leaf_pos(n))] = src[n];
    __syncthreads();
    fftct_stages<N, INV, F::L.n - 1, NTH, N>(Ls, tw, T, tid);
    float2 *dst = out + (long)blockIdx.x * N;
#pragma unroll 4
    for (int n = tid; n < N; n += NTH) dst[n] = Ls[F::phys(n)];
}

template <int N>
static hipError_t launch_fft_ct(const FftPlanDev &p, const float2 *in, float2 *out, long nbatch, long in_stride, bool inv, hipStream_t s)
{
    using F = FftCt<N>;
    // the compile-time list must be the plan's (it is the same algorithm; a mismatch would mean a different build)
    if (p.nstages != F::L.n || !p.tw_pass) return hipErrorNotSupported;
    for (int i = 0; i < F::L.n; ++i)
        if (p.st[i].p != F::L.p[i] || p.st[i].m != F::L.m[i] || p.st[i].fstride != F::L.fs[i]) return hipErrorNotSupported;
    // measured per size: one transform (or a few) per wave wins from 600 to 1280 points (+2 ... +21 %) and at 384 (+13 %);
    // smaller sizes leave lanes idle in the 16-point passes, larger ones take too much LDS per workgroup
    if constexpr ((N >= 600 && N <= 1280) || N == 384) {
        using W = FftCtW<N>;
        const size_t ldsw = (size_t)4 * W::LDS_W * sizeof(float2);
        auto wf = fft_ct_wave_kernel<N, false>;
        auto wi = fft_ct_wave_kernel<N, true>;
        if (ldsw > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(inv ? wi : wf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw);
            if (e != hipSuccess) return e;
        }
        const long nwaves = (nbatch + W::TW - 1) / W::TW;
        const unsigned gridw = (unsigned)((nwaves + 3) / 4);
        if (inv) hipLaunchKernelGGL(wi, dim3(gridw), dim3(256), ldsw, s, in, out, p.tw, p.tw_pass, nbatch, in_stride);
        else hipLaunchKernelGGL(wf, dim3(gridw), dim3(256), ldsw, s, in, out, p.tw, p.tw_pass, nbatch, in_stride);
        return hipGetLastError();
    } else if constexpr (N > 1280 && N <= 2048) { // measured +9 ... +18 % over two transforms per 256-thread workgroup; slower above 2048
        const size_t ldsp = (size_t)(N + (N >> 3) + 8) * sizeof(float2);
        if (inv) hipLaunchKernelGGL((fft_ct_pair_kernel<N, true>), dim3((unsigned)nbatch), dim3(128), ldsp, s, in, out, p.tw, p.tw_pass, in_stride);
        else hipLaunchKernelGGL((fft_ct_pair_kernel<N, false>), dim3((unsigned)nbatch), dim3(128), ldsp, s, in, out, p.tw, p.tw_pass, in_stride);
        return hipGetLastError();
    } else if constexpr (N > 8192) { // 8193 ... 16384 points: the image takes most of a CU's LDS; sixteen waves on it
        const size_t ldsp = (size_t)(N + (N >> 3) + 8) * sizeof(float2);
        auto kf10 = fft_ct_pair_kernel<N, false, 1024>;
        auto ki10 = fft_ct_pair_kernel<N, true, 1024>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(inv ? ki10 : kf10), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
        if (e != hipSuccess) return e;
        if (inv) hipLaunchKernelGGL(ki10, dim3((unsigned)nbatch), dim3(1024), ldsp, s, in, out, p.tw, p.tw_pass, in_stride);
        else hipLaunchKernelGGL(kf10, dim3((unsigned)nbatch), dim3(1024), ldsp, s, in, out, p.tw, p.tw_pass, in_stride);
        return hipGetLastError();
    } else if constexpr (N > 5120) { // eight waves on one transform: measured +10 ... +26 % over four (5120 itself is faster with four)
        const size_t ldsp = (size_t)(N + (N >> 3) + 8) * sizeof(float2);
        auto kf5 = fft_ct_pair_kernel<N, false, 512>;
        auto ki5 = fft_ct_pair_kernel<N, true, 512>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(inv ? ki5 : kf5), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp);
        if (e != hipSuccess) return e;
        if (inv) hipLaunchKernelGGL(ki5, dim3((unsigned)nbatch), dim3(512), ldsp, s, in, out, p.tw, p.tw_pass, in_stride);
        else hipLaunchKernelGGL(kf5, dim3((unsigned)nbatch), dim3(512), ldsp, s, in, out, p.tw, p.tw_pass, in_stride);
        return hipGetLastError();
    } else {
        const size_t lds = (size_t)F::LDS_ELEMS * sizeof(float2);
        auto kf = fft_ct_kernel<N, false>;
        auto ki = fft_ct_kernel<N, true>;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(inv ? ki : kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const unsigned grid = (unsigned)((nbatch + F::T - 1) / F::T);
        if (inv) hipLaunchKernelGGL(ki, dim3(grid), dim3(256), lds, s, in, out, p.tw, p.tw_pass, nbatch, in_stride);
        else hipLaunchKernelGGL(kf, dim3(grid), dim3(256), lds, s, in, out, p.tw, p.tw_pass, nbatch, in_stride);
        return hipGetLastError();
    }
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

// ---- any N with radices 2/3/4/5 whose batch fits LDS: several transforms per workgroup --------------------
// The table-driven kernel above with the launch-bound parts removed: T transforms per 256-thread workgroup
// (about 2048 points), coalesced load with the digit reversal on the LDS side (inverse table leaf_pos), thread
// groups of G lanes per transform, and the stage's (g, k) split by a multiply-high with the plan's reciprocal
// instead of an integer division.  Same butterflies in the same order.
template <bool INV>
__global__ __launch_bounds__(256) void fft_lds_batched_kernel(FftPlanDev p, const float2 *in, float2 *out, long nbatch, long in_stride,
                                                              int T, int G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *A = reinterpret_cast<float2 *>(smem);
    const int n = p.nfft, tid = threadIdx.x, E = T * n;
    const long b0 = (long)blockIdx.x * T;
    for (int e = tid; e < E; e += 256) {
        const int xf = p.magic_n ? (int)__umulhi((unsigned)e, p.magic_n) : e, nn = e - xf * n; // magic 0: divisor 1
        const long b = (b0 + xf < nbatch) ? b0 + xf : nbatch - 1;
        A[xf * n + p.leaf_pos[nn]] = in[b * in_stride + nn];
    }
    __syncthreads();
    const int grp = tid / G, lg = tid - grp * G, ngrp = 256 / G;
    for (int s = p.nstages - 1; s >= 0; --s) {
        const FftStage st = p.st[s];
        const unsigned magic = p.magic_m[s];
        const int nb = n / st.p;
        for (int xf = grp; xf < T; xf += ngrp)
            for (int b = lg; b < nb; b += G) {
                const int g = magic ? (int)__umulhi((unsigned)b, magic) : b;
                fft_stage_butterfly_gk<INV>(A + xf * n, p.tw, st, g, b - g * st.m);
            }
        __syncthreads();
    }
    for (int e = tid; e < E; e += 256) {
        const int xf = p.magic_n ? (int)__umulhi((unsigned)e, p.magic_n) : e;
        if (b0 + xf < nbatch) out[(b0 + xf) * n + (e - xf * n)] = A[e];
    }
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

// a generic-radix stage (prime factor above 5) in global memory, out of place: one output element per thread
template <bool INV>
__global__ __launch_bounds__(256) void fft_global_generic_stage_kernel(FftPlanDev p, int s, const float2 *__restrict__ src, float2 *__restrict__ dst,
                                                                       long total)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const FftStage st = p.st[s];
    const long b = i / p.nfft;
    const int e = (int)(i - b * p.nfft);
    const int pm = st.p * st.m;
    const int g = e / pm, r = e - g * pm, q1 = r / st.m, u = r - q1 * st.m;
    dst[i] = fft_generic_output(src + b * p.nfft, p.tw, st, p.nfft, g, u, q1);
}

// ---- two radix-4 stages on 16 points in registers, twiddles fetched ahead ---------------------------------
// Stage A multiplies by tw[n kA fsA] (n = 1, 2, 3; the same for its four butterflies), stage B butterfly u by
// tw[n (kB + u step) fsB].  The 15 values are loaded as one batch (behind a scheduling barrier where the caller wants the
// next group's batch in flight during the current group's arithmetic); the compiler otherwise sinks each load to its use.
// (FftTw15 and macro16_apply: fft_big_core.h)
template <typename TwPtr>
__device__ __forceinline__ void tw15_load(FftTw15 &T, TwPtr tw, unsigned kA, unsigned fsA, unsigned kB, unsigned step, unsigned fsB)
{
    T.t[0] = tw_get(tw, kA, fsA, 1); T.t[1] = tw_get(tw, kA, fsA, 2); T.t[2] = tw_get(tw, kA, fsA, 3);
#pragma unroll
    for (unsigned u = 0; u < 4; ++u) {
        const unsigned k = kB + u * step;
        T.t[3 + 3 * u] = tw_get(tw, k, fsB, 1); T.t[4 + 3 * u] = tw_get(tw, k, fsB, 2); T.t[5 + 3 * u] = tw_get(tw, k, fsB, 3);
    }
}
// four groups with per-group twiddles: batch g + 1 is requested before group g is computed
template <bool INV, bool AHEAD = true, typename LoadFn>
__device__ __forceinline__ void macro16_x4(float2 (&a)[4][16], FftTw15 &T0, LoadFn load)
{
    FftTw15 Tn;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (AHEAD && g < 3) load(Tn, g + 1);
        RD_SCHED_BARRIER();
        macro16_apply<INV>(a[g], T0);
        if (g < 3) {
            if (AHEAD) T0 = Tn;
            else { RD_SCHED_BARRIER(); load(T0, g + 1); } // fewer live registers, one exposed latency per group
        }
    }
}
// ---- twiddle access of the tile passes (65536 points and the larger powers of two) -----------------
// a wave-uniform row pointer kept in scalar registers: with a run-time row stride the compiler otherwise folds the lane
// offset into a 64-bit vector address per row and runs out of registers
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
    asm volatile("" : "+s"(p));
    return p;
}

// (TwGather / TwOrdered / TwInter, big_macro16, tw_ordered_stage, tw_inter_stage: fft_big_core.h)
__global__ __launch_bounds__(256) void fftbig_tables_inter_kernel(const float2 *__restrict__ tw, float2 *__restrict__ T, unsigned m_lo, int nstages, unsigned N)
{
    const unsigned total = m_lo * (((1u << (2 * nstages)) - 1) / 3); // entries over all stages
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int t = 0;
        while (i >= m_lo * (((1u << (2 * (t + 1))) - 1) / 3)) ++t;
        const unsigned m = m_lo << (2 * t), k = i - m_lo * (((1u << (2 * t)) - 1) / 3), fs = N / (4 * m);
        T[4 * (size_t)i] = tw[k * fs];
        T[4 * (size_t)i + 1] = tw[2 * k * fs];
        T[4 * (size_t)i + 2] = tw[3 * k * fs];
        T[4 * (size_t)i + 3] = make_float2(0.f, 0.f);
    }
}
__global__ __launch_bounds__(256) void fftbig_tables_kernel(const float2 *__restrict__ tw, float2 *__restrict__ T, unsigned m_lo, int nstages, unsigned N)
{
    const unsigned total = m_lo * ((1u << (2 * nstages)) - 1);
    for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        int t = 0;
        while (i >= m_lo * ((1u << (2 * (t + 1))) - 1)) ++t;
        const unsigned m = m_lo << (2 * t), r = i - m_lo * ((1u << (2 * t)) - 1), n = r / m + 1, k = r - (n - 1) * m;
        T[i] = tw[n * k * (N / (4 * m))];
    }
}

// ---- N = 65536, one wavefront per 256 x 16 tile ---------------------------------------------------
// The same two passes of four stages, but a tile belongs to ONE wave: lane (col = lane & 15, q = lane >> 4) loads
// four 16-row groups of its column straight from memory into registers (64 points per lane), runs stages t = 0, 1,
// trades rows with the other three lanes of its column through a wave-private LDS region (four rounds of 64 rows,
// no workgroup barrier), runs stages t = 2, 3 and stores from registers.  One LDS round trip per pass instead of
// three, index arithmetic per lane instead of per element.  Same butterflies in the same order: bit-identical.
// (Round 3: the three 65536-point overlap-save tile kernels compiled for three resident waves per SIMD instead of two -- 168 VGPRs, 3 to 28
//  of them spilled -- run C5 in 3.65 ms instead of 2.91.)
constexpr int F64W_LD = 17;
constexpr int F64W_REGION = 64 * F64W_LD; // float2 per wave

// workgroup b runs on XCD b % 8: give each XCD a contiguous range of tiles, so that the workgroups that share the 2 KiB
// rows of one transform go through the same L2 at about the same time
// rev: the mirror image of that order (launch_fftbig: a pass walks the batch in the direction opposite to the pass before it, so that
// it starts with what that pass wrote last -- the part of the intermediate the 256 MB Infinity Cache still holds)
__device__ __forceinline__ long f64w_first_tile(int rev = 0)
{
    const unsigned g = gridDim.x, b = blockIdx.x, per = g >> 3, rem = g & 7, x = b & 7, i = b >> 3;
    const unsigned logical = x < rem ? x * (per + 1) + i : rem * (per + 1) + (x - rem) * per + i;
    return (long)(rev ? g - 1 - logical : logical) * 4;
}

// a[i][j] = row 16 g + j of the lane's column, g = 4 i + q (BY_CLASS: g = 4 q + i)   ->
// b[x][j] = row s + 16 j of column cc: (s, cc) = (q + 4 x, col), or with TRANSPOSE (lane & 15, 4 (lane >> 4) + x)
template <bool BY_CLASS, bool TRANSPOSE>
__device__ __forceinline__ void f64w_exchange(float2 (&a)[4][16], float2 (&b)[4][16], float2 *Lw, int lane)
{
    const int col = lane & 15, q = lane >> 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) { // round r moves the 64 rows of the four groups {slot r of every q}
#pragma unroll
        for (int j = 0; j < 16; ++j) Lw[(16 * q + j) * F64W_LD + col] = a[r][j];
        wave_lds_fence();
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int jabs = BY_CLASS ? r + 4 * jj : 4 * r + jj; // the group written by lane-quarter jj
                const int s = TRANSPOSE ? col : q + 4 * x;
                const int cc = TRANSPOSE ? 4 * q + x : col;
                b[x][jabs] = Lw[(s + 16 * jj) * F64W_LD + cc];
            }
        wave_lds_fence();
    }
}

// A launch of these tile passes is ONE round: every wavefront of the chip takes one tile, so all of them load together, then all
// compute (memory idle), then all store.  Delaying half of the wavefronts of every workgroup by about the length of a load phase
// lets one half compute while the other half moves data (MI355X_MICROARCH.md, two waves per SIMD, item 9).  REDIO_EXP_BIG_STAGGER:
// the delay in units of s_sleep 127 (127 x 64 cycles, about 4 us); 0 = none.
#ifndef REDIO_EXP_BIG_STAGGER
#define REDIO_EXP_BIG_STAGGER 0
#endif
__device__ __forceinline__ void big_stagger(int w)
{
#if REDIO_EXP_BIG_STAGGER > 0
    if (w >= 2) {
#pragma unroll
        for (int i = 0; i < REDIO_EXP_BIG_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#else
    (void)w;
#endif
}

// data that a multi-pass transform touches ONCE (the caller's input in the gather pass, the caller's output in the last pass):
// non-temporal accesses, so that the 64 MiB intermediates the passes hand to each other keep the caches (round 3;
// -DREDIO_EXP_BIG_NT=0 builds the default-policy form for comparison)
#ifndef REDIO_EXP_BIG_NT
#define REDIO_EXP_BIG_NT 3
#endif
typedef float big_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2 big_ld_once(const float2 *p)
{
#if REDIO_EXP_BIG_NT & 1
    const big_v2f v = __builtin_nontemporal_load(reinterpret_cast<const big_v2f *>(p));
    return make_float2(v.x, v.y);
#else
    return *p;
#endif
}
__device__ __forceinline__ void big_st_once(float2 *p, float2 v)
{
#if REDIO_EXP_BIG_NT & 2
    __builtin_nontemporal_store(big_v2f{v.x, v.y}, reinterpret_cast<big_v2f *>(p));
#else
    *p = v;
#endif
}

} // namespace redio
// Round 4: the two-column ("pair") form of the four-stage tile passes (fft_pair.h, fft_big_core.h): 16-byte global and LDS accesses.
// -DREDIO_TILE_PAIR=0 builds the one-column program below for comparison; both give the same bits.
#ifndef REDIO_TILE_PAIR
#define REDIO_TILE_PAIR 1
#endif
#include "fft_pair.h"
namespace redio {

// overlap-save middle pass: forward pass 1, spectrum product, inverse pass 0 on the same tile.  After the forward
// stages lane (col, q) holds rows s + 16 j, s = q + 4 x: in the inverse transform's digit-reversed order that IS
// group 4 q + x with rows in rev2 order, so the inverse starts from registers without another exchange.
__device__ __forceinline__ void ovsave64k_mid_tile(const float2 *__restrict__ a_blk, float2 *__restrict__ b_blk, const float2 *__restrict__ Tf,
                                                   const float2 *__restrict__ tw_i, const float2 *__restrict__ Hc, int c, int lane, float2 *Lw)
{
#if REDIO_TILE_PAIR
    pw_ovsave64k_mid_tile(a_blk, b_blk, Tf, tw_i, Hc, c, lane, reinterpret_cast<float4 *>(Lw));
    return;
#endif
    const int col = lane & 15, q = lane >> 4;
    const float2 *src = a_blk + F64K_COLS * c;
    float2 *dst = b_blk;
    const unsigned lo_q1 = col + 256u * q, lo_q16 = col + 4096u * q;
    float2 a[4][16], b[4][16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) a[i][j] = (src + 256 * (64 * i + j))[lo_q16];
#pragma unroll
    for (int i = 0; i < 4; ++i) big_macro16<false>(a[i], tw_inter_stage(Tf, 256u, 0), tw_inter_stage(Tf, 256u, 1), (unsigned)(F64K_COLS * c + col), 256u, 0u, 1u);
    f64w_exchange<false, false>(a, b, Lw, lane);
#pragma unroll
    for (int x = 0; x < 4; ++x) big_macro16<false>(b[x], tw_inter_stage(Tf, 256u, 2), tw_inter_stage(Tf, 256u, 3), (unsigned)(F64K_COLS * c + col), 256u, (unsigned)(q + 4 * x), 16u);
    const float2 *hc = Hc + F64K_COLS * c;
#pragma unroll
    for (int x = 0; x < 4; ++x) { // sixteen spectrum taps as one batch of loads (the compiler would wait for them one by one)
        float2 h[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) h[j] = (hc + 256 * (4 * x + 16 * j))[lo_q1];
        RD_SCHED_BARRIER();
#pragma unroll
        for (int j = 0; j < 16; ++j) // row q + 4x + 16j = digit reversal of 16 (4q + x) + rev2(j)
            a[x][((j & 3) << 2) | (j >> 2)] = cmul_rn(b[x][j], h[j]);
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) f64k_macro_regs<true>(a[x], tw_i, 0, 0, 0, 0);
    f64w_exchange<true, true>(a, b, Lw, lane);
#pragma unroll
    for (int x = 0; x < 4; ++x) f64k_macro_regs<true>(b[x], tw_i, 0, 2, col, 0);
    const int rc = ((c & 3) << 2) | (c >> 2);
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int j = 0; j < 16; ++j) (dst + 256 * (64 * x + rc) + 16 * j)[lo_q16] = b[x][j];
}

__global__ __launch_bounds__(256, 2) void ovsave64k_mid_wave_kernel(const float2 *__restrict__ a_in, float2 *__restrict__ b_out,
                                                                 const float2 *__restrict__ Tf, const float2 *__restrict__ tw_i,
                                                                 const float2 *__restrict__ Hc, long ntiles)
{
    __shared__ __attribute__((aligned(16))) float2 Ls[4 * F64W_REGION];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = f64w_first_tile() + w;
    if (tile >= ntiles) return;
    big_stagger(w);
    const long xf = tile >> 4;
    ovsave64k_mid_tile(a_in + xf * F64K_N, b_out + xf * F64K_N, Tf, tw_i, Hc, (int)(tile & 15), lane, Ls + w * F64W_REGION);
}

__device__ __forceinline__ void ovsave64k_last_tile(const float2 *__restrict__ b_blk, float2 *__restrict__ out_blk, const float2 *__restrict__ Ti,
                                                    long hop, float scale, int c, int lane, float2 *Lw)
{
#if REDIO_TILE_PAIR
    pw_ovsave64k_last_tile(b_blk, out_blk, Ti, hop, scale, c, lane, reinterpret_cast<float4 *>(Lw));
    return;
#endif
    const int col = lane & 15, q = lane >> 4;
    const float2 *src = b_blk + F64K_COLS * c;
    float2 *dst = out_blk + F64K_COLS * c;
    const unsigned lo_q1 = col + 256u * q, lo_q16 = col + 4096u * q;
    float2 a[4][16], b[4][16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) a[i][j] = (src + 256 * (64 * i + j))[lo_q16];
#pragma unroll
    for (int i = 0; i < 4; ++i) big_macro16<true>(a[i], tw_inter_stage(Ti, 256u, 0), tw_inter_stage(Ti, 256u, 1), (unsigned)(F64K_COLS * c + col), 256u, 0u, 1u);
    f64w_exchange<false, false>(a, b, Lw, lane);
#pragma unroll
    for (int x = 0; x < 4; ++x) big_macro16<true>(b[x], tw_inter_stage(Ti, 256u, 2), tw_inter_stage(Ti, 256u, 3), (unsigned)(F64K_COLS * c + col), 256u, (unsigned)(q + 4 * x), 16u);
    const long lim = hop - F64K_COLS * c - (long)lo_q1; // pos < hop
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (256 * (4 * x + 16 * j) < lim) big_st_once((dst + 256 * (4 * x + 16 * j)) + lo_q1, make_float2(mul_rn(b[x][j].x, scale), mul_rn(b[x][j].y, scale)));
}

__global__ __launch_bounds__(256, 2) void ovsave64k_last_wave_kernel(const float2 *__restrict__ b_in, float2 *__restrict__ out,
                                                                  const float2 *__restrict__ Ti, long hop, float scale, long ntiles)
{
    __shared__ __attribute__((aligned(16))) float2 Ls[4 * F64W_REGION];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = f64w_first_tile() + w;
    if (tile >= ntiles) return;
    const long xf = tile >> 4;
    ovsave64k_last_tile(b_in + xf * F64K_N, out + xf * hop, Ti, hop, scale, (int)(tile & 15), lane, Ls + w * F64W_REGION);
}

// ---- N = 4096, one wavefront per transform ---------------------------------------------------------
// Position e of the in-place working array = six base-4 digits (d5 .. d0); stage t combines digit d_t with the twiddle
// index e mod 4^t.  A lane keeps 64 points in registers and runs the stages two at a time on 16-point groups:
//   A: stages 0, 1 on (d0, d1)   lane = (d3 d4 d5) = the low digits of the digit-reversed source index, slot = d2
//   B: stages 2, 3 on (d2, d3)   lane = (d1 d4 d5), slot = d0
//   C: stages 4, 5 on (d4, d5)   lane = (d2 d1 d0) = the low digits of the output index, slot = d3
// so the loads and the stores are 512 contiguous bytes per instruction, and the two regroupings go through a wave-private
// LDS region in four rounds of 1024 points (no workgroup barrier anywhere).  kissfft's butterflies in kissfft's order.
constexpr int F4W_SA = 80, F4W_SB = 65;     // padded strides of the two exchange layouts
constexpr int F4W_REGION = 16 * F4W_SA;     // float2 per wave

template <bool INV, bool AHEAD = true, typename TwPtr>
__device__ __forceinline__ void fft4k_wave_regs(float2 (&a)[4][16], float2 (&b)[4][16], TwPtr tw, float2 *Lw, int lane)
{
    const unsigned hi = lane >> 4, low = lane & 15;
    FftTw15 T;
    // A: stages 0, 1 (twiddle index 0, then d0): one batch for the four groups
    tw15_load(T, tw, 0u, 1024u, 0u, 1u, 256u);
    RD_SCHED_BARRIER();
#pragma unroll
    for (int i = 0; i < 4; ++i) macro16_apply<INV>(a[i], T);
    tw15_load(T, tw, 4u * hi, 64u, 4u * hi, 16u, 16u); // group d0 = 0 of B, in flight during the exchange
    // A -> B: round r moves slot d2 = r; the four lanes that share (d4, d5) trade d3 against d1
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int j = 0; j < 16; ++j) Lw[((j & 3) * 4 + (j >> 2)) * F4W_SA + lane] = a[r][j]; // (d0, d1 | d3, d4, d5)
        wave_lds_fence();
#pragma unroll
        for (int d0 = 0; d0 < 4; ++d0)
#pragma unroll
            for (int d3 = 0; d3 < 4; ++d3) b[d0][r + 4 * d3] = Lw[(d0 * 4 + hi) * F4W_SA + d3 * 16 + low];
        wave_lds_fence();
    }
    // B: stages 2, 3 with k = d0 + 4 d1 (d1 = lane >> 4, d0 = slot)
    macro16_x4<INV, AHEAD>(b, T, [&](FftTw15 &Tn, int d0) { tw15_load(Tn, tw, d0 + 4u * hi, 64u, d0 + 4u * hi, 16u, 16u); });
    tw15_load(T, tw, (unsigned)lane, 4u, (unsigned)lane, 256u, 1u); // group d3 = 0 of C
    // B -> C: round r moves d3 = r; all 64 lanes trade (d1, d4, d5) against (d2, d1, d0)
    const unsigned c2 = lane >> 4, c1 = (lane >> 2) & 3, c0 = lane & 3;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int d0 = 0; d0 < 4; ++d0)
#pragma unroll
            for (int d2 = 0; d2 < 4; ++d2) Lw[(d0 * 4 + d2) * F4W_SB + lane] = b[d0][d2 + 4 * r]; // (d0, d2 | d1, d4, d5)
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 16; ++j) a[r][j] = Lw[(c0 * 4 + c2) * F4W_SB + c1 * 16 + (j & 3) * 4 + (j >> 2)]; // j = d4 + 4 d5
        wave_lds_fence();
    }
    // C: stages 4, 5 with k = 64 d3 + lane
    macro16_x4<INV, AHEAD>(a, T, [&](FftTw15 &Tn, int d3) { tw15_load(Tn, tw, 64u * d3 + lane, 4u, 64u * d3 + lane, 256u, 1u); });
}

template <bool INV>
__global__ __launch_bounds__(256, 2) void fft4k_wave_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, long nbatch,
                                                            long in_stride)
{
    __shared__ float2 Ls[4 * F4W_REGION];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long xf = (long)blockIdx.x * 4 + w;
    if (xf >= nbatch) return; // wave-uniform
    const float2 *src = in + xf * in_stride;
    float2 *dst = out + xf * 4096;
    float2 a[4][16], b[4][16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) a[i][j] = (src + 1024 * (j & 3) + 256 * (j >> 2) + 64 * i)[(unsigned)lane];
    RD_SCHED_BARRIER(); // all 64 loads requested before the arithmetic starts
    fft4k_wave_regs<INV>(a, b, TwProgram<4096, 1>{tw}, Ls + w * F4W_REGION, lane); // tw: the plan's stage-ordered copy
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) (dst + 1024 * (j >> 2) + 256 * (j & 3) + 64 * i)[(unsigned)lane] = a[i][j];
}

// overlap-save with 4096-point blocks, one wavefront per block: after the forward stages a lane holds output positions
// e = 1024 d5 + 256 d4 + 64 d3 + lane (registers j = d4 + 4 d5, slot d3).  As input of the inverse transform that index is
// digit-reversed, which lands every point in the same lane and slot with j' = d5 + 4 d4 -- the product with conj(H)
// and the whole inverse start from registers.  One read of the block, one write of the hop valid outputs, four
// wave-private exchanges, no workgroup barrier.  Same operations as transform -> multiply -> transform -> scaled copy.
__global__ __launch_bounds__(256, 2) void ovsave4k_wave_kernel(const float2 *__restrict__ x, long hop, const float2 *__restrict__ tw_f,
                                                               const float2 *__restrict__ tw_i, const float2 *__restrict__ Hc,
                                                               float2 *__restrict__ out, long nblk, float scale)
{
    __shared__ float2 Ls[4 * F4W_REGION];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long blk = (long)blockIdx.x * 4 + w;
    if (blk >= nblk) return; // wave-uniform
    const float2 *src = x + blk * hop;
    float2 *dst = out + blk * hop;
    float2 *Lw = Ls + w * F4W_REGION;
    float2 a[4][16], b[4][16];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) a[i][j] = (src + 1024 * (j & 3) + 256 * (j >> 2) + 64 * i)[(unsigned)lane];
    RD_SCHED_BARRIER();
    fft4k_wave_regs<false>(a, b, TwProgram<4096, 1>{tw_f}, Lw, lane); // tw_f, tw_i: the plans' stage-ordered copies
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) b[i][j] = (Hc + 1024 * (j >> 2) + 256 * (j & 3) + 64 * i)[(unsigned)lane];
        RD_SCHED_BARRIER();
#pragma unroll
        for (int j = 0; j < 16; ++j) b[i][j] = cmul_rn(a[i][j], b[i][j]);
#pragma unroll
        for (int j = 0; j < 16; ++j) a[i][((j & 3) << 2) | (j >> 2)] = b[i][j];
    }
    int lane_i = lane;
    asm volatile("" : "+v"(lane_i)); // recompute the twiddle offsets: keeping the forward transform's sixty alive spills them
    fft4k_wave_regs<true>(a, b, TwProgram<4096, 1>{tw_i}, Lw, lane_i);
    const long lim = hop - lane;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int e = 1024 * (j >> 2) + 256 * (j & 3) + 64 * i;
            if (e < lim) (dst + e)[(unsigned)lane] = make_float2(mul_rn(a[i][j].x, scale), mul_rn(a[i][j].y, scale));
        }
}

// ---- N = 16384 = 4 x 4096: four wavefronts, one sub-transform each, last stage across them -------------------
// Wave q runs the one-wave 4096-point program on x[4 n + q] (its twiddles are every fourth entry of the table, read from the stage-ordered copy T), which
// leaves F_q[k], k = 1024 d5 + 256 d4 + 64 d3 + lane, in its registers.  The last kissfft stage (m = 4096) needs the four
// F_q[k] of one k in one thread: four rounds (d3 = r) through a 32 KiB LDS image, after which thread (wave w, lane) owns
// k = 1024 w + 256 d4 + 64 r + lane and stores X[k + 4096 rr] (512-byte runs).
#ifndef REDIO_F16K_LDS_DEAL
#define REDIO_F16K_LDS_DEAL 1
#endif
// Input of the four-wave kernels (16384 = 4 x 4096, 8192 = 4 x 2048 points): wave q transforms x[4 n + q].  Read by wave q itself that
// is 16 cache lines per 512 useful bytes and wave instruction.  Instead the four waves read the block in 512-byte runs (wave w: samples
// 256 t + 64 w + lane of each round) and DEAL the samples to their sub-sequences through LDS: sample e = 4 n + q goes to plane q, cell n;
// wave q then reads its positions lane-contiguous.  Two rounds of half a block; the plane stride is 8 mod 16 cells, so the four planes
// start 16 banks apart and the 32 lanes of a half-wave (8 cells in each plane) write 64 different banks.  Round 4: 16384 points
// 52.0 -> 55.5 %, 8192 points 57.8 -> 60.5 % (16384 fed contiguous rows of the wrong samples: 59.3 %); 8192-point overlap-save
// 1.45 -> 1.43 ms, 16384-point overlap-save 1.63 -> 1.66 ms: kept on the strided reads (profiles/r04_four_wave_deal_loads.txt).
#ifndef REDIO_F16K_TWO_IMAGES
#define REDIO_F16K_TWO_IMAGES 1
#endif
#ifndef REDIO_OV16K_LDS_DEAL
#define REDIO_OV16K_LDS_DEAL 0
#endif
__device__ __forceinline__ void f16k_deal_load(float2 (&a)[4][16], float2 (&b)[4][16], const float2 *blk, float2 *Ls, int w, int lane)
{
    const float2 *row = blk + 64 * w;
#pragma unroll
    for (int u = 0; u < 64; ++u) b[u >> 4][u & 15] = (row + 8192 * (u >> 5) + 256 * (u & 31))[(unsigned)lane];
    RD_SCHED_BARRIER();
    float2 *cellw = Ls + deal_write_cell(F16K_PS, w, lane, 0);
    const float2 *cellr = Ls + deal_read_cell(F16K_PS, w, lane, 0);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int t = 0; t < 32; ++t) cellw[64 * t] = b[(32 * r + t) >> 4][(32 * r + t) & 15];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (f4k_reg_pos(i, j) / 2048 == r) a[i][j] = cellr[f4k_reg_pos(i, j) - 2048 * r];
        __syncthreads();
    }
}
// the same for 8192 points: rounds of 4096 samples, 1024 positions per wave and round
__device__ __forceinline__ void f8k_deal_load(float2 (&a)[4][8], float2 (&b)[2][16], const float2 *blk, float2 *Ls, int w, int lane)
{
    const float2 *row = blk + 64 * w;
#pragma unroll
    for (int u = 0; u < 32; ++u) b[u >> 4][u & 15] = (row + 4096 * (u >> 4) + 256 * (u & 15))[(unsigned)lane];
    RD_SCHED_BARRIER();
    float2 *cellw = Ls + deal_write_cell(F8K_PS, w, lane, 0);
    const float2 *cellr = Ls + deal_read_cell(F8K_PS, w, lane, 0);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int t = 0; t < 16; ++t) cellw[64 * t] = b[r][t];
        __syncthreads();
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (f2k_reg_pos(d2, j) / 1024 == r) a[d2][j] = cellr[f2k_reg_pos(d2, j) - 1024 * r];
        __syncthreads();
    }
}
static_assert(4 * F16K_PS <= 4 * F4W_REGION + 4096 && 4 * F8K_PS <= 4 * F4W_REGION, "the four planes of a round fit the kernels' LDS");
template <bool INV>
__global__ __launch_bounds__(256, 2) void fft16k_wave_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, const float2 *__restrict__ T, long in_stride)
{
    __shared__ float2 Ls[4 * F4W_REGION + 4096];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float2 *src = in + (long)blockIdx.x * in_stride + w;
    float2 *dst = out + (long)blockIdx.x * 16384;
    float2 a[4][16], b[4][16];
    if (REDIO_F16K_LDS_DEAL) f16k_deal_load(a, b, in + (long)blockIdx.x * in_stride, Ls, w, lane);
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) a[i][j] = (src + 4 * (1024 * (j & 3) + 256 * (j >> 2) + 64 * i))[4u * lane];
        RD_SCHED_BARRIER();
    }
    fft4k_wave_regs<INV>(a, b, TwProgram<4096, 1>{T}, Ls + w * F4W_REGION, lane);
    // [q][1024]; two images alternate so that a round costs ONE barrier: the second one lies where the private images were (every wave is
    // done with those before round 0's barrier), and a round's image is written again two rounds later, after the next round's barrier
    static_assert(4 * F4W_REGION >= 4096, "the second image fits where the private ones were");
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float2 *X = REDIO_F16K_TWO_IMAGES && (r & 1) ? Ls : Ls + 4 * F4W_REGION;
        if (!REDIO_F16K_TWO_IMAGES && r) __syncthreads(); // the previous round has been read
#pragma unroll
        for (int j = 0; j < 16; ++j) X[1024 * w + 64 * j + lane] = a[r][j]; // j = d4 + 4 d5
        __syncthreads();
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
            const int jj = d4 + 4 * w;                 // d5 = w
            const unsigned k = 1024u * w + 256u * d4 + 64u * r + lane;
            float2 f0 = X[64 * jj + lane], f1 = X[1024 + 64 * jj + lane], f2 = X[2048 + 64 * jj + lane], f3 = X[3072 + 64 * jj + lane];
            const TwOrdered lst = tw_ordered_stage(T, 1u, 6);
            bfly4<INV>(f0, f1, f2, f3, lst.get(1, k), lst.get(2, k), lst.get(3, k));
            dst[k] = f0; dst[k + 4096] = f1; dst[k + 8192] = f2; dst[k + 12288] = f3;
        }
    }
}

// overlap-save with 16384-point blocks in one kernel of four wavefronts (two workgroups per CU).  Forward as in
// fft16k_wave_kernel, but the last forward stage runs in rounds over d4 so that its products with conj(H), regrouped
// through a second LDS image, fill one register slot of the inverse sub-transforms per round: sample n = k + 4096 rr
// belongs to inverse wave n & 3 at position n >> 2 = 1024 rr + 256 d5 + 64 d4 + 16 d3 + (lane >> 2).  Then the inverse
// sub-transforms from registers, the inverse last stage, 1/N and the store of the hop valid outputs.  One read of the
// block, one write of the outputs; same operations as transform -> multiply -> transform -> scaled copy.
constexpr int OV16W_YS = 1040; // stride of one inverse wave's image (bank spread for the lane & 3 scatter)
__global__ __launch_bounds__(256, 2) void ovsave16k_wave_kernel(const float2 *__restrict__ x, long hop, const float2 *__restrict__ tw_f,
                                                                const float2 *__restrict__ tw_i, const float2 *__restrict__ Tf,
                                                                const float2 *__restrict__ Ti, const float2 *__restrict__ Hc,
                                                                float2 *__restrict__ out, float scale)
{
    __shared__ float2 Ls[4096 + 4 * OV16W_YS];
    static_assert(4 * F4W_REGION <= 4096 + 4 * OV16W_YS, "the wave-private exchange images live inside the two shared ones");
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float2 *src = x + (long)blockIdx.x * hop + w;
    float2 *dst = out + (long)blockIdx.x * hop;
    float2 *X = Ls, *Y = Ls + 4096, *Lw = Ls + w * F4W_REGION;
    float2 a[4][16], b[4][16];
    static_assert(4 * F16K_PS <= 4096 + 4 * OV16W_YS, "the dealt planes fit");
    static_assert(4 * OV16W_YS >= 4096, "the second image of the inverse last stage fits the product image");
    if (REDIO_OV16K_LDS_DEAL) f16k_deal_load(a, b, x + (long)blockIdx.x * hop, Ls, w, lane); // measured: 1.655 against 1.631 ms (127 taps): not adopted here
    else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) a[i][j] = (src + 4 * (1024 * (j & 3) + 256 * (j >> 2) + 64 * i))[4u * lane];
        RD_SCHED_BARRIER();
    }
    fft4k_wave_regs<false>(a, b, TwProgram<4096, 1>{Tf}, Lw, lane);
    __syncthreads(); // every wave is done with its private image
#pragma unroll
    for (int r = 0; r < 4; ++r) { // d4 = r
#pragma unroll
        for (int d3 = 0; d3 < 4; ++d3)
#pragma unroll
            for (int d5 = 0; d5 < 4; ++d5) X[1024 * w + 256 * d5 + 64 * d3 + lane] = a[d3][r + 4 * d5];
        __syncthreads();
#pragma unroll
        for (int d3 = 0; d3 < 4; ++d3) { // this thread's k: d5 = w
            const unsigned k = 1024u * w + 256u * r + 64u * d3 + lane;
            float2 f[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = X[1024 * q + 256 * w + 64 * d3 + lane];
            const TwOrdered lst = tw_ordered_stage(Tf, 1u, 6);
            bfly4<false>(f[0], f[1], f[2], f[3], lst.get(1, k), lst.get(2, k), lst.get(3, k));
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                Y[OV16W_YS * (lane & 3) + 64 * (rr + 4 * w) + 16 * d3 + (lane >> 2)] = cmul_rn(f[rr], Hc[k + 4096u * rr]);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) b[r][j] = Y[OV16W_YS * w + 64 * j + lane]; // slot r of inverse sub-transform w
    }
    __syncthreads(); // the images are free again
    int lane_i = lane;
    asm volatile("" : "+v"(lane_i)); // fresh twiddle offsets for the inverse (keeping the forward ones alive spills)
    fft4k_wave_regs<true>(b, a, TwProgram<4096, 1>{Ti}, Lw, lane_i);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) { // d3 = r, as in fft16k_wave_kernel: the rounds alternate between the two images, one barrier each
        float2 *Xr = REDIO_F16K_TWO_IMAGES && (r & 1) ? Y : X;
        if (!REDIO_F16K_TWO_IMAGES && r) __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) Xr[1024 * w + 64 * j + lane_i] = b[r][j];
        __syncthreads();
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
            const int jj = d4 + 4 * w;
            const unsigned k = 1024u * w + 256u * d4 + 64u * r + lane_i;
            float2 f[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = Xr[1024 * q + 64 * jj + lane_i];
            const TwOrdered lsti = tw_ordered_stage(Ti, 1u, 6);
            bfly4<true>(f[0], f[1], f[2], f[3], lsti.get(1, k), lsti.get(2, k), lsti.get(3, k));
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if ((long)(k + 4096u * rr) < hop) dst[k + 4096u * rr] = make_float2(mul_rn(f[rr].x, scale), mul_rn(f[rr].y, scale));
        }
    }
}

// ---- N = 2048 = 2 x 4^5, one wavefront per transform; N = 8192 = 4 x 2048, four wavefronts -------------------
// Position e = b0 + 2 d1 + 8 d2 + 32 d3 + 128 d4 + 512 d5 (kissfft runs the radix-2 stage first, on b0).  32 points per lane:
//   A: radix-2 + radix-4 on (b0, d1), four 8-point groups (slot d2)   lane = (d3 d4 d5) = source index mod 64
//   B: stages on (d2, d3), two 16-point groups (slot b0)               lane = (d1 d4 d5)
//   C: stages on (d4, d5), two 16-point groups (slot d3 >> 1)          lane = e mod 64
// with the two regroupings through a wave-private LDS image in two rounds of 1024 points (fft4k_wave_regs' layouts).
template <bool INV, typename TwPtr>
__device__ __forceinline__ void fft2k_wave_regs(float2 (&a)[4][8], float2 (&b)[2][16], TwPtr tw, float2 *Lw, int lane)
{
    const unsigned hi = lane >> 4, low = lane & 15;
    {
        const float2 w0 = tw_get(tw, 0u, 256u, 1), w1 = tw_get(tw, 1u, 256u, 1), w2 = tw_get(tw, 1u, 256u, 2), w3 = tw_get(tw, 1u, 256u, 3);
        RD_SCHED_BARRIER();
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2) {
#pragma unroll
            for (int d1 = 0; d1 < 4; ++d1) bfly2(a[d2][2 * d1], a[d2][2 * d1 + 1], w0);
            bfly4x2<INV>(a[d2][0], a[d2][2], a[d2][4], a[d2][6], w0, w0, w0, a[d2][1], a[d2][3], a[d2][5], a[d2][7], w1, w2, w3);
        }
    }
    FftTw15 T;
    tw15_load(T, tw, 2u * hi, 64u, 2u * hi, 8u, 16u); // group b0 = 0 of B: k = b0 + 2 d1
    // A -> B: round r moves b0 = r; the four lanes that share (d4, d5) trade d3 against d1
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
            for (int d1 = 0; d1 < 4; ++d1) Lw[(d2 * 4 + d1) * F4W_SA + lane] = a[d2][r + 2 * d1]; // (d2, d1 | d3, d4, d5)
        wave_lds_fence();
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
            for (int d3 = 0; d3 < 4; ++d3) b[r][d2 + 4 * d3] = Lw[(d2 * 4 + hi) * F4W_SA + d3 * 16 + low];
        wave_lds_fence();
    }
    {
        FftTw15 Tn;
        tw15_load(Tn, tw, 1u + 2u * hi, 64u, 1u + 2u * hi, 8u, 16u);
        RD_SCHED_BARRIER();
        macro16_apply<INV>(b[0], T);
        macro16_apply<INV>(b[1], Tn);
    }
    tw15_load(T, tw, (unsigned)lane, 4u, (unsigned)lane, 128u, 1u); // slot 0 of C: k = e mod 128
    // B -> C: round r moves d3 >> 1 = r; (d1 d4 d5) against (b0 d1 d2 d3&1)
    const unsigned c0 = lane & 1, c1 = (lane >> 1) & 3, c2 = (lane >> 3) & 3, c3 = lane >> 5;
    float2 (&c)[2][16] = b; // phase C reuses the registers round by round
    float2 t[2][16];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int b0 = 0; b0 < 2; ++b0)
#pragma unroll
            for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
                for (int d3l = 0; d3l < 2; ++d3l) Lw[(b0 * 8 + d2 * 2 + d3l) * F4W_SB + lane] = c[b0][d2 + 4 * (2 * r + d3l)];
        wave_lds_fence();
#pragma unroll
        for (int j = 0; j < 16; ++j) t[r][j] = Lw[(c0 * 8 + c2 * 2 + c3) * F4W_SB + c1 * 16 + (j & 3) * 4 + (j >> 2)]; // j = d4 + 4 d5
        wave_lds_fence();
    }
    {
        FftTw15 Tn;
        tw15_load(Tn, tw, 64u + lane, 4u, 64u + lane, 128u, 1u);
        RD_SCHED_BARRIER();
        macro16_apply<INV>(t[0], T);
        macro16_apply<INV>(t[1], Tn);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) b[r][j] = t[r][j]; // result: b[slot][j = d4 + 4 d5] = position lane + 64 slot + 128 d4 + 512 d5
}

template <bool INV>
__global__ __launch_bounds__(256) void fft2k_wave_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, long nbatch, long in_stride)
{
    __shared__ float2 Ls[4 * F4W_REGION];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long xf = (long)blockIdx.x * 4 + w;
    if (xf >= nbatch) return; // wave-uniform
    const float2 *src = in + xf * in_stride;
    float2 *dst = out + xf * 2048;
    float2 a[4][8], b[2][16];
#pragma unroll
    for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[d2][j] = (src + 64 * (d2 + 4 * (j >> 1) + 16 * (j & 1)))[(unsigned)lane]; // j = b0 + 2 d1
    RD_SCHED_BARRIER();
    fft2k_wave_regs<INV>(a, b, TwProgram<2048, 2>{tw}, Ls + w * F4W_REGION, lane); // tw: the plan's stage-ordered copy
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) (dst + 128 * (j & 3) + 512 * (j >> 2) + 64 * r)[(unsigned)lane] = b[r][j];
}

// overlap-save with 2048-point blocks, one wavefront per block (the 4096-point scheme): after the forward stages a lane holds
// e = lane + 64 (slot + 2 d4 + 8 d5); as input of the inverse transform that is the same lane with d2 + 4 d1 + 16 b0 =
// slot + 2 d4 + 8 d5, a renaming of registers.
__global__ __launch_bounds__(256) void ovsave2k_wave_kernel(const float2 *__restrict__ x, long hop, const float2 *__restrict__ Tf,
                                                            const float2 *__restrict__ Ti, const float2 *__restrict__ Hc,
                                                            float2 *__restrict__ out, long nblk, float scale)
{
    __shared__ float2 Ls[4 * F4W_REGION];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long blk = (long)blockIdx.x * 4 + w;
    if (blk >= nblk) return; // wave-uniform
    const float2 *src = x + blk * hop;
    float2 *dst = out + blk * hop;
    float2 *Lw = Ls + w * F4W_REGION;
    float2 a[4][8], b[2][16];
#pragma unroll
    for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[d2][j] = (src + 64 * (d2 + 4 * (j >> 1) + 16 * (j & 1)))[(unsigned)lane];
    RD_SCHED_BARRIER();
    fft2k_wave_regs<false>(a, b, TwProgram<2048, 2>{Tf}, Lw, lane);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int q5 = r + 2 * (j & 3) + 8 * (j >> 2); // position lane + 64 q5
            a[q5 & 3][(q5 >> 4) + 2 * ((q5 >> 2) & 3)] = cmul_rn(b[r][j], (Hc + 64 * q5)[(unsigned)lane]);
        }
    int lane_i = lane;
    asm volatile("" : "+v"(lane_i)); // fresh twiddle offsets for the inverse
    fft2k_wave_regs<true>(a, b, TwProgram<2048, 2>{Ti}, Lw, lane_i);
    const long lim = hop - lane;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int e = 128 * (j & 3) + 512 * (j >> 2) + 64 * r;
            if (e < lim) (dst + e)[(unsigned)lane] = make_float2(mul_rn(b[r][j].x, scale), mul_rn(b[r][j].y, scale));
        }
}

// 8192: wave q runs the 2048-point program on x[4 n + q]; the last stage (m = 2048) across the waves, as in fft16k_wave_kernel
template <bool INV>
__global__ __launch_bounds__(256) void fft8k_wave_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, const float2 *__restrict__ T, long in_stride)
{
    __shared__ float2 Ls[4 * F4W_REGION]; // the wave-private images, then (after a barrier) the [q][1024] image of the last stage
    static_assert(4 * F4W_REGION >= 4096, "the shared image fits where the private ones were");
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float2 *src = in + (long)blockIdx.x * in_stride + w;
    float2 *dst = out + (long)blockIdx.x * 8192;
    float2 a[4][8], b[2][16];
    if (REDIO_F16K_LDS_DEAL) f8k_deal_load(a, b, in + (long)blockIdx.x * in_stride, Ls, w, lane);
    else {
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[d2][j] = (src + 4 * 64 * (d2 + 4 * (j >> 1) + 16 * (j & 1)))[4u * lane];
        RD_SCHED_BARRIER();
    }
    fft2k_wave_regs<INV>(a, b, TwProgram<2048, 2>{T}, Ls + w * F4W_REGION, lane);
    float2 *X = Ls;
#pragma unroll
    for (int r = 0; r < 2; ++r) { // slot = r
        __syncthreads(); // private images / the previous round are no longer read
#pragma unroll
        for (int j = 0; j < 16; ++j) X[1024 * w + 64 * j + lane] = b[r][j];
        __syncthreads();
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
            const int jj = d4 + 4 * w; // d5 = w
            const unsigned k = 512u * w + 128u * d4 + 64u * r + lane;
            float2 f0 = X[64 * jj + lane], f1 = X[1024 + 64 * jj + lane], f2 = X[2048 + 64 * jj + lane], f3 = X[3072 + 64 * jj + lane];
            const TwOrdered lst = tw_ordered_stage(T, 2u, 5);
            bfly4<INV>(f0, f1, f2, f3, lst.get(1, k), lst.get(2, k), lst.get(3, k));
            dst[k] = f0; dst[k + 2048] = f1; dst[k + 4096] = f2; dst[k + 6144] = f3;
        }
    }
}

// ---- N = 4^L, L = 9 ... 12 (262144 ... 16777216): the 65536-point scheme with one or two more passes -------------
// Position e of the working array has the base-4 digits d0 (first stage) ... d(L-1).  Pass A gathers the digit-reversed
// input and runs stages 0-3 on tiles of 256 rows (d0..d3, 4^L / 256 apart in the source) x 16 source columns, writing the
// working order (2 KiB runs); pass B runs four more stages in place on rows m_lo apart (m_lo = 4^4, then 4^8) x 16
// neighbouring positions; the last one to three stages (rows 65536 apart) need no regrouping at all: a lane keeps whole
// columns in registers and every load and store is 512 contiguous bytes.  16 B/sample per pass.  Twiddle index of the stage
// with sub-length m: (e mod m) * N / (4 m), exactly kissfft's k * fstride.
// T1: the gather pass's own ordered twiddle copy (sub-lengths 1 ... 256: 1023 entries, fftbig_tables_build) -- in the table
// order a wave's loads of one stage touch 16 different cache lines
template <bool INV, typename TA, typename TB, typename TC, typename TD>
__device__ __forceinline__ void fftbig_first_stages(float2 (&a)[4][16], float2 (&b)[4][16], TA t0, TB t1, TC t2, TD t3, int col, int lane, float2 *Lw)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) big_macro16<INV>(a[i], t0, t1, 0u, 1u, 0u, 1u);
    f64w_exchange<false, true>(a, b, Lw, lane);
#pragma unroll
    for (int x = 0; x < 4; ++x) big_macro16<INV>(b[x], t2, t3, 0u, 1u, (unsigned)col, 16u);
}

// one 256 x 16 tile of the gather pass: block at in_blk (source columns 16c .. 16c + 15) -> working order at out_blk
template <bool INV, bool MULH = true>
__device__ __forceinline__ void fftbig_first_tile(const float2 *in_blk, float2 *out_blk, const float2 *__restrict__ tw, int L, unsigned c,
                                                  int lane, float2 *Lw, const float2 *__restrict__ mulH, const float2 *__restrict__ T1)
{
#if REDIO_TILE_PAIR
    (void)tw;
    pw_first_tile<INV, MULH>(in_blk, out_blk, L, c, lane, reinterpret_cast<float4 *>(Lw), mulH, T1);
    return;
#endif
    const unsigned N = 1u << (2 * L), S = N >> 8; // S: source row stride
    const int col = lane & 15, q = lane >> 4;
    const float2 *src = in_blk + 16 * c;
    float2 *dst = out_blk;
    float2 a[4][16], b[4][16];
    const unsigned lo_src = col + 4u * S * q;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) a[i][j] = big_ld_once((src + (long)S * (16 * (((j & 3) << 2) | (j >> 2)) + i)) + lo_src); // source row rev4(16 (4i + q) + j)
    if (mulH) { // overlap-save: the spectrum product on the way in (wave-uniform branch)
        const float2 *hsrc = mulH + 16 * c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float2 hv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) hv[j] = (hsrc + (long)S * (16 * (((j & 3) << 2) | (j >> 2)) + i))[lo_src];
            RD_SCHED_BARRIER();
#pragma unroll
            for (int j = 0; j < 16; ++j) a[i][j] = cmul_rn(a[i][j], hv[j]);
        }
    }
    RD_SCHED_BARRIER();
    (void)tw;
    fftbig_first_stages<INV>(a, b, tw_ordered_stage(T1, 1u, 0), tw_ordered_stage(T1, 1u, 1), tw_ordered_stage(T1, 1u, 2), tw_ordered_stage(T1, 1u, 3), col, lane, Lw);
    // column r = 16c + 4q + x of the source is column h = digit reversal of r (L - 4 digits) of the working array
    unsigned rc = 0;
    for (int d = 0, cc = c; d < L - 6; ++d, cc >>= 2) rc = (rc << 2) | (cc & 3);
    const unsigned hq = 1u << (2 * (L - 6)), hx = hq << 2;
    const unsigned lo_dst = col + 256u * hq * q;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int j = 0; j < 16; ++j) (dst + 256l * (hx * x + rc) + 16 * j)[lo_dst] = b[x][j];
}

template <bool INV, bool MULH = true> // MULH = false: the pass without the spectrum product (mulH is ignored): no registers kept for that branch
__global__ __launch_bounds__(256, 2) void fftbig_first_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, long in_stride,
                                                           long ntiles, int L, const float2 *__restrict__ mulH = nullptr, const float2 *__restrict__ T1 = nullptr)
{
    __shared__ __attribute__((aligned(16))) float2 Ls[4 * F64W_REGION];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = f64w_first_tile() + w;
    if (tile >= ntiles) return;
    const long xf = tile >> (2 * (L - 6));
    const unsigned c = (unsigned)(tile & ((1u << (2 * (L - 6))) - 1)); // source columns 16c .. 16c + 15
    fftbig_first_tile<INV, MULH>(in + xf * in_stride, out + xf * (long)(1u << (2 * L)), tw, L, c, lane, Ls + w * F64W_REGION, mulH, T1);
}

// G128 gather pass (N = 2 * 4^L'; fft_pair.h): one wavefront per tile of 128 source rows x 32 source columns
template <bool INV, bool MULH = true>
__global__ __launch_bounds__(256, 2) void fftbig_g128_kernel(const float2 *in, float2 *out, long in_stride, long ntiles, int lgN,
                                                          const float2 *__restrict__ mulH, const float2 *__restrict__ Tg)
{
    __shared__ float4 Lg[4 * PW_G_UNITS];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = f64w_first_tile() + w;
    if (tile >= ntiles) return;
    const long xf = tile >> (lgN - 12);
    const unsigned ctile = (unsigned)(tile & ((1u << (lgN - 12)) - 1)); // source columns 32 ctile .. + 31
    pw_g128_tile<INV, MULH>(in + xf * in_stride, out + xf * (long)(1u << lgN), lgN, ctile, lane, Lg + w * PW_G_UNITS, mulH, Tg);
}
// G512 gather pass (round 4; fft_pair.h pw_g512_tile): one workgroup of four wavefronts per tile of 512 source rows x 32 source columns
template <bool INV>
__global__ __launch_bounds__(256, 2) void fftbig_g512_kernel(const float2 *in, float2 *out, long in_stride, long ngroups, int lgN, const float2 *__restrict__ Tg5)
{
    __shared__ float4 Lg[4 * PW_G_UNITS];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long group = f64w_first_tile() >> 2;
    if (group >= ngroups) return; // whole workgroup
    const long xf = group >> (lgN - 14);
    const unsigned ctile = (unsigned)(group & ((1u << (lgN - 14)) - 1));
    pw_g512_tile<INV>(in + xf * in_stride, out + xf * (long)(1u << lgN), lgN, ctile, lane, w, Lg, Tg5);
}
// overlap-save with 32768-point blocks, middle pass (fft_pair.h): eight tiles of 256 rows x 16 columns per block
__global__ __launch_bounds__(256, 2) void ovsave32k_mid_kernel(const float2 *__restrict__ a_in, float2 *__restrict__ b_out, const float2 *__restrict__ Tf,
                                                            const float2 *__restrict__ Tgi, const float2 *__restrict__ Hc, long ntiles)
{
    __shared__ float4 Lg[4 * PW_G_UNITS];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = f64w_first_tile() + w;
    if (tile >= ntiles) return;
    const long xf = tile >> 3;
    pw_ovsave32k_mid_tile(a_in + xf * 32768, b_out + xf * 32768, Tf, Tgi, Hc, (int)(tile & 7), lane, Lg + w * PW_G_UNITS);
}

template <bool INV>
__global__ __launch_bounds__(256, 2) void fftbig_mid_kernel(float2 *data, const float2 *__restrict__ T, long ntiles, int lgN, int lm, float2 *__restrict__ vout = nullptr,
                                                         long hop = 0, float scale = 1.0f, int rev = 0)
{
    __shared__ __attribute__((aligned(16))) float2 Ls[4 * F64W_REGION];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = f64w_first_tile(rev) + w;
    if (tile >= ntiles) return;
    float2 *Lw = Ls + w * F64W_REGION;
    const unsigned N = 1u << lgN, m_lo = 1u << lm; // rows m_lo = 2^lm apart
    const long xf = tile >> (lgN - 12);
    const unsigned tt = (unsigned)(tile & ((1u << (lgN - 12)) - 1));
    const unsigned c = tt & ((m_lo >> 4) - 1), h = tt >> (lm - 4); // positions l = 16c + col of block h
    const int col = lane & 15, q = lane >> 4;
    float2 *base = data + xf * (long)N + (long)h * 256 * m_lo + 16 * c;
#if REDIO_TILE_PAIR
    pw_mid_tile<INV>(base, (long)m_lo, 16 * c, T, reinterpret_cast<float4 *>(Lw), lane, vout ? vout + xf * hop : nullptr, (long)h * 256 * m_lo + 16 * c, hop, scale);
    return;
#endif
    const unsigned l = 16 * c + col;
    float2 a[4][16], b[4][16];
    const unsigned lo_ld = col + 16u * m_lo * q, lo_st = col + m_lo * q;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) a[i][j] = (base + (long)m_lo * (64 * i + j))[lo_ld]; // row 16 (4i + q) + j
    RD_SCHED_BARRIER();
#pragma unroll
    for (int i = 0; i < 4; ++i) big_macro16<INV>(a[i], tw_inter_stage(T, m_lo, 0), tw_inter_stage(T, m_lo, 1), l, m_lo, 0u, 1u);
    f64w_exchange<false, false>(a, b, Lw, lane);
#pragma unroll
    for (int x = 0; x < 4; ++x) big_macro16<INV>(b[x], tw_inter_stage(T, m_lo, 2), tw_inter_stage(T, m_lo, 3), l, m_lo, (unsigned)(q + 4 * x), 16u);
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (vout) { // overlap-save: this was the last pass; 1/N and only the hop valid outputs, packed
                const long e = (long)h * 256 * m_lo + (long)m_lo * (q + 4 * x + 16 * j) + l;
                if (e < hop) vout[xf * hop + e] = make_float2(mul_rn(b[x][j].x, scale), mul_rn(b[x][j].y, scale));
            } else (base + (long)m_lo * (4 * x + 16 * j))[lo_st] = b[x][j]; // row q + 4x + 16j
        }
}

// ---- five-stage passes: 4^9 and 4^10 points in TWO passes instead of three -------------------------------------------------
// A tile of 1024 rows x 16 columns belongs to one workgroup: wave w runs the four-stage wave program above on the quarter of
// the rows that forms one 256-row sub-transform (mid pass: rows 256 w .. 256 w + 255; gather pass: source rows 4 rho + w), then
// the fifth stage combines position r of the four quarters.  It runs in four rounds through a 34 KiB LDS image: in round x
// every wave parks its register group b[x][.] (64 rows x 16 columns), and wave w' takes slots 4 w' .. 4 w' + 3 of every
// quarter -- the four inputs of a butterfly -- multiplies by the stage's twiddles and stores the four results (rows r,
// r + 256, r + 512, r + 768 of the tile) straight from registers.  Same butterflies in the same order as the four-stage
// passes followed by a one-stage pass: bit-identical.
constexpr int F5_LD = 17;
struct F5Image { float2 v[4][64][F5_LD]; }; // [quarter][slot 16 q + j][lane & 15]

template <bool INV>
__global__ __launch_bounds__(256, 2) void fftbig_mid5_kernel(float2 *data, const float2 *__restrict__ T, long ngroups, int lgN, int lm,
                                                          float2 *__restrict__ vout = nullptr, long hop = 0, float scale = 1.0f, int rev = 0)
{
#if REDIO_TILE_PAIR // the pair form (fft_pair.h: pw_mid5_tile): two images of 32 KiB, image 0 = the four wave-private images of the four-stage program
    __shared__ float4 Lx[2 * PW_X5_UNITS];
    {
        const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const long group = f64w_first_tile(rev) >> 2;
        if (group >= ngroups) return; // whole workgroup
        const unsigned N = 1u << lgN, m_lo = 1u << lm;
        const long xf = group >> (lgN - 14);
        const unsigned gg = (unsigned)(group & ((1u << (lgN - 14)) - 1));
        const unsigned c = gg & ((m_lo >> 4) - 1), H = gg >> (lm - 4);
        float2 *tile = data + xf * (long)N + (long)H * 1024 * m_lo + 16 * c;
        pw_mid5_tile<INV>(tile, (long)m_lo, 16 * c, T, Lx, lane, w, vout ? vout + xf * hop : nullptr, (long)H * 1024 * m_lo + 16 * c, hop, scale);
        return;
    }
#endif
    __shared__ __attribute__((aligned(16))) float2 Ls[4 * F64W_REGION];
    __shared__ F5Image X;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long group = f64w_first_tile(rev) >> 2;
    if (group >= ngroups) return; // whole workgroup
    float2 *Lw = Ls + w * F64W_REGION;
    const unsigned N = 1u << lgN, m_lo = 1u << lm; // rows m_lo = 2^lm apart, 1024 of them per tile
    const long xf = group >> (lgN - 14);
    const unsigned gg = (unsigned)(group & ((1u << (lgN - 14)) - 1));
    const unsigned c = gg & ((m_lo >> 4) - 1), H = gg >> (lm - 4); // positions l = 16c + col of 1024-row block H
    const int col = lane & 15, q = lane >> 4;
    float2 *tile = data + xf * (long)N + (long)H * 1024 * m_lo + 16 * c;
    float2 *base = tile + (long)256 * m_lo * w; // this wave's quarter
    const unsigned l = 16 * c + col;
    float2 a[4][16], b[4][16];
    const unsigned lo_ld = col + 16u * m_lo * q;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) a[i][j] = (base + (long)m_lo * (64 * i + j))[lo_ld]; // row 16 (4i + q) + j
    RD_SCHED_BARRIER();
#pragma unroll
    for (int i = 0; i < 4; ++i) big_macro16<INV>(a[i], tw_ordered_stage(T, m_lo, 0), tw_ordered_stage(T, m_lo, 1), l, m_lo, 0u, 1u);
    f64w_exchange<false, false>(a, b, Lw, lane);
#pragma unroll
    for (int x = 0; x < 4; ++x) big_macro16<INV>(b[x], tw_ordered_stage(T, m_lo, 2), tw_ordered_stage(T, m_lo, 3), l, m_lo, (unsigned)(q + 4 * x), 16u);
    // b[x][j] = row q + 4x + 16j of this quarter, column col.  Fifth stage: sub-length 256 m_lo, twiddle index l + m_lo * row
    const TwOrdered t4 = tw_ordered_stage(T, m_lo, 4);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int j = 0; j < 16; ++j) X.v[w][16 * q + j][col] = b[x][j];
        __syncthreads();
        float2 v[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int n = 0; n < 4; ++n) v[jj][n] = X.v[n][16 * q + 4 * w + jj][col];
#pragma unroll
        for (int jj = 0; jj < 4; jj += 2) {
            const unsigned row = (unsigned)(q + 4 * x + 16 * (4 * w + jj)), k = l + m_lo * row, kb = k + 16 * m_lo;
            bfly4x2<INV>(v[jj][0], v[jj][1], v[jj][2], v[jj][3], t4.get(1, k), t4.get(2, k), t4.get(3, k),
                         v[jj + 1][0], v[jj + 1][1], v[jj + 1][2], v[jj + 1][3], t4.get(1, kb), t4.get(2, kb), t4.get(3, kb));
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const long r = 256 * n + q + 4 * x + 16 * (4 * w + jj);
                if (vout) { // overlap-save: this was the last pass; 1/N and only the hop valid outputs, packed
                    const long e = (long)H * 1024 * m_lo + (long)m_lo * r + l;
                    if (e < hop) vout[xf * hop + e] = make_float2(mul_rn(v[jj][n].x, scale), mul_rn(v[jj][n].y, scale));
                } else (tile + (long)m_lo * r)[col] = v[jj][n];
            }
        __syncthreads();
    }
}

// gather pass with five stages (4^L points, L >= 9): 1024 source rows N / 1024 apart x 16 source columns per workgroup
template <bool INV>
__global__ __launch_bounds__(256, 2) void fftbig_first5_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, const float2 *__restrict__ T1,
                                                            long in_stride, long ngroups, int L, const float2 *__restrict__ mulH = nullptr)
{
#if REDIO_TILE_PAIR // the pair form (fft_pair.h: pw_first5_tile)
    __shared__ float4 Lx[2 * PW_X5_UNITS];
    {
        const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const long group = f64w_first_tile() >> 2;
        if (group >= ngroups) return;
        const long xf = group >> (2 * (L - 7));
        const unsigned c = (unsigned)(group & ((1u << (2 * (L - 7))) - 1));
        (void)tw;
        if (mulH) pw_first5_tile<INV, true>(in + xf * in_stride, out + xf * (long)(1u << (2 * L)), L, c, lane, w, Lx, mulH, T1);
        else pw_first5_tile<INV, false>(in + xf * in_stride, out + xf * (long)(1u << (2 * L)), L, c, lane, w, Lx, nullptr, T1);
        return;
    }
#endif
    __shared__ __attribute__((aligned(16))) float2 Ls[4 * F64W_REGION];
    __shared__ F5Image X;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long group = f64w_first_tile() >> 2;
    if (group >= ngroups) return;
    float2 *Lw = Ls + w * F64W_REGION;
    const unsigned N = 1u << (2 * L), S = N >> 8, S5 = N >> 10; // source row strides of the four-stage program and of the tile
    const long xf = group >> (2 * (L - 7));
    const unsigned c = (unsigned)(group & ((1u << (2 * (L - 7))) - 1)); // source columns 16c .. 16c + 15 (of N / 1024)
    const int col = lane & 15, q = lane >> 4;
    const float2 *src = in + xf * in_stride + 16 * c + (long)S5 * w; // source rows 4 rho + w: the sub-transform of quarter w
    float2 *dst = out + xf * (long)N;
    float2 a[4][16], b[4][16];
    const unsigned lo_src = col + 4u * S * q;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) a[i][j] = (src + (long)S * (16 * (((j & 3) << 2) | (j >> 2)) + i))[lo_src]; // source row rev4(16 (4i + q) + j)
    if (mulH) { // overlap-save: the spectrum product on the way in (wave-uniform branch)
        const float2 *hsrc = mulH + 16 * c + (long)S5 * w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float2 hv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) hv[j] = (hsrc + (long)S * (16 * (((j & 3) << 2) | (j >> 2)) + i))[lo_src];
            RD_SCHED_BARRIER();
#pragma unroll
            for (int j = 0; j < 16; ++j) a[i][j] = cmul_rn(a[i][j], hv[j]);
        }
    }
    RD_SCHED_BARRIER();
    fftbig_first_stages<INV>(a, b, tw_ordered_stage(T1, 1u, 0), tw_ordered_stage(T1, 1u, 1), tw_ordered_stage(T1, 1u, 2), tw_ordered_stage(T1, 1u, 3), col, lane, Lw);
    // b[x][j] = position col + 16 j of the quarter's 256-point sub-transform, source column 16c + 4q + x.  Fifth stage: sub-length
    // 256, twiddle tw[n * position * N / 1024]; column r of the source is column rev(r) (L - 5 digits) of the working array
    unsigned rc = 0;
    for (int d = 0, cc = c; d < L - 7; ++d, cc >>= 2) rc = (rc << 2) | (cc & 3);
    const unsigned hq = 1u << (2 * (L - 7)), hx = hq << 2;
    const TwOrdered t4 = tw_ordered_stage(T1, 1u, 4); // tw[n * position * N / 1024], neighbouring positions in neighbouring entries
    (void)tw;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
#pragma unroll
        for (int j = 0; j < 16; ++j) X.v[w][16 * q + j][col] = b[x][j];
        __syncthreads();
        float2 v[4][4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int n = 0; n < 4; ++n) v[jj][n] = X.v[n][16 * q + 4 * w + jj][col];
#pragma unroll
        for (int jj = 0; jj < 4; jj += 2) {
            const unsigned k = (unsigned)(col + 16 * (4 * w + jj)), kb = k + 16;
            bfly4x2<INV>(v[jj][0], v[jj][1], v[jj][2], v[jj][3], t4.get(1, k), t4.get(2, k), t4.get(3, k),
                         v[jj + 1][0], v[jj + 1][1], v[jj + 1][2], v[jj + 1][3], t4.get(1, kb), t4.get(2, kb), t4.get(3, kb));
        }
        float2 *colbase = dst + 1024l * (hx * x + hq * q + rc); // the 1024 working positions of source column 16c + 4q + x
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
            for (int n = 0; n < 4; ++n) (colbase + 256 * n + 16 * (4 * w + jj))[col] = v[jj][n];
        __syncthreads();
    }
}

// the last LG = 1, 2 or 3 stages: G = 4^LG rows, m_lo = N / G apart; a wave takes 4096 / G neighbouring columns
template <bool INV, int LG>
__global__ __launch_bounds__(256, 2) void fftbig_last_kernel(float2 *data, const float2 *__restrict__ tw, const float2 *__restrict__ T, long ntiles, int lgN,
                                                          float2 *__restrict__ vout = nullptr, long hop = 0, float scale = 1.0f, int rev = 0)
{
    constexpr int G = 1 << (2 * LG), CPT = 4096 / G; // rows, columns per tile
    constexpr int GG = G < 16 ? G : 16, NG = G / GG; // row g = 16 d2 + j lives in a[.][d2][j]
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = (long)(rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x) * 4 + w;
    if (tile >= ntiles) return;
    const unsigned N = 1u << lgN, m_lo = N / G;
    const long xf = tile >> (lgN - 12);
    const unsigned l0 = (unsigned)(tile & ((1u << (lgN - 12)) - 1)) * CPT;
    float2 *base = data + xf * (long)N + l0;
    float2 a[CPT / 64][NG][GG];
#pragma unroll
    for (int i = 0; i < CPT / 64; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) a[i][g / GG][g % GG] = uniform_ptr(base + (long)m_lo * g + 64 * i)[(unsigned)lane];
    RD_SCHED_BARRIER();
#pragma unroll
    for (int i = 0; i < CPT / 64; ++i) {
        const unsigned l = l0 + 64 * i + lane;
        if constexpr (LG == 1) {
            bfly4<INV>(a[i][0][0], a[i][0][1], a[i][0][2], a[i][0][3], tw[l], tw[2 * l], tw[3 * l]);
        } else if constexpr (LG == 2) {
            big_macro16<INV>(a[i][0], TwGather{tw, 4u}, TwGather{tw, 1u}, l, m_lo, 0u, 1u);
        } else {
#pragma unroll
            for (int d2 = 0; d2 < 4; ++d2) {
                big_macro16<INV>(a[i][d2], tw_ordered_stage(T, m_lo, 0), tw_ordered_stage(T, m_lo, 1), l, m_lo, 0u, 1u);
                RD_SCHED_BARRIER(); // keeps the other groups' twiddle loads from being hoisted here (64 points are live)
            }
#pragma unroll
            for (int jj = 0; jj < 16; jj += 2) {
                const unsigned k = l + m_lo * jj, kb = k + m_lo;
                bfly4x2<INV>(a[i][0][jj], a[i][1][jj], a[i][2][jj], a[i][3][jj], tw[k], tw[2 * k], tw[3 * k],
                             a[i][0][jj + 1], a[i][1][jj + 1], a[i][2][jj + 1], a[i][3][jj + 1], tw[kb], tw[2 * kb], tw[3 * kb]);
                if ((jj & 6) == 6) RD_SCHED_BARRIER();
            }
        }
    }
#pragma unroll
    for (int i = 0; i < CPT / 64; ++i)
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (vout) { // overlap-save: 1/N and only the hop valid outputs, packed
                const long e = (long)m_lo * g + l0 + 64 * i + lane;
                const float2 v = a[i][g / GG][g % GG];
                if (e < hop) vout[xf * hop + e] = make_float2(mul_rn(v.x, scale), mul_rn(v.y, scale));
            } else uniform_ptr(base + (long)m_lo * g + 64 * i)[(unsigned)lane] = a[i][g / GG][g % GG];
        }
}

// N = 2 * 4^L (32768 ... 8388608): kissfft runs the radix-2 stage first.  The gather pass takes the three stages on
// (b0, d1, d2) = 32 rows, 4^L * 2 / 32 apart in the source, for 64 neighbouring source columns per wave, entirely in registers
// (the first phase of fft2k_wave_regs), and writes each column's 32 results as one 256-byte run of the working order; the radix-4
// stages that remain go through fftbig_mid_kernel / fftbig_last_kernel with rows 32, 8192, ... apart.
template <bool INV>
__global__ __launch_bounds__(256) void fftbig_first2_kernel(const float2 *in, float2 *out, const float2 *__restrict__ tw, long in_stride,
                                                            long ntiles, int lgN, const float2 *__restrict__ mulH = nullptr)
{
    __shared__ float2 Ls2[4 * 64 * 17];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long tile = (long)blockIdx.x * 4 + w;
    if (tile >= ntiles) return;
    const unsigned N = 1u << lgN, S = N >> 5; // S: source row stride
    const int nd = (lgN - 5) / 2;             // base-4 digits of a source column index
    const long xf = tile >> (lgN - 11);
    const unsigned c = (unsigned)(tile & ((1u << (lgN - 11)) - 1)); // source columns 64c .. 64c + 63
    const float2 *src = in + xf * in_stride + 64 * c;
    float2 *dst = out + xf * (long)N;
    float2 a[4][8]; // [d2][b0 + 2 d1]
#pragma unroll
    for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
        for (int j = 0; j < 8; ++j) a[d2][j] = (src + (long)S * (16 * (j & 1) + 4 * (j >> 1) + d2))[(unsigned)lane];
    if (mulH) { // overlap-save: the spectrum product on the way in
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[d2][j] = cmul_rn(a[d2][j], (mulH + 64 * c + (long)S * (16 * (j & 1) + 4 * (j >> 1) + d2))[(unsigned)lane]);
    }
    const float2 w0 = tw[0u], w1 = tw[N >> 3], w2 = tw[2 * (N >> 3)], w3 = tw[3 * (N >> 3)];
    RD_SCHED_BARRIER();
#pragma unroll
    for (int d2 = 0; d2 < 4; ++d2) {
#pragma unroll
        for (int d1 = 0; d1 < 4; ++d1) bfly2(a[d2][2 * d1], a[d2][2 * d1 + 1], w0);
        bfly4x2<INV>(a[d2][0], a[d2][2], a[d2][4], a[d2][6], w0, w0, w0, a[d2][1], a[d2][3], a[d2][5], a[d2][7], w1, w2, w3);
    }
    const unsigned fs = N >> 5; // stage on d2: sub-length 8, k = b0 + 2 d1
#pragma unroll
    for (int k = 0; k < 8; k += 2)
        bfly4x2<INV>(a[0][k], a[1][k], a[2][k], a[3][k], tw[k * fs], tw[2 * k * fs], tw[3 * k * fs],
                     a[0][k + 1], a[1][k + 1], a[2][k + 1], a[3][k + 1], tw[(k + 1) * fs], tw[2 * (k + 1) * fs], tw[3 * (k + 1) * fs]);
    // source column r = 64 c + run is column h = digit reversal of r (nd digits) of the working array: 32 rows = one 256-byte run per
    // column, and the runs of a wave's 64 columns lie far apart (their low digits become the high ones).  A lane storing its own run
    // would put 64 separate 16-byte pieces into every store instruction; the results go through a wave-private LDS image instead
    // (two rounds of 16 rows), and each instruction writes eight complete 128-byte half-runs (eight lanes each).
    unsigned hc = 0;
    for (int d = 0, cc = c; d < nd - 3; ++d, cc >>= 2) hc = (hc << 2) | (cc & 3);
    float2 *Lw = Ls2 + w * (64 * 17);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int k = 0; k < 16; ++k) Lw[lane * 17 + k] = a[2 * half + (k >> 3)][k & 7]; // row b0 + 2 d1 + 8 d2 = 16 half + k
        wave_lds_fence();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned run = 8u * i + (lane >> 3), f = lane & 7;
            const unsigned rev = ((run & 3) << 4) | (run & 12) | (run >> 4); // the run's three digits reversed
            const unsigned h = (rev << (2 * (nd - 3))) | hc;
            const float2 v0 = Lw[run * 17 + 2 * f], v1 = Lw[run * 17 + 2 * f + 1];
            reinterpret_cast<float4 *>(dst + 32l * h + 16 * half)[f] = make_float4(v0.x, v0.y, v1.x, v1.y);
        }
        wave_lds_fence();
    }
}

// the passes after the gather pass: rows 2^lm apart, `left` radix-4 stages to go
// overlap-save with 8192-point blocks: four wavefronts, the ovsave16k scheme on the 2048-point program.  The forward last stage
// runs in rounds over d5; thread (w, lane) of a round owns k = lane + 64 slot + 128 w + 512 r (slot = 0, 1), and sample
// n = k + 2048 rr of the spectrum belongs to inverse wave n & 3 at position n >> 2 = lane' + 64 (d2' + 4 d1' + 16 b0') with
// lane' = (lane >> 2) + 16 slot + 32 (w & 1), d2' = (w >> 1) + 2 (r & 1), d1' = (r >> 1) + 2 (rr & 1), b0' = rr >> 1.
constexpr int OV8W_YS = 528; // stride of one inverse wave's image of a round (8 registers x 64 lanes, padded)
__global__ __launch_bounds__(256) void ovsave8k_wave_kernel(const float2 *__restrict__ x, long hop, const float2 *__restrict__ Tf,
                                                            const float2 *__restrict__ Ti, const float2 *__restrict__ Hc,
                                                            float2 *__restrict__ out, float scale)
{
    __shared__ float2 Ls[4 * F4W_REGION];
    static_assert(2048 + 4 * OV8W_YS <= 4 * F4W_REGION && 4096 <= 4 * F4W_REGION, "the shared images live where the private ones were");
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float2 *src = x + (long)blockIdx.x * hop + w;
    float2 *dst = out + (long)blockIdx.x * hop;
    float2 *X = Ls, *Y = Ls + 2048, *Lw = Ls + w * F4W_REGION;
    float2 a[4][8], b[2][16];
    if (REDIO_F16K_LDS_DEAL) f8k_deal_load(a, b, x + (long)blockIdx.x * hop, Ls, w, lane);
    else {
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[d2][j] = (src + 4 * 64 * (d2 + 4 * (j >> 1) + 16 * (j & 1)))[4u * lane];
        RD_SCHED_BARRIER();
    }
    fft2k_wave_regs<false>(a, b, TwProgram<2048, 2>{Tf}, Lw, lane);
    const TwOrdered lf = tw_ordered_stage(Tf, 2u, 5);
#pragma unroll
    for (int r = 0; r < 4; ++r) { // d5 = r
        __syncthreads(); // private images / the previous round's images are no longer read
#pragma unroll
        for (int slot = 0; slot < 2; ++slot)
#pragma unroll
            for (int d4 = 0; d4 < 4; ++d4) X[512 * w + 64 * (slot + 2 * d4) + lane] = b[slot][d4 + 4 * r];
        __syncthreads();
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            const unsigned k = lane + 64u * slot + 128u * w + 512u * r;
            float2 f[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = X[512 * q + 64 * (slot + 2 * w) + lane];
            bfly4<false>(f[0], f[1], f[2], f[3], lf.get(1, k), lf.get(2, k), lf.get(3, k));
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                Y[OV8W_YS * (lane & 3) + 64 * ((w >> 1) + 2 * (rr & 1) + 4 * (rr >> 1)) + (lane >> 2) + 16 * slot + 32 * (w & 1)] =
                    cmul_rn(f[rr], Hc[k + 2048u * rr]);
        }
        __syncthreads();
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8) { // inverse sub-transform w: register d2' = (i8 & 1) + 2 (r & 1), d1' = (r >> 1) + 2 ((i8 >> 1) & 1), b0' = i8 >> 2
            a[(i8 & 1) + 2 * (r & 1)][(i8 >> 2) + 2 * ((r >> 1) + 2 * ((i8 >> 1) & 1))] = Y[OV8W_YS * w + 64 * i8 + lane];
        }
    }
    __syncthreads(); // the images are free again
    int lane_i = lane;
    asm volatile("" : "+v"(lane_i)); // fresh twiddle offsets for the inverse
    fft2k_wave_regs<true>(a, b, TwProgram<2048, 2>{Ti}, Lw, lane_i);
    const TwOrdered li = tw_ordered_stage(Ti, 2u, 5);
#pragma unroll
    for (int r = 0; r < 2; ++r) { // slot = r, as in fft8k_wave_kernel
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) X[1024 * w + 64 * j + lane_i] = b[r][j];
        __syncthreads();
#pragma unroll
        for (int d4 = 0; d4 < 4; ++d4) {
            const int jj = d4 + 4 * w;
            const unsigned k = 512u * w + 128u * d4 + 64u * r + lane_i;
            float2 f[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) f[q] = X[1024 * q + 64 * jj + lane_i];
            bfly4<true>(f[0], f[1], f[2], f[3], li.get(1, k), li.get(2, k), li.get(3, k));
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                if ((long)(k + 2048u * rr) < hop) dst[k + 2048u * rr] = make_float2(mul_rn(f[rr].x, scale), mul_rn(f[rr].y, scale));
        }
    }
}

// ---- any radix-2/3/4/5 size above 16384: tile passes with a run-time stage list -------------------------------------
// The same decomposition as the power-of-two passes, table-driven: the stages are cut (innermost first) into groups whose
// radices multiply to G <= 256 rows; a 256-thread workgroup takes G rows x 16 neighbouring columns into LDS, runs the group's
// stages down the columns and writes the tile back.  The first group gathers the digit-reversed input (rows N / G apart in the
// source, 16 neighbouring source columns) and writes every column as one run of G positions of the working order; the later
// groups work in place on rows m_lo apart.  One pass per group instead of one launch per stage.
struct FtpMagic { unsigned ml[12]; unsigned g; }; // ceil(2^32 / d) of the group's sub-lengths in rows and of G: exact quotients of numbers below 2^16
constexpr int FTP_MAX_POINTS = 4096; // points per tile (8192-point tiles with 512-byte row pieces were measured: slower, as were 2048 and 1024)
// one stage of a tile pass for the radix P: a thread's butterflies (at most IT of them) in two sweeps -- first every butterfly's twiddle
// loads (global memory, L2), then the butterflies -- so that a stage waits for its twiddles once, not once per butterfly (round 3; the
// SQ counters of the one-sweep form: 73 % of the wave cycles waiting).  Requesting the first three stages' twiddles before the tile's
// samples instead (they depend on the thread and the stage only) was measured as well: slower, 65 against 79 GS/s at 20000 points.
// kbase < 0: the gather pass, whose columns are whole sub-transforms (the twiddle index has no column part).
template <bool INV, int P>
__device__ __forceinline__ void ftp_stage(float2 *L, int LD, int lcw, int CW, int ncol, int nb, int ml, unsigned magic_ml, int m, int m_lo, int kbase,
                                          const float2 *__restrict__ T, float2 epi1, float2 epi2, int tid)
{
    constexpr int IT = (FTP_MAX_POINTS / P + 255) / 256;
    float2 tw[IT][P - 1];
    int off[IT];
    const int rs = ml * LD;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int e = tid + 256 * it;
        const int bf = e >> lcw, col = e & (CW - 1);
        const bool on = e < nb * CW && col < ncol;
        const int blk = ml == 1 ? bf : (int)__umulhi((unsigned)bf, magic_ml), kl = bf - blk * ml;
        off[it] = on ? (blk * P * ml + kl) * LD + col : -1;
        const int k = on ? (kbase < 0 ? 0 : kbase + col) + m_lo * kl : 0; // e mod m
#pragma unroll
        for (int n = 0; n < P - 1; ++n) tw[it][n] = T[n * m + k];
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        if (off[it] < 0) continue;
        float2 *q = L + off[it];
        if constexpr (P == 2) {
            float2 a0 = q[0], a1 = q[rs];
            bfly2(a0, a1, tw[it][0]);
            q[0] = a0; q[rs] = a1;
        } else if constexpr (P == 3) {
            float2 a0 = q[0], a1 = q[rs], a2 = q[2 * rs];
            bfly3(a0, a1, a2, tw[it][0], tw[it][1], epi1);
            q[0] = a0; q[rs] = a1; q[2 * rs] = a2;
        } else if constexpr (P == 4) {
            float2 a0 = q[0], a1 = q[rs], a2 = q[2 * rs], a3 = q[3 * rs];
            bfly4<INV>(a0, a1, a2, a3, tw[it][0], tw[it][1], tw[it][2]);
            q[0] = a0; q[rs] = a1; q[2 * rs] = a2; q[3 * rs] = a3;
        } else {
            float2 a0 = q[0], a1 = q[rs], a2 = q[2 * rs], a3 = q[3 * rs], a4 = q[4 * rs];
            bfly5(a0, a1, a2, a3, a4, tw[it][0], tw[it][1], tw[it][2], tw[it][3], epi1, epi2);
            q[0] = a0; q[rs] = a1; q[2 * rs] = a2; q[3 * rs] = a3; q[4 * rs] = a4;
        }
    }
}
template <bool INV>
__global__ __launch_bounds__(256) void fft_tile_pass_kernel(FftPlanDev p, const float2 *src, float2 *dst, int s_hi, int s_lo, int G, int m_lo,
                                                            int first, long in_stride, int tiles_per_xf, int lcw, FtpMagic mg, int rev)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *L = reinterpret_cast<float2 *>(smem);
    const int tid = threadIdx.x, N = p.nfft, CW = 1 << lcw, LD = CW + 1; // CW columns per tile (a power of two), padded rows
    int *rowsrc = reinterpret_cast<int *>(L + G * LD), *hcol = rowsrc + G; // per-tile index tables (first pass)
    const unsigned bid = rev ? gridDim.x - 1 - blockIdx.x : blockIdx.x; // rev: this pass walks the batch against the pass before it (launch_fftbig)
    const long xf = bid / tiles_per_xf;
    const int tt = bid - (int)(xf * tiles_per_xf);
    // columns of this tile: first pass: source columns r = CW tt + col (r < N / G); later: l = CW c + col of block h
    const int ncolblk = first ? 0 : (m_lo + CW - 1) >> lcw;
    const int c = first ? tt : tt % ncolblk, h = first ? 0 : tt / ncolblk;
    const int width = first ? N / G : m_lo;
    const int ncol = width - CW * c < CW ? width - CW * c : CW;
    const float2 *in = first ? src + xf * in_stride : dst + xf * (long)N;
    float2 *out = dst + xf * (long)N;
    if (first) {
        for (int g = tid; g < G; g += 256) { // row g of the working order = source row with the group's digits reversed
            int rs = 0;
            for (int s = s_hi; s >= s_lo; --s) rs += ((g / p.st[s].m) % p.st[s].p) * p.st[s].fstride;
            rowsrc[g] = rs;
        }
        for (int col = tid; col < ncol; col += 256) { // source column r -> column of the working order (the outer stages' digits)
            const int r = CW * c + col;
            int hh = 0;
            for (int s = s_lo - 1; s >= 0; --s) hh += ((r / p.st[s].fstride) % p.st[s].p) * (p.st[s].m / G);
            hcol[col] = hh;
        }
        __syncthreads();
    }
    for (int e0 = tid; e0 < G * CW; e0 += 256 * 8) { // eight loads in flight per thread
        float2 v[8];
        int at[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + 256 * u, g = e >> lcw, col = e & (CW - 1);
            const bool on = e < G * CW && col < ncol;
            at[u] = on ? g * LD + col : -1;
            const int idx = !on ? 0 : first ? rowsrc[g] + CW * c + col : h * G * m_lo + g * m_lo + CW * c + col;
            v[u] = in[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (at[u] >= 0) L[at[u]] = v[u];
    }
    __syncthreads();
    for (int s = s_hi; s >= s_lo; --s) { // innermost stage of the group first
        const int P = p.st[s].p, m = p.st[s].m, fs = p.st[s].fstride, ml = m / m_lo; // ml: sub-length in rows
        const int nb = G / P;
        int toff = 0; // the stage's block of the stage-ordered twiddle copy: T[(n - 1) m + k] = tw[n k fstride]
        for (int u = 0; u < s; ++u) toff += (p.st[u].p - 1) * p.st[u].m;
        const float2 *__restrict__ T = p.tw_pass + toff;
        const int kbase = first ? -1 : CW * c;
        const unsigned mgl = mg.ml[s_hi - s];
        if (P == 2) ftp_stage<INV, 2>(L, LD, lcw, CW, ncol, nb, ml, mgl, m, m_lo, kbase, T, float2{}, float2{}, tid);
        else if (P == 3) ftp_stage<INV, 3>(L, LD, lcw, CW, ncol, nb, ml, mgl, m, m_lo, kbase, T, p.tw[fs * m], float2{}, tid);
        else if (P == 4) ftp_stage<INV, 4>(L, LD, lcw, CW, ncol, nb, ml, mgl, m, m_lo, kbase, T, float2{}, float2{}, tid);
        else ftp_stage<INV, 5>(L, LD, lcw, CW, ncol, nb, ml, mgl, m, m_lo, kbase, T, p.tw[fs * m], p.tw[fs * 2 * m], tid);
        __syncthreads();
    }
    if (first) { // source column r -> column hh of the working order (the outer stages' digits), G contiguous positions each
        for (int e = tid; e < G * CW; e += 256) {
            const int col = (int)__umulhi((unsigned)e, mg.g), g = e - col * G;
            if (col >= ncol) continue;
            out[(long)hcol[col] * G + g] = L[g * LD + col];
        }
    } else {
        for (int e0 = tid; e0 < G * CW; e0 += 256 * 8) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + 256 * u, g = e >> lcw, col = e & (CW - 1);
                v[u] = e < G * CW ? L[g * LD + col] : float2{};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int e = e0 + 256 * u, g = e >> lcw, col = e & (CW - 1);
                if (e < G * CW && col < ncol) out[h * G * m_lo + g * m_lo + CW * c + col] = v[u];
            }
        }
    }
}

// cut the stages (innermost first) into the fewest groups of at most 256 rows, then as evenly as that count allows
static int fft_tile_groups(const FftPlanDev &p, int limit, int *lo_of /* [nstages]: s_lo of the group starting at s_hi */)
{
    int n = 0;
    for (int s_hi = p.nstages - 1; s_hi >= 0; ++n) {
        int G = 1, s_lo = s_hi + 1;
        while (s_lo - 1 >= 0 && G * p.st[s_lo - 1].p <= limit) { --s_lo; G *= p.st[s_lo].p; }
        if (s_lo > s_hi) return 1 << 20; // a single radix above the limit
        if (lo_of) lo_of[s_hi] = s_lo;
        s_hi = s_lo - 1;
    }
    return n;
}
template <bool INV>
static hipError_t launch_fft_tile_passes(const FftPlanDev &p, const float2 *in, float2 *out, long nbatch, long in_stride, hipStream_t s)
{
    int lo_of[FFT_MAX_STAGES];
    const int npass = fft_tile_groups(p, 256, nullptr);
    int limit = 256;
    for (int cand = 8; cand < 256; ++cand)
        if (fft_tile_groups(p, cand, nullptr) == npass) { limit = cand; break; }
    fft_tile_groups(p, limit, lo_of);
    int m_lo = 1, rev = 0;
    bool first = true;
    for (int s_hi = p.nstages - 1; s_hi >= 0;) {
        const int s_lo = lo_of[s_hi];
        int G = 1;
        for (int t = s_lo; t <= s_hi; ++t) G *= p.st[t].p;
        int lcw = 4; // columns per tile: 16 ... 256, about 4096 points per tile
        while (lcw < 8 && (G << (lcw + 1)) <= FTP_MAX_POINTS) ++lcw;
        const int CW = 1 << lcw, width = first ? p.nfft / G : m_lo;
        const int colblk = (width + CW - 1) / CW;
        const int tiles = first ? colblk : colblk * (p.nfft / (G * m_lo));
        const size_t lds = (size_t)G * (CW + 1) * sizeof(float2) + (size_t)(G + CW) * sizeof(int);
        FtpMagic mg = {};
        if (s_hi - s_lo >= 12) return hipErrorNotSupported;
        for (int t = s_hi; t >= s_lo; --t) {
            const unsigned ml = (unsigned)(p.st[t].m / m_lo);
            mg.ml[s_hi - t] = ml > 1 ? (unsigned)((0x100000000ull + ml - 1) / ml) : 0u;
        }
        mg.g = G > 1 ? (unsigned)((0x100000000ull + G - 1) / G) : 0u;
        hipLaunchKernelGGL(fft_tile_pass_kernel<INV>, dim3((unsigned)(nbatch * tiles)), dim3(256), lds, s, p, in, out, s_hi, s_lo, G, m_lo,
                           first ? 1 : 0, in_stride, tiles, lcw, mg, rev);
        rev ^= 1;
        m_lo *= G;
        s_hi = s_lo - 1;
        first = false;
    }
    return hipGetLastError();
}

// Plan B: sizes for which five-stage passes save a pass or end on a cheaper last pass.  first = stages of the gather pass (2: the
// radix-2 stage + two radix-4 stages of 2 * 4^L), mid = stages of each in-place pass, last = trailing register-only stages.
//   2^15: 3 + 5            (two passes instead of three)      2^18: 4 + 5, 2^20: 5 + 5   (two instead of three)
//   2^19: 3 + 5 + 2  (three either way; shorter last pass; 2^22 as 5 + 5 + 1 and 2^24 as 5 + 5 + 2 measured slower)     2^23: 3 + 5 + 4   (three instead of four)
// Its in-place passes read their own ordered twiddle copies, stored behind plan A's tables.
constexpr int FFTBIG_MID4_ELEMS = 340; // float2 per unit of row stride in the interleaved copy of a four-stage pass: 85 entries x 4
struct BigPlanB { int first, nmid, mid[2], last; };
static bool fftbig_plan_b(int lgN, BigPlanB &p)
{
    switch (lgN) {
    case 15: p = {2, 1, {5, 0}, 0}; return true;
    case 18: p = {4, 1, {5, 0}, 0}; return true;
    case 19: p = {2, 1, {5, 0}, 2}; return true;
    case 20: p = {5, 1, {5, 0}, 0}; return true;
    // (2^22 as 5 + 5 + 1 was the faster plan until the four-stage pass got its interleaved twiddle copy; re-measured in round 4 with the pair-form
    // five-stage passes: 0.720 against 0.705 ms per 2^26 points, and 2^24 as 5 + 5 + 2 0.879 against 0.690 -- profiles/r04_plan_b_large_sizes_null.txt)
    case 22: if (measure_env("REDIO_FFT_PLAN_B_22")) { p = {5, 1, {5, 0}, 1}; return true; } return false; // measurement: 5 + 5 + 1 against 4 + 4 + 3
    case 24: if (measure_env("REDIO_FFT_PLAN_B_24")) { p = {5, 1, {5, 0}, 2}; return true; } return false; // measurement: 5 + 5 + 2 against 4 + 4 + 4
    case 23: p = {2, 2, {5, 4}, 0}; return true;
    default: return false;
    }
}
static int fftbig_plan_b_lm0(const BigPlanB &p) { return p.first == 2 ? 5 : p.first == 4 ? 8 : 10; }
static size_t fftbig_five_elems(int lgN)
{
    BigPlanB p;
    if (!fftbig_plan_b(lgN, p)) return 0;
    size_t total = 0;
    int lm = fftbig_plan_b_lm0(p);
    for (int i = 0; i < p.nmid; ++i) { total += (size_t)(p.mid[i] == 5 ? 1023 : FFTBIG_MID4_ELEMS) << lm; lm += 2 * p.mid[i]; }
    return total;
}
// every 4^L size: the gather pass's ordered copy (sub-lengths 1, 4, 16, 64, 256), at the very end of the tables
static size_t fftbig_first_elems(int lgN) { return (lgN & 1) ? 0 : 1023; }
static bool fftbig_size(int nfft) { return nfft >= (1 << 15) && nfft <= (1 << 24) && (nfft & (nfft - 1)) == 0 && nfft != 16384; }
// Plan G (round 4; pair builds): 2^15 and 2^17 points run the FOUR-stage gather pass G128 (fft_big_core.h) and ONE in-place pass on rows
// 128 apart -- four stages (2^15) or five (2^17) -- instead of the three-stage gather pass plus a five-stage pass / two more passes.
// Its tables sit at the very end of the plan's tables: the G128 copy (128 entries), then the in-place pass's ordered copy.
static bool fftbig_plan_g(int lgN)
{
    if (!REDIO_TILE_PAIR || !(lgN & 1) || lgN < 15 || lgN > 23) return false;
    // per 2^26 points (profiles/r04_plan_g_large_sizes_ab.txt): 2^19 0.661 -> 0.581 ms, 2^23 0.690 -> 0.667 ms; 2^21 (four-stage pass + three
    // register-only stages) 0.611 -> 0.639 ms: that size keeps the three-stage gather pass and two four-stage passes
    if (lgN == 21 && !measure_env("REDIO_FFT_PLAN_G_21")) return false;
    if (lgN >= 19 && measure_env("REDIO_FFT_NO_PLAN_G")) return false; // measurement builds: 2^19 / 2^23 through the three-stage gather pass as before
    return true;
}
// after G128 (rows 128 apart): `left` radix-4 stages to go = 4 (2^15: one four-stage pass), 5 (2^17: one five-stage pass), 6 (2^19: four-stage
// pass + two register-only stages), 7 (2^21: four + three), 8 (2^23: two four-stage passes)
// Plan G5 (round 4): 2^19 points as TWO passes -- the five-stage gather pass G512 and ONE five-stage in-place pass on rows 512 apart -- for the
// stand-alone transform (the overlap-save entry, with its spectrum product and masked store, keeps plan G).  Its tables (G128's copy extended by
// the sub-length-128 stage, then the in-place pass's ordered copy) follow plan G's at the very end.
static bool fftbig_plan_g5(int lgN) { return REDIO_TILE_PAIR && lgN == 19 && !measure_env("REDIO_FFT_NO_PLAN_G5"); }
static size_t fftbig_g5_elems(int lgN) { return fftbig_plan_g5(lgN) ? (size_t)PW_G5_TABLE + ((size_t)1023 << 9) : 0; }
static size_t fftbig_g_elems(int lgN)
{
    if (!fftbig_plan_g(lgN)) return 0;
    const int left = (lgN - 7) / 2;
    size_t n = (size_t)PW_G_TABLE + ((size_t)(left == 5 ? 1023 : FFTBIG_MID4_ELEMS) << 7);
    if (left == 7) n += (size_t)15 << (lgN - 6);
    if (left == 8) n += (size_t)FFTBIG_MID4_ELEMS << 15;
    return n + fftbig_g5_elems(lgN);
}
static void fftbig_after_first(int lgN, int &lm, int &left)
{
    if (lgN & 1) { lm = 5; left = (lgN - 5) / 2; } // 2 * 4^L: the gather pass did the radix-2 stage and two radix-4 stages
    else { lm = 8; left = (lgN - 8) / 2; }
}
size_t fftbig_tables_elems(int nfft)
{
    if (nfft == 4096) return 4095;  // the one-wave 4096-point program: stages of sub-length 1 ... 1024
    if (nfft == 16384) return 16383; // ... and the last stage across the four waves (sub-length 4096)
    if (nfft == 512) return 510;    // fft_p2_kernel<9>: sub-lengths 2 ... 128
    if (nfft == 2048) return 2046;  // the 2048-point program: sub-lengths 2 ... 512
    if (nfft == 8192) return 8190;  // ... and the last stage (sub-length 2048)
    if (!fftbig_size(nfft)) return 0;
    const int lgN = __builtin_ctz((unsigned)nfft);
    int lm, left;
    fftbig_after_first(lgN, lm, left);
    size_t total = 0;
    for (; left >= 4; lm += 8, left -= 4) total += (size_t)FFTBIG_MID4_ELEMS << lm;
    if (left == 3) total += (size_t)15 << (lgN - 6);
    return total + fftbig_five_elems(lgN) + fftbig_first_elems(lgN) + fftbig_g_elems(lgN);
}
__global__ __launch_bounds__(128) void fftbig_g_table_kernel(const float2 *__restrict__ tw, float2 *__restrict__ T, unsigned N)
{
    float2 v;
    pw_g_table_entry(tw, N, (int)threadIdx.x, v);
    T[threadIdx.x] = v;
}
__global__ __launch_bounds__(512) void fftbig_g5_table_kernel(const float2 *__restrict__ tw, float2 *__restrict__ T, unsigned N)
{
    float2 v;
    pw_g5_table_entry(tw, N, (int)threadIdx.x, v);
    T[threadIdx.x] = v;
}
hipError_t fftbig_tables_build(const float2 *tw, float2 *tables, int nfft, hipStream_t s)
{
    if (nfft == 512) {
        hipLaunchKernelGGL(fftbig_tables_kernel, dim3(2), dim3(256), 0, s, tw, tables, 2u, 4, 512u);
        return hipGetLastError();
    }
    if (nfft == 4096 || nfft == 16384 || nfft == 2048 || nfft == 8192) { // a quarter of 8192 / 16384 uses every fourth entry: the same values
        const bool p4 = nfft == 4096 || nfft == 16384;
        hipLaunchKernelGGL(fftbig_tables_kernel, dim3(64), dim3(256), 0, s, tw, tables, p4 ? 1u : 2u, (p4 ? 6 : 5) + (nfft > 4096 ? 1 : 0), (unsigned)nfft);
        return hipGetLastError();
    }
    if (!fftbig_size(nfft)) return hipErrorInvalidValue;
    const int lgN = __builtin_ctz((unsigned)nfft);
    int lm, left;
    fftbig_after_first(lgN, lm, left);
    float2 *T = tables;
    for (; left >= 4; lm += 8, left -= 4) {
#if REDIO_TILE_PAIR // the pair tile program reads the ORDERED copy two neighbouring entries at a time (fft_big_core.h, TwPairOrdered); 255 of the 340 float2 per unit
        hipLaunchKernelGGL(fftbig_tables_kernel, dim3(2048), dim3(256), 0, s, tw, T, 1u << lm, 4, (unsigned)nfft);
#else
        hipLaunchKernelGGL(fftbig_tables_inter_kernel, dim3(2048), dim3(256), 0, s, tw, T, 1u << lm, 4, (unsigned)nfft);
#endif
        T += (size_t)FFTBIG_MID4_ELEMS << lm;
    }
    if (left == 3) { hipLaunchKernelGGL(fftbig_tables_kernel, dim3(2048), dim3(256), 0, s, tw, T, 1u << (lgN - 6), 2, (unsigned)nfft); T += (size_t)15 << (lgN - 6); }
    BigPlanB pb;
    if (fftbig_plan_b(lgN, pb)) {
        int lmb = fftbig_plan_b_lm0(pb);
        for (int i = 0; i < pb.nmid; ++i) {
            if (pb.mid[i] == 5) hipLaunchKernelGGL(fftbig_tables_kernel, dim3(2048), dim3(256), 0, s, tw, T, 1u << lmb, 5, (unsigned)nfft);
#if REDIO_TILE_PAIR
            else hipLaunchKernelGGL(fftbig_tables_kernel, dim3(2048), dim3(256), 0, s, tw, T, 1u << lmb, 4, (unsigned)nfft);
#else
            else hipLaunchKernelGGL(fftbig_tables_inter_kernel, dim3(2048), dim3(256), 0, s, tw, T, 1u << lmb, 4, (unsigned)nfft);
#endif
            T += (size_t)(pb.mid[i] == 5 ? 1023 : FFTBIG_MID4_ELEMS) << lmb;
            lmb += 2 * pb.mid[i];
        }
    }
    if (fftbig_first_elems(lgN)) { hipLaunchKernelGGL(fftbig_tables_kernel, dim3(4), dim3(256), 0, s, tw, T, 1u, 5, (unsigned)nfft); T += fftbig_first_elems(lgN); }
    if (fftbig_plan_g(lgN)) {
        const int left = (lgN - 7) / 2;
        hipLaunchKernelGGL(fftbig_g_table_kernel, dim3(1), dim3(128), 0, s, tw, T, (unsigned)nfft);
        T += PW_G_TABLE;
        hipLaunchKernelGGL(fftbig_tables_kernel, dim3(2048), dim3(256), 0, s, tw, T, 1u << 7, left == 5 ? 5 : 4, (unsigned)nfft);
        T += (size_t)(left == 5 ? 1023 : FFTBIG_MID4_ELEMS) << 7;
        if (left == 7) { hipLaunchKernelGGL(fftbig_tables_kernel, dim3(2048), dim3(256), 0, s, tw, T, 1u << (lgN - 6), 2, (unsigned)nfft); T += (size_t)15 << (lgN - 6); }
        if (left == 8) { hipLaunchKernelGGL(fftbig_tables_kernel, dim3(2048), dim3(256), 0, s, tw, T, 1u << 15, 4, (unsigned)nfft); T += (size_t)FFTBIG_MID4_ELEMS << 15; }
        if (fftbig_plan_g5(lgN)) {
            static_assert(PW_G5_TABLE == 512, "one thread per entry");
            hipLaunchKernelGGL(fftbig_g5_table_kernel, dim3(1), dim3(512), 0, s, tw, T, (unsigned)nfft);
            T += PW_G5_TABLE;
            hipLaunchKernelGGL(fftbig_tables_kernel, dim3(2048), dim3(256), 0, s, tw, T, 1u << 9, 5, (unsigned)nfft);
        }
    }
    return hipGetLastError();
}

// mulH: multiply the input by this spectrum on the way into the first pass; vout: the last pass stores 1/N-scaled outputs
// below hop, packed per block, there instead of in `out` (the two overlap-save steps that would otherwise be passes of their own)
// Consecutive passes walk the batch in opposite directions (round 3): a pass reads what the pass before it wrote, and starts with what that
// pass wrote last -- the part of the intermediate the 256 MB Infinity Cache still holds.  65536 points, second pass alone: 247 -> 208 us per
// 2^26 points; the transform 0.431 -> 0.392 ms (2^28 points: 1.559 -> 1.505).  Running the passes chunk by chunk instead (32 ... 1024 MiB
// of output per chunk, so that the whole intermediate stays cached) was measured too: no better at any size, worse below 256 MiB (tails);
// so were chunk steps with both passes' workgroups alternating in one launch (the overlap-save scheme below): 0.413 against 0.395 ms.
template <bool INV>
static hipError_t launch_fftbig(const float2 *in, float2 *out, const float2 *tw, const float2 *tables, long nbatch, long in_stride, int lgN,
                                hipStream_t s, const float2 *mulH = nullptr, float2 *vout = nullptr, long hop = 0, float scale = 1.0f)
{
    const long ntiles = nbatch << (lgN - 12);
    int rev = 0; // direction of the pass before (the gather pass walks forward)
    const unsigned grid = (unsigned)((ntiles + 3) / 4);
    const float2 *T1 = fftbig_first_elems(lgN) ? tables + (fftbig_tables_elems(1 << lgN) - fftbig_g_elems(lgN) - fftbig_first_elems(lgN)) : nullptr;
    if (fftbig_plan_g5(lgN) && !mulH && !vout) { // G512, then ONE five-stage in-place pass on rows 512 apart
        const float2 *Tg5 = tables + (fftbig_tables_elems(1 << lgN) - fftbig_g5_elems(lgN)), *Tm5 = Tg5 + PW_G5_TABLE;
        const long ngroups = nbatch << (lgN - 14);
        hipLaunchKernelGGL(fftbig_g512_kernel<INV>, dim3((unsigned)ngroups), dim3(256), 0, s, in, out, in_stride, ngroups, lgN, Tg5);
        hipLaunchKernelGGL(fftbig_mid5_kernel<INV>, dim3((unsigned)ngroups), dim3(256), 0, s, out, Tm5, ngroups, lgN, 9, (float2 *)nullptr, 0l, 1.0f, 1);
        return hipGetLastError();
    }
    if (fftbig_plan_g(lgN)) { // G128, then in-place passes on rows 128 (and 32768) apart
        const float2 *Tg = tables + (fftbig_tables_elems(1 << lgN) - fftbig_g_elems(lgN)), *Tm = Tg + PW_G_TABLE;
        const int left = (lgN - 7) / 2;
        if (mulH) hipLaunchKernelGGL((fftbig_g128_kernel<INV, true>), dim3(grid), dim3(256), 0, s, in, out, in_stride, ntiles, lgN, mulH, Tg);
        else hipLaunchKernelGGL((fftbig_g128_kernel<INV, false>), dim3(grid), dim3(256), 0, s, in, out, in_stride, ntiles, lgN, mulH, Tg);
        if (left == 5) {
            hipLaunchKernelGGL(fftbig_mid5_kernel<INV>, dim3((unsigned)(nbatch << (lgN - 14))), dim3(256), 0, s, out, Tm, nbatch << (lgN - 14), lgN, 7, vout, hop, scale, 1);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(fftbig_mid_kernel<INV>, dim3(grid), dim3(256), 0, s, out, Tm, ntiles, lgN, 7, left == 4 ? vout : nullptr, hop, scale, 1);
        const float2 *T2 = Tm + ((size_t)FFTBIG_MID4_ELEMS << 7);
        if (left == 6) hipLaunchKernelGGL((fftbig_last_kernel<INV, 2>), dim3(grid), dim3(256), 0, s, out, tw, T2, ntiles, lgN, vout, hop, scale, 0);
        else if (left == 7) hipLaunchKernelGGL((fftbig_last_kernel<INV, 3>), dim3(grid), dim3(256), 0, s, out, tw, T2, ntiles, lgN, vout, hop, scale, 0);
        else if (left == 8) hipLaunchKernelGGL(fftbig_mid_kernel<INV>, dim3(grid), dim3(256), 0, s, out, T2, ntiles, lgN, 15, vout, hop, scale, 0);
        return hipGetLastError();
    }
    BigPlanB pb;
    if (fftbig_plan_b(lgN, pb)) {
        const float2 *T = tables + (fftbig_tables_elems(1 << lgN) - fftbig_g_elems(lgN) - fftbig_first_elems(lgN) - fftbig_five_elems(lgN));
        const long ngroups = nbatch << (lgN - 14);
        if (pb.first == 2) {
            const long nt2 = nbatch << (lgN - 11);
            hipLaunchKernelGGL(fftbig_first2_kernel<INV>, dim3((unsigned)((nt2 + 3) / 4)), dim3(256), 0, s, in, out, tw, in_stride, nt2, lgN, mulH);
        } else if (pb.first == 4) {
            { if (mulH) hipLaunchKernelGGL((fftbig_first_kernel<INV, true>), dim3(grid), dim3(256), 0, s, in, out, tw, in_stride, ntiles, lgN / 2, mulH, T1); else hipLaunchKernelGGL((fftbig_first_kernel<INV, false>), dim3(grid), dim3(256), 0, s, in, out, tw, in_stride, ntiles, lgN / 2, mulH, T1); }
        } else {
            hipLaunchKernelGGL(fftbig_first5_kernel<INV>, dim3((unsigned)ngroups), dim3(256), 0, s, in, out, tw, T1, in_stride, ngroups, lgN / 2, mulH);
        }
        int lm = fftbig_plan_b_lm0(pb);
        for (int i = 0; i < pb.nmid; ++i) {
            float2 *vo = (i + 1 == pb.nmid && pb.last == 0) ? vout : nullptr; // the last pass of all
            if (pb.mid[i] == 5) {
                hipLaunchKernelGGL(fftbig_mid5_kernel<INV>, dim3((unsigned)ngroups), dim3(256), 0, s, out, T, ngroups, lgN, lm, vo, hop, scale, rev ^= 1);
                T += (size_t)1023 << lm;
            } else {
                hipLaunchKernelGGL(fftbig_mid_kernel<INV>, dim3(grid), dim3(256), 0, s, out, T, ntiles, lgN, lm, vo, hop, scale, rev ^= 1);
                T += (size_t)FFTBIG_MID4_ELEMS << lm;
            }
            lm += 2 * pb.mid[i];
        }
        if (pb.last == 1) hipLaunchKernelGGL((fftbig_last_kernel<INV, 1>), dim3(grid), dim3(256), 0, s, out, tw, T, ntiles, lgN, vout, hop, scale, rev ^= 1);
        else if (pb.last == 2) hipLaunchKernelGGL((fftbig_last_kernel<INV, 2>), dim3(grid), dim3(256), 0, s, out, tw, T, ntiles, lgN, vout, hop, scale, rev ^= 1);
        return hipGetLastError();
    }
    if (lgN & 1) {
        const long nt2 = nbatch << (lgN - 11);
        hipLaunchKernelGGL(fftbig_first2_kernel<INV>, dim3((unsigned)((nt2 + 3) / 4)), dim3(256), 0, s, in, out, tw, in_stride, nt2, lgN, mulH);
    } else {
        { if (mulH) hipLaunchKernelGGL((fftbig_first_kernel<INV, true>), dim3(grid), dim3(256), 0, s, in, out, tw, in_stride, ntiles, lgN / 2, mulH, T1); else hipLaunchKernelGGL((fftbig_first_kernel<INV, false>), dim3(grid), dim3(256), 0, s, in, out, tw, in_stride, ntiles, lgN / 2, mulH, T1); }
    }
    int lm, left;
    fftbig_after_first(lgN, lm, left);
    const float2 *T = tables;
    for (; left >= 4; lm += 8, left -= 4) {
        float2 *vo = left == 4 ? vout : nullptr; // the last pass of all
        hipLaunchKernelGGL(fftbig_mid_kernel<INV>, dim3(grid), dim3(256), 0, s, out, T, ntiles, lgN, lm, vo, hop, scale, rev ^= 1);
        T += (size_t)FFTBIG_MID4_ELEMS << lm;
    }
    switch (left) {
    case 0: break;
    case 1: hipLaunchKernelGGL((fftbig_last_kernel<INV, 1>), dim3(grid), dim3(256), 0, s, out, tw, T, ntiles, lgN, vout, hop, scale, rev ^= 1); break;
    case 2: hipLaunchKernelGGL((fftbig_last_kernel<INV, 2>), dim3(grid), dim3(256), 0, s, out, tw, T, ntiles, lgN, vout, hop, scale, rev ^= 1); break;
    default: hipLaunchKernelGGL((fftbig_last_kernel<INV, 3>), dim3(grid), dim3(256), 0, s, out, tw, T, ntiles, lgN, vout, hop, scale, rev ^= 1); break;
    }
    return hipGetLastError();
}

// overlap-save with a block size of the multi-pass family (32768, 131072 ...; 65536 has the three-pass scheme below): forward
// transform of the hop-strided blocks into `a`, then the inverse from `a` through `b` with the spectrum product folded into its
// first pass and the scaled copy of the valid outputs into its last -- six passes instead of eight
bool ovsave_big_size(int nfft) { return fftbig_size(nfft) && nfft != F64K_N; }
hipError_t launch_ovsave_big(const FftPlanDev &fw, const FftPlanDev &bw, const float2 *x, long hop, float2 *a, float2 *b, const float2 *Hc,
                             float2 *out, long nblk, float scale, hipStream_t s)
{
    if (!ovsave_big_size(fw.nfft) || fw.nfft != bw.nfft || !fw.tw_pass || !bw.tw_pass) return hipErrorInvalidValue;
    const int lgN = __builtin_ctz((unsigned)fw.nfft);
    if (lgN == 15 && fftbig_plan_g(lgN)) { // three passes: G128 forward; [forward in-place pass x conj H x inverse G128] on one tile; inverse in-place pass
        const size_t goff = fftbig_tables_elems(1 << lgN) - fftbig_g_elems(lgN);
        const float2 *Tgf = fw.tw_pass + goff, *Tgi = bw.tw_pass + goff;
        const long ntiles = nblk << 3;
        const unsigned grid = (unsigned)((ntiles + 3) / 4);
        hipLaunchKernelGGL((fftbig_g128_kernel<false, false>), dim3(grid), dim3(256), 0, s, x, a, hop, ntiles, lgN, (const float2 *)nullptr, Tgf);
        hipLaunchKernelGGL(ovsave32k_mid_kernel, dim3(grid), dim3(256), 0, s, a, b, Tgf + PW_G_TABLE, Tgi, Hc, ntiles);
        hipLaunchKernelGGL(fftbig_mid_kernel<true>, dim3(grid), dim3(256), 0, s, b, Tgi + PW_G_TABLE, ntiles, lgN, 7, out, hop, scale, 0);
        return hipGetLastError();
    }
    hipError_t e = launch_fftbig<false>(x, a, fw.tw, fw.tw_pass, nblk, hop, lgN, s);
    if (e != hipSuccess) return e;
    return launch_fftbig<true>(a, b, bw.tw, bw.tw_pass, nblk, (long)fw.nfft, lgN, s, Hc, out, hop, scale);
}

// the last pass of one chunk and the first pass of the NEXT chunk in one launch: the two are independent (b -> out, x -> a; the
// middle pass of the earlier chunk has finished with a), so their load and store phases overlap and a chunk costs two launches
// instead of three.  Workgroups alternate between the two tile programs.
__global__ __launch_bounds__(256, 2) void ovsave64k_last_first_kernel(const float2 *__restrict__ b_in, float2 *__restrict__ out,
                                                                   const float2 *__restrict__ Ti, long hop, float scale, long ntiles_last,
                                                                   const float2 *__restrict__ x_next, float2 *__restrict__ a_out,
                                                                   const float2 *__restrict__ tw_f, long ntiles_first, const float2 *__restrict__ T1)
{
    __shared__ __attribute__((aligned(16))) float2 Ls[4 * F64W_REGION];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *Lw = Ls + w * F64W_REGION;
    // workgroup b: kind = b & 1 while both kinds have work left, the longer kind takes the rest
    const long nb_last = (ntiles_last + 3) / 4, nb_first = (ntiles_first + 3) / 4, both = 2 * (nb_last < nb_first ? nb_last : nb_first);
    const long b = blockIdx.x;
    bool first;
    long wg;
    if (b < both) { first = (b & 1) != 0; wg = b >> 1; }
    else { first = nb_first > nb_last; wg = (both >> 1) + (b - both); }
    const long tile = wg * 4 + w;
    if (first) {
        if (tile >= ntiles_first) return;
        fftbig_first_tile<false, false>(x_next + (tile >> 4) * hop, a_out + (tile >> 4) * (long)F64K_N, tw_f, 8, (unsigned)(tile & 15), lane, Lw, nullptr, T1);
    } else {
        if (tile >= ntiles_last) return;
        ovsave64k_last_tile(b_in + (tile >> 4) * (long)F64K_N, out + (tile >> 4) * hop, Ti, hop, scale, (int)(tile & 15), lane, Lw);
    }
}

// Round 3: one launch per chunk step with all THREE tile programs interleaved -- the middle pass of chunk k (a[k & 1] -> b[k & 1]), the last
// pass of chunk k - 1 (b[(k - 1) & 1] -> out) and the gather pass of chunk k + 1 (x -> a[(k + 1) & 1]); the three are independent once the
// work buffers are doubled.  A 64 MiB chunk is exactly one resident set of wavefronts per pass: launched alone, the middle pass has every
// wave of the chip in its load phase, then its arithmetic, then its store phase at the same time; mixed with the other two programs the
// phases of different waves overlap.
struct Ovs64kStep {
    const float2 *a_in; float2 *b_out; long nt_mid;                 // middle pass
    const float2 *b_in; float2 *out; long nt_last;                  // last pass
    const float2 *x_next; float2 *a_out; long nt_first;             // gather pass
};
__global__ __launch_bounds__(256, 2) void ovsave64k_step_kernel(Ovs64kStep st, const float2 *__restrict__ Tf, const float2 *__restrict__ Ti,
                                                             const float2 *__restrict__ tw_f, const float2 *__restrict__ tw_i,
                                                             const float2 *__restrict__ Hc, const float2 *__restrict__ T1, long hop, float scale)
{
    __shared__ __attribute__((aligned(16))) float2 Ls[4 * F64W_REGION];
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float2 *Lw = Ls + w * F64W_REGION;
    // workgroup b -> (kind, workgroup of that kind): round r hands one workgroup to every kind that still has more than r.  (An XCD-aware
    // form -- every XCD a contiguous range of each program's workgroups, the programs alternating per XCD -- was measured: 2.89-2.91 ms against 2.73-2.79.)
    const long n[3] = {(st.nt_mid + 3) / 4, (st.nt_last + 3) / 4, (st.nt_first + 3) / 4};
    long lo = n[0] < n[1] ? n[0] : n[1]; lo = lo < n[2] ? lo : n[2];
    long hi = n[0] > n[1] ? n[0] : n[1]; hi = hi > n[2] ? hi : n[2];
    const long md = n[0] + n[1] + n[2] - lo - hi;
    long b = blockIdx.x, r;
    int kind = 0, skip;
    if (b < 3 * lo) { r = b / 3; skip = (int)(b - 3 * r); for (kind = 0; skip > 0; ++kind) --skip; }
    else if ((b -= 3 * lo) < 2 * (md - lo)) {
        r = lo + b / 2; skip = (int)(b & 1);
        for (kind = 0; kind < 3; ++kind) if (n[kind] > lo && skip-- == 0) break;
    } else {
        r = md + (b - 2 * (md - lo));
        for (kind = 0; kind < 3; ++kind) if (n[kind] > md) break;
    }
    const long tile = r * 4 + w;
    if (kind == 0) {
        if (tile >= st.nt_mid) return;
        ovsave64k_mid_tile(st.a_in + (tile >> 4) * F64K_N, st.b_out + (tile >> 4) * F64K_N, Tf, tw_i, Hc, (int)(tile & 15), lane, Lw);
    } else if (kind == 1) {
        if (tile >= st.nt_last) return;
        ovsave64k_last_tile(st.b_in + (tile >> 4) * (long)F64K_N, st.out + (tile >> 4) * hop, Ti, hop, scale, (int)(tile & 15), lane, Lw);
    } else {
        if (tile >= st.nt_first) return;
        fftbig_first_tile<false, false>(st.x_next + (tile >> 4) * hop, st.a_out + (tile >> 4) * (long)F64K_N, tw_f, 8, (unsigned)(tile & 15), lane, Lw, nullptr, T1);
    }
}

// nblk blocks in chunks of `chunk` blocks (the work buffers a, b hold one chunk each)
hipError_t launch_ovsave64k(const float2 *x, long hop, float2 *a, float2 *b, const float2 *tw_f, const float2 *tw_i, const float2 *Tf,
                            const float2 *Ti, const float2 *Hc, float2 *out, long nblk, long chunk, float scale, hipStream_t s, bool doubled)
{
    if (!Tf || !Ti || chunk < 1) return hipErrorInvalidValue; // the plans' pass-ordered twiddle copies (fftbig_tables_build)
    auto tiles = [&](long b0) { const long nb = nblk - b0 < chunk ? nblk - b0 : chunk; return nb * 16; };
    const float2 *T1 = Tf + (fftbig_tables_elems(F64K_N) - fftbig_first_elems(16)); // the forward plan's gather-pass copy
#if REDIO_TILE_PAIR // the pair program's middle tile reads the inverse gather pass's twiddles from the INVERSE plan's ordered copy, not from the table
    tw_i = Ti + (fftbig_tables_elems(F64K_N) - fftbig_first_elems(16));
#endif
    hipLaunchKernelGGL((fftbig_first_kernel<false, false>), dim3((unsigned)((tiles(0) + 3) / 4)), dim3(256), 0, s, x, a, tw_f, hop, tiles(0), 8, nullptr, T1);
    static const bool fused3 = !measure_env("REDIO_OVS_NO_STEP"); // measurement knob: the two-launch form below
    if (fused3 && doubled) { // a, b hold TWO chunks each
        const long nchunks = (nblk + chunk - 1) / chunk, half = chunk * (long)F64K_N;
        for (long k = 0; k <= nchunks; ++k) {
            Ovs64kStep st = {};
            if (k < nchunks) { st.a_in = a + (k & 1) * half; st.b_out = b + (k & 1) * half; st.nt_mid = tiles(k * chunk); }
            if (k >= 1) { st.b_in = b + ((k - 1) & 1) * half; st.out = out + (k - 1) * chunk * hop; st.nt_last = tiles((k - 1) * chunk); }
            if (k + 1 < nchunks) { st.x_next = x + (k + 1) * chunk * hop; st.a_out = a + ((k + 1) & 1) * half; st.nt_first = tiles((k + 1) * chunk); }
            const unsigned grid = (unsigned)((st.nt_mid + 3) / 4 + (st.nt_last + 3) / 4 + (st.nt_first + 3) / 4);
            hipLaunchKernelGGL(ovsave64k_step_kernel, dim3(grid), dim3(256), 0, s, st, Tf, Ti, tw_f, tw_i, Hc, T1, hop, scale);
        }
        return hipGetLastError();
    }
    for (long b0 = 0; b0 < nblk; b0 += chunk) {
        const long nt = tiles(b0), next = b0 + chunk;
        hipLaunchKernelGGL(ovsave64k_mid_wave_kernel, dim3((unsigned)((nt + 3) / 4)), dim3(256), 0, s, a, b, Tf, tw_i, Hc, nt);
        const long ntn = next < nblk ? tiles(next) : 0;
        const unsigned grid = (unsigned)((nt + 3) / 4 + (ntn + 3) / 4);
        hipLaunchKernelGGL(ovsave64k_last_first_kernel, dim3(grid), dim3(256), 0, s, b, out + b0 * hop, Ti, hop, scale, nt,
                           x + next * hop, a, tw_f, ntn, T1);
    }
    return hipGetLastError();
}

hipError_t launch_ovsave4k(const float2 *x, long hop, const float2 *Tf, const float2 *Ti, const float2 *Hc, float2 *out, long nblk,
                           float scale, hipStream_t s)
{
    if (!Tf || !Ti) return hipErrorInvalidValue; // the plans' stage-ordered twiddle copies
    hipLaunchKernelGGL(ovsave4k_wave_kernel, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, s, x, hop, Tf, Ti, Hc, out, nblk, scale);
    return hipGetLastError();
}

hipError_t launch_ovsave2k(const float2 *x, long hop, const float2 *Tf, const float2 *Ti, const float2 *Hc, float2 *out, long nblk,
                           float scale, hipStream_t s)
{
    if (!Tf || !Ti) return hipErrorInvalidValue; // the plans' stage-ordered twiddle copies
    hipLaunchKernelGGL(ovsave2k_wave_kernel, dim3((unsigned)((nblk + 3) / 4)), dim3(256), 0, s, x, hop, Tf, Ti, Hc, out, nblk, scale);
    return hipGetLastError();
}

hipError_t launch_ovsave8k(const float2 *x, long hop, const float2 *Tf, const float2 *Ti, const float2 *Hc, float2 *out, long nblk,
                           float scale, hipStream_t s)
{
    if (!Tf || !Ti) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ovsave8k_wave_kernel, dim3((unsigned)nblk), dim3(256), 0, s, x, hop, Tf, Ti, Hc, out, scale);
    return hipGetLastError();
}

hipError_t launch_ovsave16k(const float2 *x, long hop, const float2 *tw_f, const float2 *tw_i, const float2 *Tf, const float2 *Ti, const float2 *Hc,
                            float2 *out, long nblk, float scale, hipStream_t s)
{
    if (!Tf || !Ti) return hipErrorInvalidValue;
    hipLaunchKernelGGL(ovsave16k_wave_kernel, dim3((unsigned)nblk), dim3(256), 0, s, x, hop, tw_f, tw_i, Tf, Ti, Hc, out, scale);
    return hipGetLastError();
}

hipError_t launch_fft(const FftPlanDev &p, const float2 *in, float2 *out, long nbatch, hipStream_t s, long in_stride, float2 *work)
{
    if (in_stride <= 0) in_stride = p.nfft; // consecutive messages; smaller strides give overlapping blocks (overlap-save)
    if (nbatch <= 0) return hipSuccess;
    const bool inv = p.inverse != 0;
    if (p.nfft == 1024) {
        const size_t lds = 4 * FFT1K_LDS * sizeof(float2);
        const unsigned grid = (unsigned)((nbatch + 4 * FFT1K_RUN - 1) / (4 * FFT1K_RUN));
        if (inv) hipLaunchKernelGGL(fft1k_wave_kernel<true>, dim3(grid), dim3(256), lds, s, in, out, p.tw, nbatch, in_stride);
        else hipLaunchKernelGGL(fft1k_wave_kernel<false>, dim3(grid), dim3(256), lds, s, in, out, p.tw, nbatch, in_stride);
        return hipGetLastError();
    }
    switch (p.nfft) { // 2 * 4^L, and the two powers of four below 64
    case 2: return launch_fft_p2<1>(in, out, p.tw, p.tw_pass, nbatch, in_stride, inv, s);
    case 4: return launch_fft_p2<2>(in, out, p.tw, p.tw_pass, nbatch, in_stride, inv, s);
    case 8: return launch_fft_p2<3>(in, out, p.tw, p.tw_pass, nbatch, in_stride, inv, s);
    case 16: return launch_fft_p2<4>(in, out, p.tw, p.tw_pass, nbatch, in_stride, inv, s);
    case 32: return launch_fft_p2<5>(in, out, p.tw, p.tw_pass, nbatch, in_stride, inv, s);
    case 128: return launch_fft_p2<7>(in, out, p.tw, p.tw_pass, nbatch, in_stride, inv, s);
    case 512: return launch_fft_p2<9>(in, out, p.tw, p.tw_pass, nbatch, in_stride, inv, s);
    case 2048: {
        const unsigned grid = (unsigned)((nbatch + 3) / 4);
        if (!p.tw_pass) return hipErrorInvalidValue;
        if (inv) hipLaunchKernelGGL(fft2k_wave_kernel<true>, dim3(grid), dim3(256), 0, s, in, out, p.tw_pass, nbatch, in_stride);
        else hipLaunchKernelGGL(fft2k_wave_kernel<false>, dim3(grid), dim3(256), 0, s, in, out, p.tw_pass, nbatch, in_stride);
        return hipGetLastError();
    }
    case 8192:
        if (!p.tw_pass) return hipErrorInvalidValue;
        if (inv) hipLaunchKernelGGL(fft8k_wave_kernel<true>, dim3((unsigned)nbatch), dim3(256), 0, s, in, out, p.tw, p.tw_pass, in_stride);
        else hipLaunchKernelGGL(fft8k_wave_kernel<false>, dim3((unsigned)nbatch), dim3(256), 0, s, in, out, p.tw, p.tw_pass, in_stride);
        return hipGetLastError();
    default: break;
    }
    if (p.nfft == 64) {
        const size_t lds = 4 * FFT1K_LDS * sizeof(float2);
        const unsigned grid = (unsigned)((nbatch + 63) / 64);
        if (inv) hipLaunchKernelGGL(fft64_kernel<true>, dim3(grid), dim3(256), lds, s, in, out, p.tw, nbatch, in_stride);
        else hipLaunchKernelGGL(fft64_kernel<false>, dim3(grid), dim3(256), lds, s, in, out, p.tw, nbatch, in_stride);
        return hipGetLastError();
    }
    if (p.nfft == 256) {
        const size_t lds = 4 * FFT1K_LDS * sizeof(float2);
        const unsigned grid = (unsigned)((nbatch + 15) / 16);
        if (inv) hipLaunchKernelGGL(fft256_kernel<true>, dim3(grid), dim3(256), lds, s, in, out, p.tw, nbatch, in_stride);
        else hipLaunchKernelGGL(fft256_kernel<false>, dim3(grid), dim3(256), lds, s, in, out, p.tw, nbatch, in_stride);
        return hipGetLastError();
    }
    if (p.nfft == 4096) {
        const unsigned grid = (unsigned)((nbatch + 3) / 4);
        if (!p.tw_pass) return hipErrorInvalidValue;
        if (inv) hipLaunchKernelGGL(fft4k_wave_kernel<true>, dim3(grid), dim3(256), 0, s, in, out, p.tw_pass, nbatch, in_stride);
        else hipLaunchKernelGGL(fft4k_wave_kernel<false>, dim3(grid), dim3(256), 0, s, in, out, p.tw_pass, nbatch, in_stride);
        return hipGetLastError();
    }
    if (p.nfft == 16384) {
        if (!p.tw_pass) return hipErrorInvalidValue;
        if (inv) hipLaunchKernelGGL(fft16k_wave_kernel<true>, dim3((unsigned)nbatch), dim3(256), 0, s, in, out, p.tw, p.tw_pass, in_stride);
        else hipLaunchKernelGGL(fft16k_wave_kernel<false>, dim3((unsigned)nbatch), dim3(256), 0, s, in, out, p.tw, p.tw_pass, in_stride);
        return hipGetLastError();
    }
    {   // sizes with a compile-time pass list
        hipError_t e = hipErrorNotSupported;
        switch (p.nfft) {
#define REDIO_CT(NN) case NN: e = launch_fft_ct<NN>(p, in, out, nbatch, in_stride, inv, s); break;
            REDIO_CT(6) REDIO_CT(9) REDIO_CT(10) REDIO_CT(12) REDIO_CT(15) REDIO_CT(18) REDIO_CT(20) REDIO_CT(24)
            REDIO_CT(25) REDIO_CT(27) REDIO_CT(30) REDIO_CT(36) REDIO_CT(40) REDIO_CT(45) REDIO_CT(48) REDIO_CT(50)
            REDIO_CT(54) REDIO_CT(60) REDIO_CT(72) REDIO_CT(75) REDIO_CT(80) REDIO_CT(81) REDIO_CT(90) REDIO_CT(96)
            REDIO_CT(100) REDIO_CT(108) REDIO_CT(120) REDIO_CT(125) REDIO_CT(135) REDIO_CT(144) REDIO_CT(150) REDIO_CT(160)
            REDIO_CT(162) REDIO_CT(180) REDIO_CT(192) REDIO_CT(200) REDIO_CT(216) REDIO_CT(225) REDIO_CT(240) REDIO_CT(243)
            REDIO_CT(250) REDIO_CT(270) REDIO_CT(288) REDIO_CT(300) REDIO_CT(320) REDIO_CT(324) REDIO_CT(360) REDIO_CT(375)
            REDIO_CT(384) REDIO_CT(400) REDIO_CT(405) REDIO_CT(432) REDIO_CT(450) REDIO_CT(480) REDIO_CT(486) REDIO_CT(500)
            REDIO_CT(540) REDIO_CT(576) REDIO_CT(600) REDIO_CT(625) REDIO_CT(640) REDIO_CT(648) REDIO_CT(675) REDIO_CT(720)
            REDIO_CT(729) REDIO_CT(750) REDIO_CT(768) REDIO_CT(800) REDIO_CT(810) REDIO_CT(864) REDIO_CT(900) REDIO_CT(960)
            REDIO_CT(972) REDIO_CT(1000) REDIO_CT(1080) REDIO_CT(1125) REDIO_CT(1152) REDIO_CT(1200) REDIO_CT(1215) REDIO_CT(1250)
            REDIO_CT(1280) REDIO_CT(1296) REDIO_CT(1350) REDIO_CT(1440) REDIO_CT(1458) REDIO_CT(1500) REDIO_CT(1536) REDIO_CT(1600)
            REDIO_CT(1620) REDIO_CT(1728) REDIO_CT(1800) REDIO_CT(1875) REDIO_CT(1920) REDIO_CT(1944) REDIO_CT(2000) REDIO_CT(2025)
            REDIO_CT(2160) REDIO_CT(2187) REDIO_CT(2250) REDIO_CT(2304) REDIO_CT(2400) REDIO_CT(2430) REDIO_CT(2500) REDIO_CT(2560)
            REDIO_CT(2592) REDIO_CT(2700) REDIO_CT(2880) REDIO_CT(2916) REDIO_CT(3000) REDIO_CT(3072) REDIO_CT(3125) REDIO_CT(3200)
            REDIO_CT(3240) REDIO_CT(3375) REDIO_CT(3456) REDIO_CT(3600) REDIO_CT(3645) REDIO_CT(3750) REDIO_CT(3840) REDIO_CT(3888)
            REDIO_CT(4000) REDIO_CT(4050) REDIO_CT(4320) REDIO_CT(4374) REDIO_CT(4500) REDIO_CT(4608) REDIO_CT(4800) REDIO_CT(4860)
            REDIO_CT(5000) REDIO_CT(5120) REDIO_CT(5184) REDIO_CT(5400) REDIO_CT(5625) REDIO_CT(5760) REDIO_CT(5832) REDIO_CT(6000)
            REDIO_CT(6075) REDIO_CT(6144) REDIO_CT(6250) REDIO_CT(6400) REDIO_CT(6480) REDIO_CT(6561) REDIO_CT(6750) REDIO_CT(6912)
            REDIO_CT(7200) REDIO_CT(7290) REDIO_CT(7500) REDIO_CT(7680) REDIO_CT(7776) REDIO_CT(8000) REDIO_CT(8100)
            REDIO_CT(8640) REDIO_CT(8748) REDIO_CT(9000) REDIO_CT(9216) REDIO_CT(9375) REDIO_CT(9600) REDIO_CT(9720) REDIO_CT(10000)
            REDIO_CT(10125) REDIO_CT(10240) REDIO_CT(10368) REDIO_CT(10800) REDIO_CT(10935) REDIO_CT(11250) REDIO_CT(11520) REDIO_CT(11664)
            REDIO_CT(12000) REDIO_CT(12150) REDIO_CT(12288) REDIO_CT(12500) REDIO_CT(12800) REDIO_CT(12960) REDIO_CT(13122) REDIO_CT(13500)
            REDIO_CT(13824) REDIO_CT(14400) REDIO_CT(14580) REDIO_CT(15000) REDIO_CT(15360) REDIO_CT(15552) REDIO_CT(15625) REDIO_CT(16000)
            REDIO_CT(16200)
#undef REDIO_CT
        default: break;
        }
        if (e != hipErrorNotSupported) return e;
    }
    bool generic = false;
    for (int i = 0; i < p.nstages; ++i) generic |= p.st[i].p > 5;
    if (!generic && p.nfft <= 8192) {
        int T = 2048 / p.nfft;
        if (T < 1) T = 1;
        if (T > nbatch) T = (int)nbatch;
        int G = 256; // lanes per transform: the power of two at or above a quarter of its points
        while (G > 1 && G / 2 >= (p.nfft + 3) / 4) G /= 2;
        const size_t lds = (size_t)T * p.nfft * sizeof(float2);
        auto kf = fft_lds_batched_kernel<false>;
        auto ki = fft_lds_batched_kernel<true>;
        if (lds > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(inv ? ki : kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        const unsigned grid = (unsigned)((nbatch + T - 1) / T);
        if (inv) hipLaunchKernelGGL(ki, dim3(grid), dim3(256), lds, s, p, in, out, nbatch, in_stride, T, G);
        else hipLaunchKernelGGL(kf, dim3(grid), dim3(256), lds, s, p, in, out, nbatch, in_stride, T, G);
        return hipGetLastError();
    }
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
    if (fftbig_size(p.nfft) && p.tw_pass) { // 32768, 65536, 131072 ... 16777216 (16384 above)
        if (in == out) return hipErrorNotSupported; // the first pass is a global transposition: the C-ABI layer stages in-place calls
        const int lgN = __builtin_ctz((unsigned)p.nfft);
        return inv ? launch_fftbig<true>(in, out, p.tw, p.tw_pass, nbatch, in_stride, lgN, s)
                   : launch_fftbig<false>(in, out, p.tw, p.tw_pass, nbatch, in_stride, lgN, s);
    }
    if (!generic && p.nfft > 16384 && p.tw_pass) { // radix-2/3/4/5 sizes that are not powers of two: one pass per group of stages
        if (in == out) return hipErrorNotSupported; // the first pass is a global transposition: the C-ABI layer stages in-place calls
        if (nbatch * (long)((p.nfft + 15) / 16) > 0x7fffffffl) return hipErrorInvalidValue;
        return inv ? launch_fft_tile_passes<true>(p, in, out, nbatch, in_stride, s) : launch_fft_tile_passes<false>(p, in, out, nbatch, in_stride, s);
    }
    // global-memory stages.  The C-ABI layer routes in-place calls through a temporary and supplies `work`
    // (nbatch * nfft elements) when a generic-radix stage needs an out-of-place step.
    if (in == out || (generic && !work)) return hipErrorNotSupported;
    const long total = nbatch * p.nfft;
    const unsigned egrid = (unsigned)((total + 255) / 256);
    float2 *cur = out, *other = work;
    hipLaunchKernelGGL(fft_global_leaf_kernel, dim3(egrid), dim3(256), 0, s, p, in, cur, total, in_stride);
    for (int st = p.nstages - 1; st >= 0; --st) {
        if (p.st[st].p > 5) {
            if (inv) hipLaunchKernelGGL(fft_global_generic_stage_kernel<true>, dim3(egrid), dim3(256), 0, s, p, st, cur, other, total);
            else hipLaunchKernelGGL(fft_global_generic_stage_kernel<false>, dim3(egrid), dim3(256), 0, s, p, st, cur, other, total);
            float2 *t = cur; cur = other; other = t;
            continue;
        }
        const long nb = nbatch * (p.nfft / p.st[st].p);
        const unsigned grid = (unsigned)((nb + 255) / 256);
        if (inv) hipLaunchKernelGGL(fft_global_stage_kernel<true>, dim3(grid), dim3(256), 0, s, p, st, cur, nb);
        else hipLaunchKernelGGL(fft_global_stage_kernel<false>, dim3(grid), dim3(256), 0, s, p, st, cur, nb);
    }
    if (cur != out) {
        hipError_t e = hipMemcpyAsync(out, cur, (size_t)total * sizeof(float2), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

} // namespace redio
