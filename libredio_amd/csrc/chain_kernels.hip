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
#include <type_traits>

#ifndef REDIO_CHAIN_RELOAD_TWIDDLES
#define REDIO_CHAIN_RELOAD_TWIDDLES 0
#endif

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

// ---------------------------------------------------------------------------------------------
// v2: wave-autonomous FIR + persistent grid.
//   * Each wavefront owns a private LDS image of the 1402 input samples behind its 256 outputs of a
//     block (the four waves of a workgroup split a 1024-sample block), so the FIR phase has no
//     workgroup barrier: load -> wave fence -> compute.
//   * The next sub-tile is fetched into registers (11 x 16 B per lane) BEFORE the current one is
//     computed and written to LDS after it, so HBM latency hides behind the 508 packed FMAs.
//   * Windows are read with ds_read_b128 (two samples per read; FirGeomV lane stride 22 float2).
//   * A workgroup walks groups of four blocks with a grid stride; after four FIR tiles two adjacent
//     barriers hand the four blocks to the four waves, one 1024-point transform each, with the
//     wave's own (now idle) input image as exchange scratch.  The fourth block is parked in the idle
//     input images instead of a fourth block buffer, which is what lets two workgroups share a CU.
// ---------------------------------------------------------------------------------------------
// native 16-byte vector: struct-typed float4 copies lower to memcpy and keep arrays in scratch
typedef float v4f __attribute__((ext_vector_type(4)));

template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int K, int D, bool FUSED>
__global__ __launch_bounds__(256, 2) void chain_v2_kernel(const float2 *__restrict__ x, const float *__restrict__ taps,
                                                          const float2 *__restrict__ tw, float2 *__restrict__ out,
                                                          long nblocks)
{
    constexpr int R = 4;
    using G = FirGeomV<K, D, R>;
    constexpr int SUB_OUT = 64 * R;               // outputs per wave per block
    constexpr int SUB_IN = G::tile_in(SUB_OUT);   // input samples behind them
    static_assert(SUB_IN % 2 == 0 && 4 * SUB_OUT == 1024, "geometry");
    constexpr int SUB_V = SUB_IN / 2;             // float4 loads per sub-tile
    constexpr int NLD = (SUB_V + 63) / 64;        // per lane
    constexpr int XS4 = (G::lds_elems(SUB_OUT) > FFT1K_LDS ? G::lds_elems(SUB_OUT) : FFT1K_LDS) / 2; // float4 per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v4f *xs_all = reinterpret_cast<v4f *>(smem);            // [4][XS4]
    float2 *yb = reinterpret_cast<float2 *>(xs_all + 4 * XS4);    // [3][1024]

    // wave index as a scalar: every block-dependent address becomes SGPR base + lane offset
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    v4f *xs4 = xs_all + wave * XS4;
    const long ngroups = (nblocks + 3) >> 2;
    long g = blockIdx.x;
    if (g >= ngroups) return; // whole workgroup

    v4f pre[NLD];
    auto fetch = [&](long blk) {
        const v4f *src = reinterpret_cast<const v4f *>(x + (blk * 1024 + wave * SUB_OUT) * (long)D) + lane;
        static_for<NLD - 1>([&](auto I) { pre[I.value] = src[64 * I.value]; });
        if (lane + 64 * (NLD - 1) < SUB_V) pre[NLD - 1] = src[64 * (NLD - 1)];
    };
    auto park = [&]() { // registers -> this wave's LDS image
        static_for<NLD - 1>([&](auto I) { xs4[G::lds_index(2 * (lane + 64 * I.value)) / 2] = pre[I.value]; });
        if (lane + 64 * (NLD - 1) < SUB_V) xs4[G::lds_index(2 * (lane + 64 * (NLD - 1))) / 2] = pre[NLD - 1];
    };

    Fft1kTw twl; // lane-dependent twiddles: loaded once, live in registers across the persistent loop
    fft1k_load_tw(twl, lane, tw);
    fetch(4 * g); // the first block of a group always exists
    park();
    wave_lds_fence();
    for (; g < ngroups; g += gridDim.x) {
#pragma unroll 1
        for (int t = 0; t < 4; ++t) {
            const long blk = 4 * g + t;
            const long nxt = (t < 3) ? blk + 1 : 4 * (g + gridDim.x);
            if (nxt < nblocks) fetch(nxt);
            if (blk < nblocks) {
                float2 acc[R];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);
                int lf = lane; // opaque per tile: keeps LDS address arithmetic out of the persistent loop's live set
                asm volatile("" : "+v"(lf));
                fir_lane_v<K, D, R, FUSED>(xs4, lf, taps, acc);
                wave_lds_fence(); // window reads done before the image is reused
                v4f *yt = (t < 3) ? reinterpret_cast<v4f *>(yb + t * 1024 + wave * SUB_OUT) + 2 * lane
                                  : xs4 + 2 * lane; // block 3 lives in the idle input images
                yt[0] = v4f{acc[0].x, acc[0].y, acc[1].x, acc[1].y};
                yt[1] = v4f{acc[2].x, acc[2].y, acc[3].x, acc[3].y};
            }
            if (t < 3) {
                if (nxt < nblocks) park();
                wave_lds_fence();
            }
        }
        __syncthreads(); // the group's four blocks are complete
        const long blk = 4 * g + wave;
        float2 v[16];
        int ln = lane; // opaque per group, same reason
        asm volatile("" : "+v"(ln));
        if (blk < nblocks) {
            if (wave < 3) {
                const float2 *src = yb + wave * 1024;
#pragma unroll
                for (int q = 0; q < 16; ++q) v[q] = src[ln + 64 * q];
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    v[q] = reinterpret_cast<const float2 *>(xs_all + (q >> 2) * XS4)[ln + 64 * (q & 3)];
            }
        }
        __syncthreads(); // every block is in registers: images and block buffers are free again
#if REDIO_CHAIN_RELOAD_TWIDDLES
        fft1k_load_tw(twl, ln, tw); // re-read the 27 per-lane twiddles every group (L1/L2 hits)
#endif
        if (blk < nblocks) fft1k_wave_regs<false>(v, out + blk * 1024, reinterpret_cast<float2 *>(xs4), tw, twl, ln);
        wave_lds_fence();
        if (4 * (g + gridDim.x) < nblocks) park();
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------------
// v3: one wavefront per 1024-sample block, no workgroup at all.
//   * The wave runs the four FIR sub-tiles of ITS OWN block back to back and keeps the 16 results per
//     lane in registers: a[4 s + r] = y[256 s + 4 lane + r].  That is exactly the operand layout of
//     the "native" transform in fft_core.h, whose first and last stages are register-only -- so the
//     decimated block is never written to or read from LDS, there is no block buffer and no barrier.
//   * LDS per wave = one 1402-sample input image (12.3 KB), reused as the transform's exchange
//     scratch; twelve waves fit a CU.
//   * Input prefetch into registers one sub-tile ahead (across block boundaries too), tap and window
//     double buffering as in v2.
// ---------------------------------------------------------------------------------------------
// ABLATE (timing-only builds, wrong results): 1 = skip the FIR multiply-adds, 2 = skip the transform,
// 4 = fetch only the very first sub-tile from HBM.  They produced the breakdown quoted in DESIGN.md 5.1; the
// dispatch no longer instantiates them (a public call must never return wrong results).
template <int K, int D, bool FUSED, int WPS, int CH, int ABLATE = 0>
__global__ __launch_bounds__(64, WPS) void chain_v3_kernel(const float2 *__restrict__ x, const float *__restrict__ taps,
                                                         const float2 *__restrict__ tw, float2 *__restrict__ out,
                                                         long nblocks)
{
    constexpr int R = 4;
    using G = FirGeomV<K, D, R>;
    constexpr int SUB_OUT = 64 * R;
    constexpr int SUB_IN = G::tile_in(SUB_OUT);
    static_assert(SUB_IN % 2 == 0 && 4 * SUB_OUT == 1024, "geometry");
    constexpr int SUB_V = SUB_IN / 2;
    constexpr int NLD = (SUB_V + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v4f *xs4 = reinterpret_cast<v4f *>(smem);
    float2 *ex = reinterpret_cast<float2 *>(smem);

    const int lane = threadIdx.x;
    long blk = blockIdx.x;
    if (blk >= nblocks) return;

    v4f pre[NLD];
    auto fetch = [&](long b, int s) {
        const v4f *src = reinterpret_cast<const v4f *>(x + (b * 1024 + s * SUB_OUT) * (long)D) + lane;
        static_for<NLD - 1>([&](auto I) { pre[I.value] = src[64 * I.value]; });
        if (lane + 64 * (NLD - 1) < SUB_V) pre[NLD - 1] = src[64 * (NLD - 1)];
    };
    auto park = [&]() {
        static_for<NLD - 1>([&](auto I) { xs4[G::lds_index(2 * (lane + 64 * I.value)) / 2] = pre[I.value]; });
        if (lane + 64 * (NLD - 1) < SUB_V) xs4[G::lds_index(2 * (lane + 64 * (NLD - 1))) / 2] = pre[NLD - 1];
    };

    fetch(blk, 0);
    park();
    wave_lds_fence();
    for (; blk < nblocks; blk += gridDim.x) {
        float2 a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) a[i] = make_float2(0.f, 0.f);
        const bool more = blk + gridDim.x < nblocks;
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            if (!(ABLATE & 4)) {
                if (s < 3) fetch(blk, s + 1);
                else if (more) fetch(blk + gridDim.x, 0);
            }
            float2 acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);
            int lf = lane; // opaque per sub-tile: keeps LDS address arithmetic out of the loop's live set
            asm volatile("" : "+v"(lf));
            if (ABLATE & 1) {
                const v4f q0 = xs4[lf * (G::LANE_STRIDE / 2)];
                acc[0] = make_float2(q0.x, q0.y); acc[1] = make_float2(q0.z, q0.w); acc[2] = acc[0]; acc[3] = acc[1];
            } else {
                fir_lane_v<K, D, R, FUSED, CH>(xs4, lf, taps, acc);
            }
            // rotate: after four sub-tiles a[4 s + r] holds sub-tile s
#pragma unroll
            for (int i = 0; i < 12; ++i) a[i] = a[i + 4];
#pragma unroll
            for (int r = 0; r < R; ++r) a[12 + r] = acc[r];
            wave_lds_fence(); // window reads done before the image is overwritten
            if (s < 3) {
                park();
                wave_lds_fence();
            }
        }
        // the block is in registers; the input image is free -> exchange scratch
        int ln = lane;
        asm volatile("" : "+v"(ln));
        if (ABLATE & 2) {
            if (ABLATE & 8) { // 16-byte stores (timing probe for the store width)
                v4f *d4 = reinterpret_cast<v4f *>(out + blk * 1024) + ln;
#pragma unroll
                for (int i = 0; i < 8; ++i) d4[64 * i] = v4f{a[2 * i].x, a[2 * i].y, a[2 * i + 1].x, a[2 * i + 1].y};
            } else {
                float2 *d0p = out + blk * 1024 + ln;
#pragma unroll
                for (int i = 0; i < 16; ++i) d0p[64 * i] = a[i];
            }
            if (more) park();
            wave_lds_fence();
            continue;
        }
        fft1kn_stage0<false>(a, tw);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4)
#pragma unroll
            for (int d0 = 0; d0 < 4; ++d0) ex[fft1kn_x1_store(ln, k4, d0)] = a[4 * k4 + d0];
        Fft1knTw12 t12;
        fft1kn_load_tw12(t12, ln, tw);
        wave_lds_fence();
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] = ex[fft1kn_x1_load(ln, e)];
        wave_lds_fence();
        fft1kn_pass12<false>(a, t12);
#pragma unroll
        for (int k3 = 0; k3 < 4; ++k3)
#pragma unroll
            for (int k2 = 0; k2 < 4; ++k2) ex[fft1kn_x2_store(ln, k2, k3)] = a[k2 + 4 * k3];
        Fft1knTw34 t34;
        fft1kn_load_tw34(t34, ln, tw);
        wave_lds_fence();
#pragma unroll
        for (int f = 0; f < 16; ++f) a[f] = ex[fft1kn_x2_load(ln, f)];
        wave_lds_fence();
        fft1kn_pass34<false>(a, t34);
        float2 *dst = out + blk * 1024 + ln;
#pragma unroll
        for (int k0 = 0; k0 < 4; ++k0)
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) dst[64 * k1 + 256 * k0] = a[k1 + 4 * k0];
        if (more) park();
        wave_lds_fence();
    }
}

hipError_t launch_chain_v4(const float2 *x, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused, int wps,
                           hipStream_t s, unsigned long long *dbg); // chain_v4.hip
hipError_t launch_chain_v5(const float2 *x, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused, int tuning,
                           unsigned *queue, hipStream_t s); // chain_v5.hip
static unsigned long long *g_chain_dbg = nullptr; // diagnostic stamps (tools/clock_probe.py), never set in production
void chain_set_debug_buffer(unsigned long long *p) { g_chain_dbg = p; }

hipError_t launch_chain_v4_shape(int K, int D, const float2 *x, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused,
                                 hipStream_t s); // chain_v4.hip

bool chain_supported(int K, long D, int nfft)
{
    if (nfft != 1024) return false;
    return (K == 127 && (D == 5 || D == 1 || D == 3)) || (K == 63 && (D == 5 || D == 1));
}

static int num_cus()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int K, int D>
static hipError_t launch_chain_v2(const FftPlanDev &p, const float2 *x, const float *taps, float2 *out,
                                  long nblocks, bool fused, hipStream_t s)
{
    using G = FirGeomV<K, D, 4>;
    constexpr int XS4 = (G::lds_elems(256) > FFT1K_LDS ? G::lds_elems(256) : FFT1K_LDS) / 2;
    constexpr size_t LDS = (size_t)4 * XS4 * sizeof(float4) + 3 * 1024 * sizeof(float2);
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
    auto kf = chain_v2_kernel<K, D, true>;
    auto ke = chain_v2_kernel<K, D, false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fused ? kf : ke),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
    if (e != hipSuccess) return e;
    const long ngroups = (nblocks + 3) / 4;
    long grid = 2L * num_cus();
    if (grid > ngroups) grid = ngroups;
    if (fused) hipLaunchKernelGGL(kf, dim3((unsigned)grid), dim3(256), LDS, s, x, taps, p.tw, out, nblocks);
    else hipLaunchKernelGGL(ke, dim3((unsigned)grid), dim3(256), LDS, s, x, taps, p.tw, out, nblocks);
    return hipGetLastError();
}

template <int K, int D, int WPS, int CH, int ABLATE = 0>
static hipError_t launch_chain_v3(const FftPlanDev &p, const float2 *x, const float *taps, float2 *out,
                                  long nblocks, bool fused, hipStream_t s)
{
    using G = FirGeomV<K, D, 4>;
    constexpr int ELEMS = G::lds_elems(256) > FFT1KN_LDS ? G::lds_elems(256) : FFT1KN_LDS;
    constexpr size_t LDS = (size_t)ELEMS * sizeof(float2);
    static_assert(4 * WPS * LDS <= 160 * 1024, "4*WPS waves per CU");
    long grid = 4L * WPS * num_cus();
    if (grid > nblocks) grid = nblocks;
    if (fused) hipLaunchKernelGGL((chain_v3_kernel<K, D, true, WPS, CH, ABLATE>), dim3((unsigned)grid), dim3(64), LDS, s, x, taps, p.tw, out, nblocks);
    else hipLaunchKernelGGL((chain_v3_kernel<K, D, false, WPS, CH, ABLATE>), dim3((unsigned)grid), dim3(64), LDS, s, x, taps, p.tw, out, nblocks);
    return hipGetLastError();
}

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
                        float2 *out, long nblocks, bool fused, int variant, hipStream_t s, unsigned *queue)
{
    if (nblocks <= 0) return hipSuccess;
    if (p.nfft == 1024 && !p.inverse && K == 127 && D == 5) {
        // v2/v3 need 16-byte aligned input (every sub-tile starts on an even sample)
        const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0;
        // default: v4 (static contiguous block ranges).  v5 (dynamic chunk queue) measures the same, 7-10 select it.
        if (variant == 0 && aligned) return launch_chain_v4(x, taps, p.tw, out, nblocks, fused, 2, s, g_chain_dbg);
        if (variant >= 7 && variant <= 10 && aligned && queue) return launch_chain_v5(x, taps, p.tw, out, nblocks, fused, variant - 7, queue, s);
        if (variant == 5 && aligned) return launch_chain_v4(x, taps, p.tw, out, nblocks, fused, 3, s, g_chain_dbg);
        if (variant == 31 && aligned) return launch_chain_v4(x, taps, p.tw, out, nblocks, fused, 12, s, g_chain_dbg);
        if (variant == 6 && aligned) return launch_chain_v3<127, 5, 2, 8>(p, x, taps, out, nblocks, fused, s);
        if (variant == 3 && aligned) return launch_chain_v3<127, 5, 3, 6>(p, x, taps, out, nblocks, fused, s);
        if (variant == 4 && aligned) return launch_chain_v3<127, 5, 3, 8>(p, x, taps, out, nblocks, fused, s);
        if (variant == 2 && aligned) return launch_chain_v2<127, 5>(p, x, taps, out, nblocks, fused, s);
        return launch_chain_t<127, 5>(p, x, n_in, taps, out, nblocks, fused, s);
    }
    if (p.nfft == 1024 && !p.inverse && (reinterpret_cast<uintptr_t>(x) & 15) == 0) // other fused shapes need the aligned stream
        return launch_chain_v4_shape(K, (int)D, x, taps, p.tw, out, nblocks, fused, s);
    return hipErrorNotSupported; // the C-ABI layer then runs the two-kernel path
}

} // namespace redio
