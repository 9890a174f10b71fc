// chain_v4.hip -- the north-star chain kernel: one wavefront per run of consecutive 1024-sample decimated blocks.
//   * a wave owns a CONTIGUOUS range of blocks, so consecutive sub-tiles (256 outputs) are consecutive in the stream;
//   * the 122-sample FIR halo that two consecutive sub-tiles share is carried over inside the CU
//     (one 16-byte LDS read + write per lane) instead of being fetched again: per sub-tile exactly
//     640 x 16 B = 10 loads per lane of NEW samples;
//   * the FIR results stay in registers in the operand layout of the one-wave 1024-point transform.
// Bit-exact with the oracle (FIR fold order of dsputils.rs:30-32, kissfft butterfly order).
#ifndef REDIO_EXP_CHAIN_NT
#define REDIO_EXP_CHAIN_NT 3 // bit 0: non-temporal stream loads, bit 1: non-temporal spectrum stores (fft_wave.h)
#endif
#include "fir_core.h"
#include "fft_wave.h"
#include "redio_internal.h"
#include <type_traits>

namespace redio {

typedef float v4f_t __attribute__((ext_vector_type(4)));

template <int N, typename F>
__device__ __forceinline__ void static_for4(F &&f)
{
    if constexpr (N > 0) {
        static_for4<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// FIR_ONLY: the same data movement and FIR, but the decimated samples are stored instead of transformed
// (the stand-alone decimating FIR of redio_fir_* for this shape); `out` then holds 1024 outputs per block.
// IN_U8: `x` is the receiver's own format, interleaved u8 I/Q bytes (rtlsdr::data_to_samples, rtlsdr.rs:159-162: i as f32 / 127.0 - 1.0):
// a lane's 16-byte load of two cf32 samples becomes a 4-byte load of the same two samples, converted on the way into the LDS image --
// 2 + 1.6 bytes per sample through HBM instead of 8 + 8 + 1.6 for the conversion kernel followed by this one.
// PF2: TWO sub-tiles of loads in flight (round 6): a second register set of NLD pairs, the loop unrolled by two with the
// roles of the sets swapped -- the request for sub-tile j + 2 is issued before the FIR of sub-tile j, the set holding j + 1 is parked behind it.
template <int K, int D, bool FUSED, int WPS, int CH, bool FIR_ONLY = false, bool TWP = false, bool IN_U8 = false, bool PF2 = false>
__global__ __launch_bounds__(64, WPS) void chain_v4_kernel(const float2 *__restrict__ x, const float *__restrict__ taps,
                                                           const float2 *__restrict__ tw, float2 *__restrict__ out,
                                                           long nblocks, long blocks_per_wave, unsigned long long *dbg, long dbg_cap)
{
    // diagnostic only (dbg != nullptr): shader-clock and 100 MHz real-time stamps around the wave's life,
    // written to a buffer of their own; run times of such launches are never quoted
    unsigned long long t0c = 0, t0r = 0;
    if (dbg) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
    constexpr int R = 4;
    using G = FirGeomV<K, D, R>;
    constexpr int SUB_OUT = 64 * R;                   // 256 outputs per sub-tile
    constexpr int SUB_NEW = SUB_OUT * D;              // 1280 new input samples per sub-tile
    constexpr int HALO = G::tile_in(SUB_OUT) - SUB_NEW; // 122 samples shared with the previous sub-tile
    static_assert(HALO % 2 == 0 && SUB_NEW % 128 == 0 && 4 * SUB_OUT == 1024, "geometry");
    constexpr int HALO_V = HALO / 2;                  // 61 float4
    constexpr int NLD = SUB_NEW / 2 / 64;             // 10 float4 per lane
    static_assert(HALO_V <= 64, "the halo moves with one instruction per lane");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v4f_t *xs4 = reinterpret_cast<v4f_t *>(smem);
    float2 *ex = reinterpret_cast<float2 *>(smem);

    const int lane = threadIdx.x;
    const long b0 = (long)blockIdx.x * blocks_per_wave;
    if (b0 >= nblocks) return;
    const long b1 = (b0 + blocks_per_wave < nblocks) ? b0 + blocks_per_wave : nblocks;
    const long nsub = 4 * (b1 - b0);                  // sub-tiles of this wave, consecutive in the stream
    // index of the first NEW pair of samples of sub-tile j (the halo precedes it): a pair is one float4, or one u32 of I/Q bytes
    using pair_t = typename std::conditional<IN_U8, unsigned, v4f_t>::type;
    const pair_t *src0 = (IN_U8 ? reinterpret_cast<const pair_t *>(reinterpret_cast<const unsigned *>(x) + b0 * 512 * (long)D)
                                : reinterpret_cast<const pair_t *>(x + b0 * 1024 * (long)D)) + lane;
    auto samples = [](pair_t w) -> v4f_t {
        if constexpr (IN_U8) return v4f_t{i2f(w & 255u), i2f((w >> 8) & 255u), i2f((w >> 16) & 255u), i2f(w >> 24)};
        else return w;
    };

    pair_t pre[NLD], pre2[PF2 ? NLD : 1];
    // PF2 reads the stream through a buffer descriptor of the wave's own run, [head of sub-tile 0, end of sub-tile nsub - 1): a request
    // for a sub-tile behind the run's last one (the last two steps of a run ask for them) lies past num_records and returns zeros
    // WITHOUT a memory access, so every step requests unconditionally and the number of loads in flight at every wait is a
    // compile-time fact (a request under a condition makes the compiler wait for ALL loads at the next use, i.e. also for the
    // request it issued a sub-tile ago -- which is the whole point of the second set)
    __amdgpu_buffer_rsrc_t rs;
    if constexpr (PF2) {
        static_assert(!PF2 || !IN_U8, "PF2: cf32 input");
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2 *>(x + b0 * 1024 * (long)D), 0, (int)((HALO + nsub * SUB_NEW) * 8), 0x00020000);
    }
    auto fetch_to = [&](pair_t(&dst)[NLD], long j) { // new samples of sub-tile j: [HALO + j*SUB_NEW, HALO + (j+1)*SUB_NEW)
        if constexpr (PF2) {
            typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
            const unsigned vo = 16u * (unsigned)lane + (unsigned)((HALO + j * SUB_NEW) * 8);
            static_for4<NLD>([&](auto I) {
                constexpr int k = I.value;
                const v4u_t w = __builtin_amdgcn_raw_buffer_load_b128(rs, vo + 1024u * (k & 3), 4096 * (k >> 2), 2 /* nt */);
                dst[k] = __builtin_bit_cast(v4f_t, w);
            });
        } else {
            const pair_t *src = src0 + HALO_V + j * (SUB_NEW / 2);
            // the stream is read once: non-temporal loads (round 3: read streams run 5-9 % faster with them on this part,
            // profiles/r03_stream_probe3.txt; -DREDIO_EXP_CHAIN_NT=0 builds the default-policy form for comparison)
#if REDIO_EXP_CHAIN_NT & 1
            static_for4<NLD>([&](auto I) { dst[I.value] = __builtin_nontemporal_load(src + 64 * I.value); });
#else
            static_for4<NLD>([&](auto I) { dst[I.value] = src[64 * I.value]; });
#endif
        }
    };
    auto park_from = [&](pair_t(&from)[NLD]) {
#if REDIO_EXP_ABLATE == 3 // timing-only experiment (tools/chain_variants.sh; results WRONG): the loads stay, the ten ds_write_b128 that
                          // park them in the image do not -- an upper bound on what staging by LDS-DMA instead of registers could buy
#pragma unroll
        for (int i = 0; i < NLD; ++i) { pair_t keep = from[i]; asm volatile("" : : "v"(keep)); }
#else
        static_for4<NLD>([&](auto I) { xs4[G::lds_index(HALO + 2 * (lane + 64 * I.value)) / 2] = samples(from[I.value]); });
#endif
    };

    // prologue: the head (the only halo this wave ever fetches) and the first sub-tile
    if (lane < HALO_V) xs4[G::lds_index(2 * lane) / 2] = samples(src0[0]);
    // PF2: the 60 twiddles of the transform's middle stages (Fft1knTw12: 15 per lane, 4 distinct sets by lane >> 4) sit in LDS behind the
    // image for the life of the wave, so the transform issues NO global load: its twiddle loads would queue behind the sub-tile request
    // (loads return in order) and make the transform wait for it
    constexpr int T12_OFF = G::lds_elems(SUB_OUT) > FFT1KN_LDS ? G::lds_elems(SUB_OUT) : FFT1KN_LDS;
    if constexpr (PF2 && !FIR_ONLY) {
        if (lane < 60) {
            const int k4 = lane / 15, i = lane % 15;
            const int idx = i < 3 ? 64 * k4 * (i + 1) : 16 * (k4 + 4 * ((i - 3) / 3)) * ((i - 3) % 3 + 1);
            ex[T12_OFF + lane] = tw[idx];
        }
    }
    fetch_to(pre, 0);
    park_from(pre);
    wave_lds_fence();

    float2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = make_float2(0.f, 0.f);
    // the 15 lane-dependent twiddles of the last two stages stay in registers for the life of the wave
    Fft1knTw34 t34;
    if (!FIR_ONLY && TWP) fft1kn_load_tw34(t34, lane, tw);
    // one sub-tile: request sub-tile j + AHEAD into `req`, FIR of the image, transform when a block is complete,
    // then park `parked` (= sub-tile j + 1, requested AHEAD - 1 sub-tiles ago; for AHEAD == 1 it is `req` itself) into the image.
    // S: the sub-tile's place in its block when that is a compile-time fact (PF2: the loop below is unrolled by a whole block, the
    // transform sits in step 3 unconditionally, and so do the request, the halo move and the park), -1 when it is not.
    auto step = [&](long j, auto ahead, auto place, pair_t(&req)[NLD], pair_t(&parked)[NLD]) {
        constexpr int AHEAD = decltype(ahead)::value, S = decltype(place)::value;
        const bool more = S >= 0 || j + 1 < nsub;
        if (S >= 0 || j + AHEAD < nsub) fetch_to(req, j + AHEAD);
        float2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);
        int lf = lane; // opaque per sub-tile: keeps LDS address arithmetic out of the loop's live set
        asm volatile("" : "+v"(lf));
        fir_lane_v<K, D, R, FUSED, CH>(xs4, lf, taps, acc);
        if (FIR_ONLY) { // 32 contiguous bytes per lane, 2 KiB per wave
            v4f_t *y4 = reinterpret_cast<v4f_t *>(out + (b0 * 4 + j) * SUB_OUT) + 2 * lane;
            y4[0] = v4f_t{acc[0].x, acc[0].y, acc[1].x, acc[1].y};
            y4[1] = v4f_t{acc[2].x, acc[2].y, acc[3].x, acc[3].y};
        } else {
#pragma unroll
            for (int i = 0; i < 12; ++i) a[i] = a[i + 4];
#pragma unroll
            for (int r = 0; r < R; ++r) a[12 + r] = acc[r];
        }
        wave_lds_fence(); // window reads done
        // the last HALO samples of this image are the first HALO samples of the next one
        v4f_t halo = v4f_t{0.f, 0.f, 0.f, 0.f};
        if (more && lf < HALO_V) halo = xs4[G::lds_index(SUB_NEW + 2 * lf) / 2];
        if (!FIR_ONLY && (S == 3 || (S < 0 && (j & 3) == 3))) { // a block is complete in registers: transform it, the image is scratch meanwhile
            wave_lds_fence();
            int ln = lane;
            asm volatile("" : "+v"(ln));
            Fft1knTw12 t12;
            if constexpr (PF2) {
                const float2 *T = ex + T12_OFF + 15 * (ln >> 4);
#pragma unroll
                for (int i = 0; i < 3; ++i) t12.a[i] = T[i];
#pragma unroll
                for (int i = 0; i < 12; ++i) t12.b[i] = T[3 + i];
            } else fft1kn_load_tw12(t12, ln, tw);
            if (!TWP) fft1kn_load_tw34(t34, ln, tw);
            fft1kn_wave_tw<false>(a, ex, tw, t12, t34, out + (b0 + (j >> 2)) * 1024, ln);
            wave_lds_fence();
        }
        if (more) {
            if (lf < HALO_V) xs4[G::lds_index(2 * lf) / 2] = halo;
            park_from(parked);
        }
        wave_lds_fence();
    };
    if constexpr (PF2) {
        fetch_to(pre, 1);
        using two = std::integral_constant<int, 2>;
#pragma unroll 1
        for (long j = 0; j < nsub; j += 4) { // one block per iteration
            step(j, two{}, std::integral_constant<int, 0>{}, pre2, pre);
            step(j + 1, two{}, std::integral_constant<int, 1>{}, pre, pre2);
            step(j + 2, two{}, std::integral_constant<int, 2>{}, pre2, pre);
            step(j + 3, two{}, std::integral_constant<int, 3>{}, pre, pre2);
        }
    } else {
#pragma unroll 1
        for (long j = 0; j < nsub; ++j) step(j, std::integral_constant<int, 1>{}, std::integral_constant<int, -1>{}, pre, pre);
    }
    if (dbg && lane == 0 && (long)blockIdx.x < dbg_cap) { // the caller's buffer holds dbg_cap records: later waves leave no stamp
        dbg[4 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0c;
        dbg[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - t0r;
        dbg[4 * blockIdx.x + 2] = t0r; // absolute start (100 MHz ticks)
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        dbg[4 * blockIdx.x + 3] = ((unsigned long long)xcc << 32) | hwid;
    }
}

// blocks per wavefront of a launch over nblocks blocks at WPS wavefronts per SIMD (the chain: 2; the FIR alone: 3)
long chain_v4_blocks_per_wave(long nblocks, int WPS)
{
    // Short runs in dispatch order, not one long run per residency slot: the wavefronts resident at any moment then read one
    // compact window of the stream (2048 x 4 blocks = 335 MB apart at most, instead of 2048 places spread over all of it).
    // Measured with non-temporal accesses 0.533 -> 0.517 ms per 2^28 samples (profiles/r03_chain_variants.txt); a run
    // re-fetches its 122-sample head: 0.6 % more reads at four blocks per run.
    long waves = 4L * WPS * num_cus();
    if (waves > nblocks) waves = nblocks;
    if (waves < 1) waves = 1;
    // ... and at least about six sets of wavefronts per launch: with fewer the last, partly filled set is most of a run's length of idle
    // slots (2^26 samples, 127 taps / 5 alone: 0.153-0.159 ms at four blocks per run, 0.131-0.139 at one; 63 taps / 5: 0.137 -> 0.117)
    long bpw = nblocks / (6 * waves);
    if (bpw < 1) bpw = 1;
    if (bpw > 4) bpw = 4;
    if (const char *e = measure_env("REDIO_CHAIN_BPW")) { const long v = atol(e); if (v >= 1) bpw = v; } // -DREDIO_MEASURE builds only: blocks per wavefront
    return bpw;
}

template <int K, int D, int WPS, int CH, bool FIR_ONLY = false, bool TWP = false, bool IN_U8 = false, bool PF2 = false>
static hipError_t launch_v4_t(const float2 *x, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused,
                              hipStream_t s, unsigned long long *dbg, long dbg_cap = 0)
{
    static_assert(WPS == 2 || WPS == 3, "two wavefronts per SIMD (the chain: its transform needs 197 registers) or three (the FIR alone: 94-145)");
    using G = FirGeomV<K, D, 4>;
    constexpr int ELEMS = (G::lds_elems(256) > FFT1KN_LDS ? G::lds_elems(256) : FFT1KN_LDS) + (PF2 ? 64 : 0); // PF2: the middle stages' twiddles behind the image
    constexpr size_t LDS_NEED = (size_t)ELEMS * sizeof(float2);
    static_assert(4 * WPS * LDS_NEED <= 160 * 1024, "4*WPS waves per CU");
    // The grid is exactly one wave per residency slot.  Ask for 1/(4*WPS) of the CU's LDS (minus a
    // margin for allocation granularity) so that a CU can NOT take more than 4*WPS waves: without this
    // the dispatcher packs up to 12-13 of these small workgroups on some CUs and leaves others
    // half empty, and the launch ends when the most crowded CU does (measured: wave lifetimes 320-570 us).
    constexpr size_t LDS = (160 * 1024 / (4 * WPS)) - 480 > LDS_NEED ? (160 * 1024 / (4 * WPS)) - 480 : LDS_NEED;
    static_assert((4 * WPS + 1) * LDS > 160 * 1024, "one more wave must not fit");
    const long bpw = chain_v4_blocks_per_wave(nblocks, WPS);
    const long grid = (nblocks + bpw - 1) / bpw;
    if (fused) hipLaunchKernelGGL((chain_v4_kernel<K, D, true, WPS, CH, FIR_ONLY, TWP, IN_U8, PF2>), dim3((unsigned)grid), dim3(64), LDS, s, x, taps, tw, out, nblocks, bpw, dbg, dbg_cap);
    else hipLaunchKernelGGL((chain_v4_kernel<K, D, false, WPS, CH, FIR_ONLY, TWP, IN_U8, PF2>), dim3((unsigned)grid), dim3(64), LDS, s, x, taps, tw, out, nblocks, bpw, dbg, dbg_cap);
    return hipGetLastError();
}

hipError_t launch_chain_v4(const float2 *x, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused,
                           hipStream_t s, unsigned long long *dbg, long dbg_cap)
{
    // (three wavefronts per SIMD were measured again in round 3 with the new access policy: the transform's registers spill at
    // 168 per lane, 0.56-0.60 against 0.517 ms)
#ifdef REDIO_MEASURE // three wavefronts per SIMD without the resident last-stage twiddles (168 registers, 80 bytes of scratch): measured in round 4, see DESIGN.md 5.1
    if (measure_env("REDIO_CHAIN_WPS3")) return launch_v4_t<127, 5, 3, 8, false, false>(x, taps, tw, out, nblocks, fused, s, dbg, dbg_cap);
    // two sub-tiles of loads in flight (round 6): A/B against the product form in one process, tools/chain_pf2_ab.py
    if (measure_env("REDIO_CHAIN_PF2")) return launch_v4_t<127, 5, 2, 8, false, true, false, true>(x, taps, tw, out, nblocks, fused, s, dbg, dbg_cap);
#endif
    return launch_v4_t<127, 5, 2, 8, false, true>(x, taps, tw, out, nblocks, fused, s, dbg, dbg_cap); // last-stage twiddles resident
}

// u8 I/Q bytes in (4-byte aligned), spectra out: data_to_samples -> 127-tap FIR / 5 -> 1024-point FFT in one kernel
hipError_t launch_chain_v4_u8(const void *bytes, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused, hipStream_t s)
{
    return launch_v4_t<127, 5, 2, 8, false, true, true>((const float2 *)bytes, taps, tw, out, nblocks, fused, s, nullptr); // 3 waves per SIMD: 0.52 against 0.46 ms
}

hipError_t launch_chain_v4_shape_u8(int K, int D, const void *bytes, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused,
                                    hipStream_t s)
{
    const float2 *x = (const float2 *)bytes;
    if (K == 63 && D == 5) return launch_v4_t<63, 5, 2, 8, false, true, true>(x, taps, tw, out, nblocks, fused, s, nullptr);
    if (K == 127 && D == 1) return launch_v4_t<127, 1, 2, 8, false, true, true>(x, taps, tw, out, nblocks, fused, s, nullptr);
    if (K == 63 && D == 1) return launch_v4_t<63, 1, 2, 8, false, true, true>(x, taps, tw, out, nblocks, fused, s, nullptr);
    if (K == 127 && D == 3) return launch_v4_t<127, 3, 2, 8, false, true, true>(x, taps, tw, out, nblocks, fused, s, nullptr);
    return hipErrorNotSupported;
}

// the same kernel for the other tap / decimation pairs with a fused build (K - D even, image within the per-wave LDS budget)
hipError_t launch_chain_v4_shape(int K, int D, const float2 *x, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused,
                                 hipStream_t s)
{
    if (K == 63 && D == 5) return launch_v4_t<63, 5, 2, 8, false, true>(x, taps, tw, out, nblocks, fused, s, nullptr);
    if (K == 127 && D == 1) return launch_v4_t<127, 1, 2, 8, false, true>(x, taps, tw, out, nblocks, fused, s, nullptr);
    if (K == 63 && D == 1) return launch_v4_t<63, 1, 2, 8, false, true>(x, taps, tw, out, nblocks, fused, s, nullptr);
    if (K == 127 && D == 3) return launch_v4_t<127, 3, 2, 8, false, true>(x, taps, tw, out, nblocks, fused, s, nullptr);
    return hipErrorNotSupported;
}

// stand-alone decimating FIR on whole 1024-output blocks (16-byte aligned cf32 in and out) for the shapes the chain kernel is
// built for: the chain's data path (wave-private image, halo carried in LDS, register prefetch, taps in SGPRs), storing the
// decimated samples instead of transforming them.  hipErrorNotSupported: no such instantiation (the tiled kernels run).
hipError_t launch_fir_v4(int K, int D, const float2 *x, const float *taps, float2 *y, long nblocks, bool fused, hipStream_t s)
{
    // Three wavefronts per SIMD (round 5): without the transform these instantiations need 94-145 registers, and the third wavefront
    // hides more of the window reads' and the stream's latency -- 2^26 samples: 127 / 5 0.1189 -> 0.1109 ms (67.7 -> 72.6 % of 8 TB/s),
    // 63 / 5 0.1007 -> 0.0983, 63 / 1 0.2561 -> 0.2441; 2^28: 0.4901 -> 0.4818, 0.4655 -> 0.4664, 0.9973 -> 0.9688; the reference-rounding
    // builds 4-6 % (profiles/r05_fir_three_waves.txt; interleaved A/B in one process, bit-identical).  REDIO_FIR_WPS2: measurement builds.
    if (!measure_env("REDIO_FIR_WPS2")) {
        if (K == 127 && D == 5) return launch_v4_t<127, 5, 3, 8, true>(x, taps, nullptr, y, nblocks, fused, s, nullptr);
        if (K == 63 && D == 5) return launch_v4_t<63, 5, 3, 8, true>(x, taps, nullptr, y, nblocks, fused, s, nullptr);
        if (K == 63 && D == 1) return launch_v4_t<63, 1, 3, 8, true>(x, taps, nullptr, y, nblocks, fused, s, nullptr);
        if (K == 127 && D == 3) return launch_v4_t<127, 3, 3, 8, true>(x, taps, nullptr, y, nblocks, fused, s, nullptr);
    }
#ifdef REDIO_MEASURE
    if (K == 127 && D == 5) return launch_v4_t<127, 5, 2, 8, true>(x, taps, nullptr, y, nblocks, fused, s, nullptr);
    if (K == 63 && D == 5) return launch_v4_t<63, 5, 2, 8, true>(x, taps, nullptr, y, nblocks, fused, s, nullptr);
    if (K == 63 && D == 1) return launch_v4_t<63, 1, 2, 8, true>(x, taps, nullptr, y, nblocks, fused, s, nullptr);
#endif
    // (127 taps, no decimation: VALU-bound, and the 256-thread tiled kernel with 8 outputs per lane is faster -- 0.44 against 0.56 ms per 2^26 samples;
    //  127 taps / 3 runs here since round 5 with three wavefronts per SIMD: 0.1589 against the chunked tiled kernel's 0.1713 ms, profiles/r05_fir_mid_shapes.txt)
    return hipErrorNotSupported;
}

} // namespace redio
