// fir_core.h -- register-blocked valid-mode FIR (cross-correlation) for one lane.
//
// Arithmetic contract: dsputils::convolve, src/dsputils/src/dsputils.rs:30-32:
//   out[i] = fold(0.0, +) over j of u[i+j]*v[j]  -- taps NOT reversed, strict left-to-right order.
// Each accumulator below receives its products in ascending tap order j, so with FUSED=false
// (separate rounded multiply and add) a result is bit-identical to the reference fold, and with
// FUSED=true it is bit-identical to the same fold written with fmaf.  Decimation keeps out[D*i].
//
// Work split: a lane owns R consecutive kept outputs and walks its (R-1)*D+K input samples once;
// sample m feeds accumulator r with tap j = m - r*D.  The walk is fully unrolled so every tap index
// is a compile-time constant (taps are wave-uniform and live in SGPRs), and one LDS read feeds up to
// min(R, ceil(K/D)) multiply-adds.
//
// Host-compilable (tests/emu).
#pragma once
#include "redio_device.h"
#ifndef REDIO_EXP_ABLATE
#define REDIO_EXP_ABLATE 0
#endif
#include <type_traits>

namespace redio {

template <int K, int D, int R>
struct FirGeom {
    static constexpr int LSTR = R * D;             // input samples between neighbouring lanes
    static constexpr int SPAN = (R - 1) * D + K;   // input samples one lane touches
    static constexpr bool PAD = (LSTR % 2) == 0;   // LDS lane stride must be odd (in elements) so
                                                   // that 32 lanes hit 32 different bank pairs
    // LDS element index of tile-relative input sample n
    RD_HD static constexpr int lds_index(int n) { return PAD ? n + n / LSTR : n; }
    RD_HD static constexpr int tile_in(int tile_out) { return (tile_out - 1) * D + K; }
    RD_HD static constexpr int lds_elems(int tile_out) { return lds_index(tile_in(tile_out) - 1) + 1; }
};

// lane_first = tile-relative index of the first input sample of this lane's first output
// (= lane_slot * R * D).  xs is the LDS image written with FirGeom::lds_index.
template <typename T, int K, int D, int R, bool FUSED, typename LdsPtr, typename TapPtr>
RD_HD void fir_lane(LdsPtr xs, int lane_slot, TapPtr h, T (&acc)[R])
{
    using G = FirGeom<K, D, R>;
    const int base = lane_slot * (G::LSTR + (G::PAD ? 1 : 0)); // lds_index(lane_slot*LSTR)
#pragma unroll
    for (int m = 0; m < G::SPAN; ++m) {
        const T xv = xs[base + G::lds_index(m)];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = m - r * D;
            if (j >= 0 && j < K) acc[r] = mac<FUSED>(xv, h[j], acc[r]);
        }
    }
}

} // namespace redio

// ---------------------------------------------------------------------------------------------
// Variant for interleaved cf32 with 16-byte LDS reads: a lane's window is read two samples at a time
// (ds_read_b128), which needs every lane base 16-byte aligned and, for the four 16-lane groups of
// ds_read_b128 to be conflict-free, a lane stride S (in float2) with S/2 odd.
// ---------------------------------------------------------------------------------------------
namespace redio {

template <int K, int D, int R>
struct FirGeomV {
    static constexpr int LSTR = R * D;
    static_assert(LSTR % 2 == 0, "16-byte windows need an even lane stride");
    static constexpr int SPAN = (R - 1) * D + K;
    static constexpr int PADN = ((LSTR / 2) % 2 == 1) ? 0 : 2; // make (LSTR+PADN)/2 odd
    static constexpr int LANE_STRIDE = LSTR + PADN;
    RD_HD static constexpr int lds_index(int n) { return n + PADN * (n / LSTR); }
    RD_HD static constexpr int tile_in(int tile_out) { return (tile_out - 1) * D + K; }
    // float2 elements, rounded up to an even count (whole float4s)
    RD_HD static constexpr int lds_elems(int tile_out) { return (lds_index(tile_in(tile_out) - 1) + 2) & ~1; }
};

// xs4: the LDS image viewed as 16-byte elements (two cf32 samples each), written with
// FirGeomV::lds_index.  The window is consumed in chunks of CH reads with the next chunk's reads
// issued before the current chunk's multiply-adds and a scheduling barrier between chunks: LDS
// latency hides behind ~100 packed FMAs while only two chunks of samples are ever live in
// registers (an unconstrained schedule hoists dozens of reads and spills).
template <int N, typename F>
RD_HD void fir_static_for(F &&f)
{
    if constexpr (N > 0) {
        fir_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// Taps needed by chunk c: samples m in [2*CH*c, 2*CH*(c+1)) feed output r with tap j = m - r*D, so
// j spans [2*CH*c - (R-1)*D, 2*CH*(c+1) - 1], clipped to [0, K).
template <int K, int D, int R, int CH>
struct FirChunkTaps {
    static constexpr int NT = 2 * CH + (R - 1) * D; // taps a chunk can touch
    RD_HD static constexpr int lo(int c) { return 2 * CH * c - (R - 1) * D; }
};

template <int c, int K, int D, int R, bool FUSED, int CH, typename Q, typename Lds4Ptr, typename TapPtr>
RD_HD void fir_chunks_v(Lds4Ptr xs4, int base4, TapPtr h, Q (&q)[2][CH],
                        float (&hb)[2][FirChunkTaps<K, D, R, CH>::NT], float2 (&acc)[R])
{
    using G = FirGeomV<K, D, R>;
    using T = FirChunkTaps<K, D, R, CH>;
    constexpr int NRD = (G::SPAN + 1) / 2; // 16-byte reads in a window
    constexpr int NCH = (NRD + CH - 1) / CH;
    if constexpr (c < NCH) {
        // everything this chunk consumes was requested one chunk ago; naming one tap and one sample
        // here makes the single lgkmcnt(0) land BEFORE the next chunk's requests are issued
        RD_PIN_SV(hb[c & 1][T::NT - 1], q[c & 1][0]);
        fir_static_for<CH>([&](auto I) { // the next chunk's samples (LDS) ...
            constexpr int i = (c + 1) * CH + I.value;
#if REDIO_EXP_ABLATE == 1 // timing-only experiment (tools/ablate.sh), never in the product build: half the window reads
            if constexpr (i < NRD && (I.value & 1)) q[(c + 1) & 1][I.value] = q[(c + 1) & 1][I.value - 1];
            else
#endif
            if constexpr (i < NRD) q[(c + 1) & 1][I.value] = xs4[base4 + G::lds_index(2 * i) / 2];
        });
        fir_static_for<T::NT>([&](auto J) { // ... and taps (scalar cache)
            constexpr int j = T::lo(c + 1) + J.value;
            if constexpr (c + 1 < NCH && j >= 0 && j < K) hb[(c + 1) & 1][J.value] = h[j];
        });
        fir_static_for<CH>([&](auto I) {
            constexpr int m = 2 * (c * CH + I.value);
            if constexpr (m < G::SPAN) {
                const Q v = q[c & 1][I.value];
                const float2 x0 = make_float2(v.x, v.y), x1 = make_float2(v.z, v.w);
                fir_static_for<R>([&](auto RR) {
                    constexpr int r = RR.value;
#if REDIO_EXP_ABLATE == 2 // timing-only experiment: half the multiply-adds (two of the four accumulators)
                    if constexpr (r & 1) return;
#endif
                    constexpr int j0 = m - r * D, j1 = m + 1 - r * D;
                    if constexpr (j0 >= 0 && j0 < K) acc[r] = mac<FUSED>(x0, hb[c & 1][j0 - T::lo(c)], acc[r]);
                    if constexpr (j1 >= 0 && j1 < K && m + 1 < G::SPAN) acc[r] = mac<FUSED>(x1, hb[c & 1][j1 - T::lo(c)], acc[r]);
                });
            }
        });
        // pin the accumulators: without a data dependency the compiler sinks every multiply-add below
        // the last read and the chunking is lost
#pragma unroll
        for (int r = 0; r < R; ++r) RD_PIN_F2(acc[r]);
        RD_SCHED_BARRIER();
        fir_chunks_v<c + 1, K, D, R, FUSED, CH>(xs4, base4, h, q, hb, acc);
    }
}

template <int K, int D, int R, bool FUSED, int CH = 8 /* 16-byte reads per chunk */, typename Lds4Ptr, typename TapPtr>
RD_HD void fir_lane_v(Lds4Ptr xs4, int lane_slot, TapPtr h, float2 (&acc)[R])
{
    using G = FirGeomV<K, D, R>;
    using Q = typename std::remove_cv<typename std::remove_reference<decltype(xs4[0])>::type>::type;
    constexpr int NRD = (G::SPAN + 1) / 2;
    using T = FirChunkTaps<K, D, R, CH>;
    const int base4 = lane_slot * (G::LANE_STRIDE / 2);
    Q q[2][CH];
    float hb[2][T::NT];
    fir_static_for<T::NT>([&](auto J) { // last slot is always written so that it can be pinned
        constexpr int j = T::lo(0) + J.value;
        hb[0][J.value] = (j >= 0 && j < K) ? h[j] : 0.0f;
        hb[1][J.value] = 0.0f;
    });
    fir_static_for<CH>([&](auto I) {
        constexpr int i = I.value;
        if constexpr (i < NRD) q[0][i] = xs4[base4 + G::lds_index(2 * i) / 2];
    });
    fir_chunks_v<0, K, D, R, FUSED, CH>(xs4, base4, h, q, hb, acc);
}

} // namespace redio

// ---------------------------------------------------------------------------------------------
// REAL samples without decimation, two outputs per packed multiply-add (round 6).  A lane owns R consecutive outputs as R/2
// accumulator PAIRS (out[2p], out[2p+1]); tap j feeds pair p with the sample pair (x[2p + j], x[2p + 1 + j]) -- one v_pk_fma_f32 (or
// v_pk_mul + v_pk_add) whose two halves are two outputs' strict left folds (dsputils.rs:31), each still receiving its products in
// ascending tap order.  A pair that starts on an odd sample is no aligned 64-bit register pair of the even ones: the scalar lane
// program above leaves that to the compiler, which builds them with one v_mov per pair (49 moves per 252 packed multiply-adds in
// the 63-tap kernel).  Here the tile is stored TWICE in LDS, as E2[m] = (x[2m], x[2m+1]) and as O2[m] = (x[2m+1], x[2m+2]), so that
// every pair is one aligned 8-byte read and the multiply-adds are the only vector instructions of the body.
// Image element m (a float2) of either copy sits at lds_index(m): lane stride R/2 float2, one pad element per lane when that is even
// (odd stride: the 32 lanes of a ds_read_b64 hit 32 different bank pairs).
// ---------------------------------------------------------------------------------------------
namespace redio {

template <int K, int R>
struct FirGeomPairs {
    static_assert(R % 2 == 0, "pairs of outputs");
    static constexpr int LSTR = R / 2;                    // float2 elements between neighbouring lanes
    static constexpr bool PAD = (LSTR % 2) == 0;
    static constexpr int LANE_STRIDE = LSTR + (PAD ? 1 : 0);
    static constexpr int NPAIR = R / 2 + (K - 1) / 2 + ((K - 1) % 2 ? 1 : 0); // pair slots m one lane reads from EACH copy: m < R/2 + ceil((K-1)/2)
    RD_HD static constexpr int lds_index(int m) { return PAD ? m + m / LSTR : m; }
    RD_HD static constexpr int tile_in(int tile_out) { return tile_out - 1 + K; }
    // float2 elements of ONE copy for a tile of tile_out outputs: the tile is staged in whole 16-byte loads (four samples = two
    // elements of either copy), so a copy holds 2 * ceil(tile_in / 4) elements
    RD_HD static constexpr int copy_elems(int tile_out) { return lds_index(2 * ((tile_in(tile_out) + 3) / 4) - 1) + 1; }
};

// e2 / o2: the two copies of the image (float2 views); acc[p] = (out[2p], out[2p+1]) of this lane
template <int K, int R, bool FUSED, typename LdsPtr, typename TapPtr>
RD_HD void fir_lane_pairs(LdsPtr e2, LdsPtr o2, int lane_slot, TapPtr h, float2 (&acc)[R / 2])
{
    using G = FirGeomPairs<K, R>;
    const int base = lane_slot * G::LANE_STRIDE;
#pragma unroll
    for (int m = 0; m < G::NPAIR; ++m) {
        const float2 e = e2[base + G::lds_index(m)]; // (x[2m], x[2m+1]) relative to the lane's first output
        const float2 o = o2[base + G::lds_index(m)]; // (x[2m+1], x[2m+2])
#pragma unroll
        for (int p = 0; p < R / 2; ++p) {
            const int j = 2 * (m - p);
            if (j >= 0 && j < K) acc[p] = mac<FUSED>(e, h[j], acc[p]);
            if (j + 1 >= 0 && j + 1 < K) acc[p] = mac<FUSED>(o, h[j + 1], acc[p]);
        }
    }
}

} // namespace redio
