// fft_pair.h -- device side of the two-column ("pair") tile program of the four-stage passes (maps and LDS images: fft_big_core.h).
// Included by fft_kernels.hip after its tile helpers (wave_lds_fence, big_ld_once / big_st_once, f64k constants).
//
// One wavefront per 256-row x 16-column tile, 64 points per lane as before, but a lane holds two adjacent columns (two adjacent rows
// on the transposed side): 32 global_load_dwordx4 + 32 global_store_dwordx4 per lane and pass instead of 64 + 64 dwordx2, and
// 32 ds_write_b128 + 32 ds_read_b128 per regrouping instead of 64 + 64 b64.  Same butterflies in the same order with the same
// twiddles: the results are the bits of the one-column program (kissfft's, oracle/oracle_kiss.c).
#pragma once
#include "fft_big_core.h"

namespace redio {

typedef float pw_v4 __attribute__((ext_vector_type(4)));
// the caller's buffers (block b of the stream starts at x + b hop: 8-byte aligned only when hop is odd)
typedef float pw_v4u __attribute__((ext_vector_type(4), aligned(8)));

// pw_ld / pw_ld_mid / pw_st also reach the CALLER's `out` (a stand-alone transform of 32768 points or more works in place in it): declared
// 8-byte aligned like every caller-owned address -- the same global_load / store_dwordx4, and no 16-byte promise the caller never made
__device__ __forceinline__ float4 pw_ld(const float2 *p) { const pw_v4u v = *reinterpret_cast<const pw_v4u *>(p); return make_float4(v.x, v.y, v.z, v.w); }
// an intermediate that a pass reads exactly once (the work buffers of the multi-pass schemes): -DREDIO_EXP_PW_NT_MID=1 reads it non-temporally
#ifndef REDIO_EXP_PW_NT_MID
#define REDIO_EXP_PW_NT_MID 0
#endif
__device__ __forceinline__ float4 pw_ld_mid(const float2 *p)
{
#if REDIO_EXP_PW_NT_MID
    const pw_v4u v = __builtin_nontemporal_load(reinterpret_cast<const pw_v4u *>(p));
#else
    const pw_v4u v = *reinterpret_cast<const pw_v4u *>(p);
#endif
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void pw_st(float2 *p, float2 lo, float2 hi) { const pw_v4u v = {lo.x, lo.y, hi.x, hi.y}; *reinterpret_cast<pw_v4u *>(p) = v; }
// data touched ONCE by a multi-pass transform (the caller's input / output): non-temporal, like big_ld_once / big_st_once
__device__ __forceinline__ float4 pw_ld_once(const float2 *p)
{
#if REDIO_EXP_BIG_NT & 1
    const pw_v4u v = __builtin_nontemporal_load(reinterpret_cast<const pw_v4u *>(p));
#else
    const pw_v4u v = *reinterpret_cast<const pw_v4u *>(p);
#endif
    return make_float4(v.x, v.y, v.z, v.w);
}
// the same 16 bytes with the default cache policy to a caller-owned (8-byte aligned) address: the valid outputs of the overlap-save sizes whose passes
// run launch after launch over a chunk (32768, 131072 ...) -- measured: non-temporal stores there 3.18 against 2.74 ms (32768-point blocks) and 3.86
// against 3.37 (131072), while the 65536-point scheme, whose step launches keep two chunks of intermediates in flight, loses 17 % WITHOUT them
// (profiles/r04_ovsave_output_store_policy_ab.txt)
__device__ __forceinline__ void pw_st_out(float2 *p, float2 lo, float2 hi)
{
    const pw_v4u v = {lo.x, lo.y, hi.x, hi.y};
    *reinterpret_cast<pw_v4u *>(p) = v;
}
__device__ __forceinline__ void pw_st_once(float2 *p, float2 lo, float2 hi)
{
    const pw_v4u v = {lo.x, lo.y, hi.x, hi.y};
#if REDIO_EXP_BIG_NT & 2
    __builtin_nontemporal_store(v, reinterpret_cast<pw_v4u *>(p));
#else
    *reinterpret_cast<pw_v4u *>(p) = v;
#endif
}

// the four rounds of a regrouping; every lane has written before any lane reads, and has read before the image is written again
template <typename G>
__device__ __forceinline__ void pw_exchange_plain(float2 (&a)[2][2][16], float2 (&b)[2][2][16], float4 *Lw, int lane)
{
    pw_plain_write<0, 0>(a, Lw, lane); wave_lds_fence(); pw_plain_read<G, 0, 0>(b, Lw, lane); wave_lds_fence();
    pw_plain_write<0, 1>(a, Lw, lane); wave_lds_fence(); pw_plain_read<G, 0, 1>(b, Lw, lane); wave_lds_fence();
    pw_plain_write<1, 0>(a, Lw, lane); wave_lds_fence(); pw_plain_read<G, 1, 0>(b, Lw, lane); wave_lds_fence();
    pw_plain_write<1, 1>(a, Lw, lane); wave_lds_fence(); pw_plain_read<G, 1, 1>(b, Lw, lane); wave_lds_fence();
}
template <typename G>
__device__ __forceinline__ void pw_exchange_tr(float2 (&a)[2][2][16], float2 (&b)[2][2][16], float4 *Lw, int lane)
{
    pw_tr_write<0, 0>(a, Lw, lane); wave_lds_fence(); pw_tr_read<G, 0, 0>(b, Lw, lane); wave_lds_fence();
    pw_tr_write<0, 1>(a, Lw, lane); wave_lds_fence(); pw_tr_read<G, 0, 1>(b, Lw, lane); wave_lds_fence();
    pw_tr_write<1, 0>(a, Lw, lane); wave_lds_fence(); pw_tr_read<G, 1, 0>(b, Lw, lane); wave_lds_fence();
    pw_tr_write<1, 1>(a, Lw, lane); wave_lds_fence(); pw_tr_read<G, 1, 1>(b, Lw, lane); wave_lds_fence();
}

// ---- in-place four-stage pass on rows m_lo apart: load, stages 0-1, plain regrouping, stages 2-3 --------------------------------------
// base: the tile's origin (row 0, column 0); l0: position of column 0 inside the sub-length m_lo (the twiddle index of a column);
// T: the pass's ORDERED twiddle copy (in a pair build fftbig_tables_build lays the four-stage passes' copies out that way: a lane's two
// columns are two neighbouring entries, one 16-byte load per twiddle pair).  On return b[x][e][j] = row (q + 8 x) + 16 j of column 2 cp + e.
template <bool INV>
__device__ __forceinline__ void pw_mid_stages(float2 (&b)[2][2][16], const float2 *base, long m_lo, unsigned l0, const float2 *__restrict__ T,
                                              float4 *Lw, int lane)
{
    const int cp = lane & 7, q = lane >> 3;
    const unsigned ml = (unsigned)m_lo;
    float2 a[2][2][16];
    const unsigned lo = (unsigned)pw_mid_ld(m_lo, q, cp, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 v = pw_ld_mid((base + pw_mid_ld(m_lo, 0, 0, i, j)) + lo);
            a[i][0][j] = make_float2(v.x, v.y); a[i][1][j] = make_float2(v.z, v.w);
        }
    RD_SCHED_BARRIER();
    {
        FftTw15 T0, T1;
        big_tw15x2(T0, T1, tw_pair_stage(T, ml, 0), tw_pair_stage(T, ml, 1), l0 + 2u * cp, ml, 0u, 1u);
        RD_SCHED_BARRIER();
        macro16_apply<INV>(a[0][0], T0); macro16_apply<INV>(a[1][0], T0);
        macro16_apply<INV>(a[0][1], T1); macro16_apply<INV>(a[1][1], T1);
    }
    pw_exchange_plain<PwGroupsLinear>(a, b, Lw, lane);
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        FftTw15 T0, T1;
        big_tw15x2(T0, T1, tw_pair_stage(T, ml, 2), tw_pair_stage(T, ml, 3), l0 + 2u * cp, ml, (unsigned)(q + 8 * x), 16u);
        RD_SCHED_BARRIER();
        macro16_apply<INV>(b[x][0], T0);
        macro16_apply<INV>(b[x][1], T1);
    }
}

// the in-place pass as a whole (fftbig_mid_kernel); vout: overlap-save, this was the last pass -- 1/N and only the hop valid outputs, packed
template <bool INV>
__device__ __forceinline__ void pw_mid_tile(float2 *base, long m_lo, unsigned l0, const float2 *__restrict__ T, float4 *Lw, int lane,
                                            float2 *__restrict__ vout_blk, long e0, long hop, float scale)
{
    const int cp = lane & 7, q = lane >> 3;
    float2 b[2][2][16];
    pw_mid_stages<INV>(b, base, m_lo, l0, T, Lw, lane);
    const unsigned lo = (unsigned)pw_mid_st(m_lo, q, cp, 0, 0);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (vout_blk) { // e0: position of the tile's origin inside the block
                const long e = e0 + pw_mid_st(m_lo, 0, 0, x, j) + lo;
                const float2 v0 = make_float2(mul_rn(b[x][0][j].x, scale), mul_rn(b[x][0][j].y, scale));
                const float2 v1 = make_float2(mul_rn(b[x][1][j].x, scale), mul_rn(b[x][1][j].y, scale));
                if (e + 1 < hop) pw_st_out(vout_blk + e, v0, v1);
                else if (e < hop) vout_blk[e] = v0;
            } else pw_st((base + pw_mid_st(m_lo, 0, 0, x, j)) + lo, b[x][0][j], b[x][1][j]);
        }
}

// ---- gather pass of a 4^L-point transform: digit-reversed load, stages 0-1, transposed regrouping, stages 2-3, working-order store -----
// in_blk + 16 c: source columns 16 c .. 16 c + 15; T1: the gather pass's ordered twiddle copy (sub-lengths 1 .. 256)
// the part up to phase B: src = the tile's source origin (row 0, column 0), S = source row stride, hsrc = the spectrum at the same origin or null
template <bool INV, bool MULH>
__device__ __forceinline__ void pw_first_stages(float2 (&b)[2][2][16], const float2 *src, const float2 *__restrict__ hsrc, long S, const float2 *__restrict__ T1,
                                                float4 *Lw, int lane)
{
    const int cp = lane & 7, q = lane >> 3;
    float2 a[2][2][16];
    const unsigned lo_src = (unsigned)pw_first_ld(S, q, cp, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float4 v = pw_ld_once((src + pw_first_ld(S, 0, 0, i, j)) + lo_src);
            a[i][0][j] = make_float2(v.x, v.y); a[i][1][j] = make_float2(v.z, v.w);
        }
    if (MULH && hsrc) { // overlap-save: the spectrum product on the way in (wave-uniform branch; MULH = false: a build of the pass without it)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int jh = 0; jh < 16; jh += 8) { // eight pairs of spectrum taps per batch of loads (sixteen spill: 128 sample registers are live)
                float4 hv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) hv[j] = pw_ld((hsrc + pw_first_ld(S, 0, 0, i, jh + j)) + lo_src);
                RD_SCHED_BARRIER();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    a[i][0][jh + j] = cmul_rn(a[i][0][jh + j], make_float2(hv[j].x, hv[j].y));
                    a[i][1][jh + j] = cmul_rn(a[i][1][jh + j], make_float2(hv[j].z, hv[j].w));
                }
            }
    }
    RD_SCHED_BARRIER();
    {
        FftTw15 T0; // sub-lengths 1 and 4: the twiddle depends on the row only (wave-uniform)
        big_tw15(T0, tw_ordered_stage(T1, 1u, 0), tw_ordered_stage(T1, 1u, 1), 0u, 1u, 0u, 1u);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int e = 0; e < 2; ++e) macro16_apply<INV>(a[i][e], T0);
    }
    pw_exchange_tr<PwGroupsLinear>(a, b, Lw, lane);
    const int sp = lane & 7;
    {
        FftTw15 T0, Tb; // sub-lengths 16 and 64: index = the row inside the 256-row transform, 2 sp + e (+ 16 u): neighbouring entries
        big_tw15x2(T0, Tb, tw_pair_stage_u(T1, 1u, 2), tw_pair_stage_u(T1, 1u, 3), 0u, 1u, (unsigned)(2 * sp), 16u);
        RD_SCHED_BARRIER();
#pragma unroll
        for (int x = 0; x < 2; ++x) { macro16_apply<INV>(b[x][0], T0); macro16_apply<INV>(b[x][1], Tb); }
    }
}
template <bool INV, bool MULH = true>
__device__ __forceinline__ void pw_first_tile(const float2 *in_blk, float2 *out_blk, int L, unsigned c, int lane, float4 *Lw,
                                              const float2 *__restrict__ mulH, const float2 *__restrict__ T1)
{
    const long S = 1l << (2 * L - 8); // source row stride
    float2 b[2][2][16];
    pw_first_stages<INV, MULH>(b, in_blk + 16 * c, mulH ? mulH + 16 * c : nullptr, S, T1, Lw, lane);
    const int sp = lane & 7, qq = lane >> 3;
    unsigned rc = 0; // digit reversal of c over L - 6 digits
    for (int d = 0, cc = (int)c; d < L - 6; ++d, cc >>= 2) rc = (rc << 2) | (cc & 3);
    float2 *dst = out_blk + 256l * rc;
    const unsigned lo_dst = (unsigned)pw_first_st(L, qq, sp, 0, 0);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int j = 0; j < 16; ++j) pw_st((dst + pw_first_st(L, 0, 0, x, j)) + lo_dst, b[x][0][j], b[x][1][j]);
}

// ---- fifth stage across the four wavefronts of a workgroup (fft_big_core.h): four rounds, one workgroup barrier each -----------------------
// X: two images of PW_X5_UNITS; t4: the stage's ordered twiddle copy; k0_of(x, jh): twiddle index of the lane's first row of the round
// (the second is dk further); store(x, jh, v): v[jp][n][e] = quarter n' = n of row jp after the stage
template <bool INV, typename TP, typename KFn, typename StFn>
__device__ __forceinline__ void pw_x5_rounds(const float2 (&b)[2][2][16], float4 *X, int lane, int w, TP t4, unsigned dk, KFn k0_of, StFn store)
{
#define REDIO_X5_ROUND(XX, JH)                                                     \
    {                                                                              \
        float4 *Xi = X + ((((2 * XX + JH) & 1) != 0) ? PW_X5_UNITS : 0);           \
        pw_x5_write<XX, JH>(b, Xi, lane, w);                                       \
        __syncthreads();                                                           \
        float2 v[2][4][2];                                                         \
        pw_x5_read(v, Xi, lane, w);                                                \
        pw_x5_stage<INV>(v, t4, k0_of(XX, JH), dk);                                \
        store(XX, JH, v);                                                          \
    }
    REDIO_X5_ROUND(0, 0) REDIO_X5_ROUND(0, 1) REDIO_X5_ROUND(1, 0) REDIO_X5_ROUND(1, 1)
#undef REDIO_X5_ROUND
}

// in-place five-stage pass on rows m_lo apart: tile = origin of the 1024-row x 16-column tile, wavefront w owns rows 256 w .. 256 w + 255
template <bool INV>
__device__ __forceinline__ void pw_mid5_tile(float2 *tile, long m_lo, unsigned l0, const float2 *__restrict__ T, float4 *X, int lane, int w,
                                             float2 *__restrict__ vout_blk, long e0, long hop, float scale)
{
    const int cp = lane & 7, q = lane >> 3;
    float2 b[2][2][16];
    pw_mid_stages<INV>(b, tile + 256 * m_lo * w, m_lo, l0, T, X + 512 * w, lane);
    const unsigned ml = (unsigned)m_lo, lo = (unsigned)(m_lo * q + 2 * cp);
    pw_x5_rounds<INV>(b, X, lane, w, tw_pair_stage(T, ml, 4), 16u * ml,
        [&](int x, int jh) { return l0 + 2u * cp + ml * (unsigned)(q + 8 * x + 16 * (8 * jh + 2 * w)); },
        [&](int x, int jh, float2 (&v)[2][4][2]) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const long r = m_lo * (256 * n + 8 * x + 16 * (8 * jh + 2 * w + jp)); // + the lane's m_lo q + 2 cp
                    if (vout_blk) { // overlap-save: this was the last pass; 1/N and only the hop valid outputs, packed
                        const long e = e0 + r + lo;
                        const float2 v0 = make_float2(mul_rn(v[jp][n][0].x, scale), mul_rn(v[jp][n][0].y, scale));
                        const float2 v1 = make_float2(mul_rn(v[jp][n][1].x, scale), mul_rn(v[jp][n][1].y, scale));
                        if (e + 1 < hop) pw_st_out(vout_blk + e, v0, v1);
                        else if (e < hop) vout_blk[e] = v0;
                    } else pw_st((tile + r) + lo, v[jp][n][0], v[jp][n][1]);
                }
        });
}

// gather pass with five stages (4^L points): the workgroup's tile is 1024 source rows (N / 1024 apart) x 16 source columns; wavefront w
// runs the 256-row gather program on source rows 4 rho + w (one 256-point sub-transform), then the fifth stage across the wavefronts
template <bool INV, bool MULH>
__device__ __forceinline__ void pw_first5_tile(const float2 *in_blk, float2 *out_blk, int L, unsigned c, int lane, int w, float4 *X,
                                               const float2 *__restrict__ mulH, const float2 *__restrict__ T1)
{
    const long S = 1l << (2 * L - 8), S5 = 1l << (2 * L - 10);
    float2 b[2][2][16];
    pw_first_stages<INV, MULH>(b, in_blk + 16 * c + S5 * w, mulH ? mulH + 16 * c + S5 * w : nullptr, S, T1, X + 512 * w, lane);
    const int sp = lane & 7, qq = lane >> 3;
    unsigned rc = 0; // digit reversal of c over L - 7 digits
    for (int d = 0, cc = (int)c; d < L - 7; ++d, cc >>= 2) rc = (rc << 2) | (cc & 3);
    float2 *dst = out_blk + 1024l * rc;
    const unsigned lo_dst = (unsigned)pw_first5_st(L, qq, sp, 0, 0, 0);
    pw_x5_rounds<INV>(b, X, lane, w, tw_pair_stage_u(T1, 1u, 4), 16u,
        [&](int x, int jh) { (void)x; return (unsigned)(2 * sp + 16 * (8 * jh + 2 * w)); },
        [&](int x, int jh, float2 (&v)[2][4][2]) {
#pragma unroll
            for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                for (int n = 0; n < 4; ++n) pw_st((dst + (pw_first5_st(L, 0, 0, x, n, 0) + 16 * (8 * jh + 2 * w + jp))) + lo_dst, v[jp][n][0], v[jp][n][1]);
        });
}

// ---- overlap-save, 65536-point blocks: middle pass = forward pass 1, x conj H, inverse pass 0 on the same tile -----------------------------
// After the forward stages lane (cp, q) holds rows s + 16 j, s = q + 8 x, of its two columns: in the inverse transform's gather pass
// that is group rev2(s) with its rows in rev2 order (PwGroupsRev), so the inverse starts from registers.
__device__ __forceinline__ void pw_ovsave64k_mid_tile(const float2 *__restrict__ a_blk, float2 *__restrict__ b_blk, const float2 *__restrict__ Tf,
                                                      const float2 *__restrict__ T1i, const float2 *__restrict__ Hc, int c, int lane, float4 *Lw)
{ // T1i: the INVERSE plan's gather-pass ordered copy (sub-lengths 1 .. 256), as pw_first_tile reads it
    const int cp = lane & 7, q = lane >> 3;
    float2 a[2][2][16], b[2][2][16];
    pw_mid_stages<false>(b, a_blk + 16 * c, 256l, (unsigned)(16 * c), Tf, Lw, lane);
    const float2 *hc = Hc + 16 * c;
    const unsigned lo = (unsigned)pw_mid_st(256l, q, cp, 0, 0);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int jh = 0; jh < 16; jh += 8) { // eight pairs of spectrum taps as one batch of loads (sixteen spill: 128 sample registers are live)
            float4 h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = pw_ld((hc + pw_mid_st(256l, 0, 0, x, jh + j)) + lo);
            RD_SCHED_BARRIER();
#pragma unroll
            for (int j = 0; j < 8; ++j) { // row s + 16 j = digit reversal of 16 rev2(s) + rev2(j)
                a[x][0][pw_rev2(jh + j)] = cmul_rn(b[x][0][jh + j], make_float2(h[j].x, h[j].y));
                a[x][1][pw_rev2(jh + j)] = cmul_rn(b[x][1][jh + j], make_float2(h[j].z, h[j].w));
            }
        }
    {
        FftTw15 T0; // the inverse's sub-lengths 1 and 4: wave-uniform
        big_tw15(T0, tw_ordered_stage(T1i, 1u, 0), tw_ordered_stage(T1i, 1u, 1), 0u, 1u, 0u, 1u);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int e = 0; e < 2; ++e) macro16_apply<true>(a[x][e], T0);
    }
    pw_exchange_tr<PwGroupsRev>(a, b, Lw, lane);
    const int sp = lane & 7, qq = lane >> 3;
    {
        FftTw15 T0, Tb; // sub-lengths 16 and 64, index 2 sp + e (+ 16 u): neighbouring entries of the ordered copy
        big_tw15x2(T0, Tb, tw_pair_stage_u(T1i, 1u, 2), tw_pair_stage_u(T1i, 1u, 3), 0u, 1u, (unsigned)(2 * sp), 16u);
        RD_SCHED_BARRIER();
#pragma unroll
        for (int x = 0; x < 2; ++x) { macro16_apply<true>(b[x][0], T0); macro16_apply<true>(b[x][1], Tb); }
    }
    float2 *dst = b_blk + 256 * pw_rev2(c);
    const unsigned lo_dst = (unsigned)pw_first_st(8, qq, sp, 0, 0);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int j = 0; j < 16; ++j) pw_st((dst + pw_first_st(8, 0, 0, x, j)) + lo_dst, b[x][0][j], b[x][1][j]);
}

// last pass: inverse pass 1, 1/N, only the hop valid outputs of the block stored
__device__ __forceinline__ void pw_ovsave64k_last_tile(const float2 *__restrict__ b_blk, float2 *__restrict__ out_blk, const float2 *__restrict__ Ti,
                                                       long hop, float scale, int c, int lane, float4 *Lw)
{
    const int cp = lane & 7, q = lane >> 3;
    float2 b[2][2][16];
    pw_mid_stages<true>(b, b_blk + 16 * c, 256l, (unsigned)(16 * c), Ti, Lw, lane);
    float2 *dst = out_blk + 16 * c;
    const unsigned lo = (unsigned)pw_mid_st(256l, q, cp, 0, 0);
    const long lim = hop - 16 * c - (long)lo; // pos < hop
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const long r = pw_mid_st(256l, 0, 0, x, j);
            const float2 v0 = make_float2(mul_rn(b[x][0][j].x, scale), mul_rn(b[x][0][j].y, scale));
            const float2 v1 = make_float2(mul_rn(b[x][1][j].x, scale), mul_rn(b[x][1][j].y, scale));
            if (r + 1 < lim) pw_st_once((dst + r) + lo, v0, v1);
            else if (r < lim) big_st_once((dst + r) + lo, v0);
        }
}


// ---- G128: gather pass of N = 2 * 4^L' points with four stages (fft_big_core.h) -----------------------------------------------------------
// the part after phase A's registers are in place: in-lane stages, regrouping, stage on d3; b[r][x][e][d3] on return
template <bool INV>
__device__ __forceinline__ void pw_g128_stages(float2 (&a)[2][4][8], float2 (&b)[2][4][2][4], const float2 *__restrict__ Tg, float4 *Lw, int lane)
{
    pw_g_inlane<INV>(a[0], Tg);
    pw_g_inlane<INV>(a[1], Tg);
    pw_g_write<0>(a, Lw, lane); wave_lds_fence(); pw_g_read<0>(b, Lw, lane); wave_lds_fence();
    pw_g_write<1>(a, Lw, lane); wave_lds_fence(); pw_g_read<1>(b, Lw, lane); wave_lds_fence();
    pw_g_last<INV>(b, Tg, lane & 7);
}
// in_blk + 32 ctile: source columns 32 ctile .. + 31 (of S = N / 128); out_blk: the working order
template <bool INV, bool MULH = true>
__device__ __forceinline__ void pw_g128_tile(const float2 *in_blk, float2 *out_blk, int lgN, unsigned ctile, int lane, float4 *Lw,
                                             const float2 *__restrict__ mulH, const float2 *__restrict__ Tg)
{
    const long S = 1l << (lgN - 7);
    const int nd = (lgN - 7) / 2; // base-4 digits of a source column
    const int cp = lane & 15, q = lane >> 4;
    const float2 *src = in_blk + 32 * ctile;
    float2 a[2][4][8], b[2][4][2][4];
    const unsigned lo_src = (unsigned)pw_g_ld(S, q, cp, 0, 0);
#pragma unroll
    for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
            const float4 v = pw_ld_once((src + pw_g_ld(S, 0, 0, d2, jb)) + lo_src);
            a[0][d2][jb] = make_float2(v.x, v.y); a[1][d2][jb] = make_float2(v.z, v.w);
        }
    if (MULH && mulH) { // overlap-save: the spectrum product on the way in (wave-uniform branch)
        const float2 *hsrc = mulH + 32 * ctile;
#pragma unroll
        for (int d2 = 0; d2 < 4; ++d2) {
            float4 hv[8];
#pragma unroll
            for (int jb = 0; jb < 8; ++jb) hv[jb] = pw_ld((hsrc + pw_g_ld(S, 0, 0, d2, jb)) + lo_src);
            RD_SCHED_BARRIER();
#pragma unroll
            for (int jb = 0; jb < 8; ++jb) {
                a[0][d2][jb] = cmul_rn(a[0][d2][jb], make_float2(hv[jb].x, hv[jb].y));
                a[1][d2][jb] = cmul_rn(a[1][d2][jb], make_float2(hv[jb].z, hv[jb].w));
            }
        }
    }
    RD_SCHED_BARRIER();
    pw_g128_stages<INV>(a, b, Tg, Lw, lane);
    unsigned hc = 0; // digits of ctile >> 1 reversed (nd - 3 of them), behind the digit whose high bit is ctile & 1
    for (int d = 0, cc = (int)(ctile >> 1); d < nd - 3; ++d, cc >>= 2) hc = (hc << 2) | (cc & 3);
    float2 *dst = out_blk + 128l * (((long)(2 * (ctile & 1))) * (1l << (2 * (nd - 3))) + hc);
    const int kp = lane & 7, cg = lane >> 3;
    const unsigned lo_dst = (unsigned)pw_g_st(nd, cg, kp, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int d3 = 0; d3 < 4; ++d3) pw_st((dst + pw_g_st(nd, 0, 0, x, r, d3)) + lo_dst, b[r][x][0][d3], b[r][x][1][d3]);
}

// ---- G512: G128 on the four wavefronts of a workgroup (source columns N / 512 apart), then the radix-4 stage of sub-length 128 across them --
// (fft_big_core.h; run lane by lane on the CPU in tests/emu: test_pair_g512_gather_pass).  Lg: the four wave-private G128 images; once every
// wavefront is through phase B the fifth stage's two alternating images lie in the same 64 KiB.
template <bool INV>
__device__ __forceinline__ void pw_g512_tile(const float2 *in_blk, float2 *out_blk, int lgN, unsigned ctile, int lane, int w, float4 *Lg,
                                             const float2 *__restrict__ Tg5)
{
    static_assert(4 * PW_G_UNITS >= 2 * PW_X5_UNITS, "the fifth stage's images fit where the G128 images were");
    const long S5 = 1l << (lgN - 9), S = 4 * S5;
    const int nd = (lgN - 9) / 2; // base-4 digits of a source column (N / 512 of them)
    const int cp = lane & 15, q = lane >> 4;
    const float2 *src = in_blk + 32 * ctile + S5 * w;
    float2 a[2][4][8], b[2][4][2][4];
    const unsigned lo_src = (unsigned)pw_g_ld(S, q, cp, 0, 0);
#pragma unroll
    for (int d2 = 0; d2 < 4; ++d2)
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
            const float4 v = pw_ld_once((src + pw_g_ld(S, 0, 0, d2, jb)) + lo_src);
            a[0][d2][jb] = make_float2(v.x, v.y); a[1][d2][jb] = make_float2(v.z, v.w);
        }
    RD_SCHED_BARRIER();
    pw_g128_stages<INV>(a, b, Tg5, Lg + w * PW_G_UNITS, lane);
    __syncthreads(); // every wavefront is done with its private image
    const int kp = lane & 7, cg = lane >> 3;
    unsigned hc = 0; // digits of ctile >> 1 reversed (nd - 3 of them)
    for (int d = 0, cc = (int)(ctile >> 1); d < nd - 3; ++d, cc >>= 2) hc = (hc << 2) | (cc & 3);
    // column 32 ctile + cg + 8 x has the base-4 digits cg & 3 | (cg >> 2) + 2 (x & 1) | (x >> 1) + 2 (ctile & 1) | ctile >> 1: reversed, times 512
    float2 *dst0 = out_blk + 512l * (((long)(cg & 3) << (2 * (nd - 1))) + ((long)(cg >> 2) << (2 * (nd - 2))) + ((long)(2 * (ctile & 1)) << (2 * (nd - 3))) + hc) + 2 * kp;
    const TwPairOrderedT<false> t5{Tg5 + PW_G_TABLE, 128u};
#define REDIO_G5_ROUND(RR, XH)                                                                                                          \
    {                                                                                                                                   \
        float4 *Xi = Lg + ((((2 * RR + XH) & 1) != 0) ? PW_X5_UNITS : 0);                                                               \
        pw_g5_write<RR, XH>(b, Xi, lane, w);                                                                                            \
        __syncthreads();                                                                                                                \
        float2 v[2][4][2];                                                                                                              \
        pw_x5_read(v, Xi, lane, w);                                                                                                     \
        pw_x5_stage<INV>(v, t5, pw_g5_k0(RR, kp, w), 32u);                                                                              \
        const int x = pw_g5_x(XH, w);                                                                                                   \
        float2 *dst = dst0 + 512l * (((long)(2 * (x & 1)) << (2 * (nd - 2))) + ((long)(x >> 1) << (2 * (nd - 3)))) + 16 * RR;           \
        _Pragma("unroll") for (int jp = 0; jp < 2; ++jp)                                                                                \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) pw_st(dst + 32 * pw_g5_d3(w, jp) + 128 * u, v[jp][u][0], v[jp][u][1]);        \
    }
    REDIO_G5_ROUND(0, 0) REDIO_G5_ROUND(0, 1) REDIO_G5_ROUND(1, 0) REDIO_G5_ROUND(1, 1)
#undef REDIO_G5_ROUND
}

// ---- overlap-save, 32768-point blocks: the 65536-point scheme with this gather pass ------------------------------------------------------
// middle pass: the forward in-place pass (rows 128 apart), x conj H, and the INVERSE transform's G128 pass on the same tile.  After
// the forward stages lane (cp, q) holds rows s + 16 j, s = q + 8 x, of tile columns 2 cp + e, i.e. spectrum positions
// n = (16 c + 2 cp + e) + 128 (s + 16 j).  For the inverse's gather pass n = column' + 256 rho: column' = 16 c + 2 cp + e + 128 (q & 1),
// rho = (q >> 1) + 4 (x + 2 j) -- its d3 = q >> 1 and all of (d2, d1, b0) = the base-4 / binary digits of x + 2 j: exactly the rows
// one lane of pw_g128_tile holds for G128 column gamma = 2 (lane & 15) + e, with d3 = lane >> 4.  The inverse starts from registers.
__device__ __forceinline__ void pw_ovsave32k_mid_tile(const float2 *__restrict__ a_blk, float2 *__restrict__ b_blk, const float2 *__restrict__ Tf,
                                                      const float2 *__restrict__ Tgi, const float2 *__restrict__ Hc, int c, int lane, float4 *Lw)
{
    const int cp = lane & 7, q = lane >> 3;
    float2 f[2][2][16], a[2][4][8], b[2][4][2][4];
    pw_mid_stages<false>(f, a_blk + 16 * c, 128l, (unsigned)(16 * c), Tf, Lw, lane);
    const float2 *hc = Hc + 16 * c;
    const unsigned lo = (unsigned)pw_mid_st(128l, q, cp, 0, 0);
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int jh = 0; jh < 16; jh += 8) {
            float4 h[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) h[j] = pw_ld((hc + pw_mid_st(128l, 0, 0, x, jh + j)) + lo);
            RD_SCHED_BARRIER();
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) { // x + 2 j = d2 + 4 d1 + 16 b0: d2 = x + 2 (j & 1), d1 = (j >> 1) & 3, b0 = j >> 3
                const int j = jh + jj, d2 = x + 2 * (j & 1), jb = (j >> 3) + 2 * ((j >> 1) & 3);
                a[0][d2][jb] = cmul_rn(f[x][0][j], make_float2(h[jj].x, h[jj].y));
                a[1][d2][jb] = cmul_rn(f[x][1][j], make_float2(h[jj].z, h[jj].w));
            }
        }
    pw_g128_stages<true>(a, b, Tgi, Lw, lane);
    // G128 column gamma = cg + 8 x' is spectrum column' = 16 c + (gamma & 15) + 128 (gamma >> 4), four base-4 digits: gamma & 3, (gamma >> 2) & 3,
    // c & 3, (c >> 2) + 2 (gamma >> 4) -> position 128 h + 32 d3 + kk with h the digits reversed
    const int kp = lane & 7, cg = lane >> 3;
    float2 *dst = b_blk + 128l * (4 * (c & 3) + (c >> 2));
    const unsigned lo_dst = (unsigned)(128 * (64 * (cg & 3) + 16 * (cg >> 2)) + 2 * kp);
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int d3 = 0; d3 < 4; ++d3)
                pw_st((dst + (128 * (32 * (x & 1) + 2 * (x >> 1)) + 32 * d3 + 16 * r)) + lo_dst, b[r][x][0][d3], b[r][x][1][d3]);
}

} // namespace redio
