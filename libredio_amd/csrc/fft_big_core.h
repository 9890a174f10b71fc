// fft_big_core.h -- the pieces of the multi-pass transforms (65536 points and the larger powers of two, kissfft::fft,
// src/kissfft/src/kissfft.rs:18-31) that do not depend on the device: twiddle access in table / ordered / interleaved form, two
// radix-4 stages on 16 points, and the lane <-> row / column maps and LDS images of the two-column ("pair") tile program.
//
// Host-compilable: tests/emu runs the same lane programs on the CPU, one lane at a time (tests/test_emu_lane_programs.py).
#pragma once
#include "fft_core.h"

namespace redio {

// ---- two radix-4 stages on 16 points in registers -----------------------------------------------------------------------
// Stage A multiplies by T.t[0..2] (the same for its four butterflies), stage B butterfly u by T.t[3 + 3u ..].
struct FftTw15 { float2 t[15]; };
template <bool INV>
RD_HD void macro16_apply(float2 (&a)[16], const FftTw15 &T)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int q = 0; q < 4; q += 2)
        bfly4x2<INV>(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3], T.t[0], T.t[1], T.t[2], a[4 * q + 4], a[4 * q + 5], a[4 * q + 6], a[4 * q + 7],
                     T.t[0], T.t[1], T.t[2]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int u = 0; u < 4; u += 2)
        bfly4x2<INV>(a[u], a[u + 4], a[u + 8], a[u + 12], T.t[3 + 3 * u], T.t[4 + 3 * u], T.t[5 + 3 * u], a[u + 1], a[u + 5], a[u + 9], a[u + 13],
                     T.t[6 + 3 * u], T.t[7 + 3 * u], T.t[8 + 3 * u]);
}

// twiddle n k N / (4 m) of the stage with sub-length m: straight from kissfft's table (stride fs = N / (4 m)), or from the
// pass-ordered copy T[(n - 1) m + k] (fftbig_tables_build), where lanes with neighbouring k read neighbouring entries --
// in the table order a wave's 64 twiddles of an in-place pass sit in 64 different cache lines
struct TwGather {
    const float2 *tw; unsigned fs;
    RD_HD float2 get(unsigned n, unsigned k) const { return tw[n * k * fs]; }
    RD_HD void get3(unsigned k, float2 &t1, float2 &t2, float2 &t3) const { t1 = get(1, k); t2 = get(2, k); t3 = get(3, k); }
};
struct TwOrdered {
    const float2 *T; unsigned m;
    RD_HD float2 get(unsigned n, unsigned k) const { return T[(n - 1) * m + k]; }
    RD_HD void get3(unsigned k, float2 &t1, float2 &t2, float2 &t3) const { t1 = get(1, k); t2 = get(2, k); t3 = get(3, k); }
};
// the copy of a four-stage in-place pass, INTERLEAVED: entry k of a stage holds its three twiddles side by side, T[4 k + (n - 1)]
// (the fourth slot pads the entry to 32 bytes).  A butterfly's three twiddles are then one 16-byte and one 8-byte load from one
// 32-byte entry instead of three 8-byte loads from three planes m entries apart -- measured on the access pattern alone: the
// in-place pass of 65536 points 283 -> 233 us per 2^26 points, of 2^24 points 318 -> 254 (profiles/r02_fft_pass_times.txt).
// Round 3: a tile reads 43 KB of these entries for its 32 KB of samples, all from L2 -- and a timing-only build in which every lane reads
// entry k mod 16 runs the in-place pass of 65536 points in 253 us against 246, the overlap-save passes in 32.1 / 46.0 us against 33.6 / 47.2:
// the twiddle traffic is not what holds the passes at 4.2-5.5 TB/s, so a workgroup-resident LDS copy of them was not built.
struct TwInter {
    const float2 *T;
    RD_HD float2 get(unsigned n, unsigned k) const { return T[4 * k + (n - 1)]; }
    RD_HD void get3(unsigned k, float2 &t1, float2 &t2, float2 &t3) const
    {
        const float4 q = *reinterpret_cast<const float4 *>(T + 4 * (size_t)k); // 32-byte entries of a 256-byte aligned table
        t1 = make_float2(q.x, q.y); t2 = make_float2(q.z, q.w); t3 = T[4 * (size_t)k + 2];
    }
};
// the ordered copy read two neighbouring k at a time (the pair tile program: a lane's two columns are twiddle indices k, k + 1, k even):
// T[(n - 1) m + k] and T[(n - 1) m + k + 1] are one aligned 16-byte load, and the eight lanes of a row group read 128 contiguous
// bytes per load -- one cache line per row group and load instead of the four half-used lines of the 32-byte interleaved entries
#if defined(__HIP_DEVICE_COMPILE__)
typedef float rd_v4u __attribute__((ext_vector_type(4), aligned(8)));
RD_HD float4 rd_ld_pair(const float2 *p) { const rd_v4u v = *reinterpret_cast<const rd_v4u *>(p); return make_float4(v.x, v.y, v.z, v.w); }
#else
RD_HD float4 rd_ld_pair(const float2 *p) { return make_float4(p[0].x, p[0].y, p[1].x, p[1].y); }
#endif
// ALIGNED = false: the gather pass's copy (sub-lengths 1, 4, 16, 64, 256 back to back: a stage starts on an odd entry) -- the same
// 16-byte load from an 8-byte aligned address
template <bool ALIGNED = true>
struct TwPairOrderedT {
    const float2 *T; unsigned m;
    RD_HD void get3x2(unsigned k, float2 &a1, float2 &a2, float2 &a3, float2 &b1, float2 &b2, float2 &b3) const
    {
        const float4 q1 = ALIGNED ? *reinterpret_cast<const float4 *>(T + k) : rd_ld_pair(T + k),
                     q2 = ALIGNED ? *reinterpret_cast<const float4 *>(T + m + k) : rd_ld_pair(T + m + k),
                     q3 = ALIGNED ? *reinterpret_cast<const float4 *>(T + 2 * m + k) : rd_ld_pair(T + 2 * m + k);
        a1 = make_float2(q1.x, q1.y); b1 = make_float2(q1.z, q1.w);
        a2 = make_float2(q2.x, q2.y); b2 = make_float2(q2.z, q2.w);
        a3 = make_float2(q3.x, q3.y); b3 = make_float2(q3.z, q3.w);
    }
};
typedef TwPairOrderedT<true> TwPairOrdered;
template <bool INV, typename TA, typename TB>
RD_HD void big_macro16(float2 (&a)[16], TA ta, TB tb, unsigned l, unsigned m_lo, unsigned kk, unsigned m)
{
    {
        const unsigned k = l + m_lo * kk;
        float2 t1, t2, t3;
        ta.get3(k, t1, t2, t3);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int q = 0; q < 4; q += 2)
            bfly4x2<INV>(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3], t1, t2, t3, a[4 * q + 4], a[4 * q + 5], a[4 * q + 6], a[4 * q + 7], t1, t2, t3);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int u = 0; u < 4; u += 2) {
        const unsigned k = l + m_lo * (kk + u * m), kb = k + m_lo * m;
        float2 p1, p2, p3, r1, r2, r3;
        tb.get3(k, p1, p2, p3);
        tb.get3(kb, r1, r2, r3);
        bfly4x2<INV>(a[u], a[u + 4], a[u + 8], a[u + 12], p1, p2, p3, a[u + 1], a[u + 5], a[u + 9], a[u + 13], r1, r2, r3);
    }
}
// the fifteen twiddles big_macro16 would read, as one batch (a batch serves every 16-point group with the same (l, kk))
template <typename TA, typename TB>
RD_HD void big_tw15(FftTw15 &T, TA ta, TB tb, unsigned l, unsigned m_lo, unsigned kk, unsigned m)
{
    ta.get3(l + m_lo * kk, T.t[0], T.t[1], T.t[2]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (unsigned u = 0; u < 4; ++u) tb.get3(l + m_lo * (kk + u * m), T.t[3 + 3 * u], T.t[4 + 3 * u], T.t[5 + 3 * u]);
}
// ... for the two columns l, l + 1 (l even) of a lane of the pair program, from the ordered copy
template <typename TP>
RD_HD void big_tw15x2(FftTw15 &Ta, FftTw15 &Tb, TP ta, TP tb, unsigned l, unsigned m_lo, unsigned kk, unsigned m)
{
    ta.get3x2(l + m_lo * kk, Ta.t[0], Ta.t[1], Ta.t[2], Tb.t[0], Tb.t[1], Tb.t[2]);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (unsigned u = 0; u < 4; ++u)
        tb.get3x2(l + m_lo * (kk + u * m), Ta.t[3 + 3 * u], Ta.t[4 + 3 * u], Ta.t[5 + 3 * u], Tb.t[3 + 3 * u], Tb.t[4 + 3 * u], Tb.t[5 + 3 * u]);
}
// the ordered copy of one pass: stage t (sub-length m_lo 4^t) starts at m_lo (4^t - 1) and holds 3 m_lo 4^t entries
RD_HD TwOrdered tw_ordered_stage(const float2 *T, unsigned m_lo, int t) { return TwOrdered{T + m_lo * ((1u << (2 * t)) - 1), m_lo << (2 * t)}; }
RD_HD TwPairOrdered tw_pair_stage(const float2 *T, unsigned m_lo, int t) { return TwPairOrdered{T + m_lo * ((1u << (2 * t)) - 1), m_lo << (2 * t)}; }
RD_HD TwPairOrderedT<false> tw_pair_stage_u(const float2 *T, unsigned m_lo, int t) { return TwPairOrderedT<false>{T + m_lo * ((1u << (2 * t)) - 1), m_lo << (2 * t)}; }
// the interleaved copy: stage t (sub-length m_lo 4^t) starts at entry m_lo (4^t - 1) / 3 and holds m_lo 4^t entries of four float2
RD_HD TwInter tw_inter_stage(const float2 *T, unsigned m_lo, int t) { return TwInter{T + 4 * (size_t)(m_lo * (((1u << (2 * t)) - 1) / 3))}; }

// =============================================================================================================================
// The two-column ("pair") tile program of the four-stage passes.
//
// A tile is 256 rows x 16 columns of cf32 and belongs to ONE wavefront, as before (fft_kernels.hip, "one wavefront per 256 x 16
// tile"), but a lane now owns TWO ADJACENT columns (or, on the transposed side of the gather pass, two adjacent rows): every global
// access is one 16-byte access per lane (global_load / global_store_dwordx4: eight rows of 128 bytes per wave instruction instead of
// four rows of 128 bytes in 8-byte pieces) and every LDS access of the regroupings is a ds_write_b128 / ds_read_b128.  The
// butterflies, their order and their twiddles are those of the one-column program: same bits.
//
//   lane = cp + 8 q      cp = 0..7: columns 2 cp, 2 cp + 1      q = 0..7
//   phase A (stages t = 0, 1 of the pass):  a[i][e][j] = row 16 G(q, i) + j of column 2 cp + e     i = 0, 1 (two groups of 16 rows)
//   phase B, plain:                         b[x][e][j] = row (q + 8 x) + 16 j of column 2 cp + e   x = 0, 1
//   phase B, transposed (lane = sp + 8 qq): b[x][e][j] = row (2 sp + e) + 16 j of column 2 qq + x  (a lane owns two adjacent ROWS:
//                                           the gather pass writes the working order, where the rows of a column are contiguous)
// G(q, i) is the group map of the side that writes: 8 i + q when the tile was loaded that way (PwGroupsLinear), and the digit reversal
// of q + 8 i when phase A starts from the registers of a plain phase B (PwGroupsRev: the overlap-save middle pass, whose forward
// output row s + 16 j is source row 16 rev2(j') + rev2(g) of the inverse transform's gather pass -- group g = rev2(s)).
//
// The regroupings go through a wave-private LDS image of 8 KiB (512 units of 16 bytes) in four rounds; the images are swizzled so
// that both the ds_write_b128 and the ds_read_b128 of every round are bank-conflict free (a b128 access is served in four groups of
// 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- each of which must cover sixteen different 16-byte bank
// positions: MI355X_MICROARCH.md, LDS; checked lane group by lane group in tests/emu).
// =============================================================================================================================
struct PwGroupsLinear { static constexpr int of(int q, int i) { return 8 * i + q; } };
struct PwGroupsRev { static constexpr int of(int q, int i) { return 4 * ((q + 8 * i) & 3) + ((q + 8 * i) >> 2); } };
RD_HD constexpr int pw_rev2(int v) { return ((v & 3) << 2) | (v >> 2); } // two base-4 digits swapped

// plain regrouping, round (i, jh): the image holds rows 16 G(hi, i) + 8 jh + lo (hi = the writing lane's q, lo = 0..7) of all 16
// columns; unit (hi, lo, cp) = the two columns of lane-column cp in that row
RD_HD constexpr int pw_unit_plain(int hi, int lo, int cp) { return 16 * (4 * hi + (lo >> 1)) + 8 * ((hi ^ lo) & 1) + cp; }
// transposed regrouping, round (i, x): the image holds, for the eight columns 2 c + x (c = 0..7), the groups G(g, i) (g = the
// writing lane's q); unit (c, g, jp) = rows 2 jp, 2 jp + 1 of that group in that column
RD_HD constexpr int pw_unit_tr(int c, int g, int jp) { return 16 * (4 * c + (g >> 1)) + 8 * ((g ^ c) & 1) + (c ^ jp); }
constexpr int PW_UNITS = 512; // 16-byte units of one wave's image

// One round of the plain regrouping.  Lw: this wave's image (float4 units).  The writing half and the reading half are separate
// functions because every lane of the wave must have written before any lane reads (device: wave_lds_fence between them).
template <int I, int JH>
RD_HD void pw_plain_write(const float2 (&a)[2][2][16], float4 *Lw, int lane)
{
    const int cp = lane & 7, q = lane >> 3;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int jj = 0; jj < 8; ++jj) {
        const int j = 8 * JH + jj;
        Lw[pw_unit_plain(q, jj, cp)] = make_float4(a[I][0][j].x, a[I][0][j].y, a[I][1][j].x, a[I][1][j].y);
    }
}
template <typename G, int I, int JH>
RD_HD void pw_plain_read(float2 (&b)[2][2][16], const float4 *Lw, int lane)
{
    const int cp = lane & 7, q = lane >> 3;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int qw = 0; qw < 8; ++qw) { // the group written by the lanes with q = qw: row q + 8 JH of it
        const float4 v = Lw[pw_unit_plain(qw, q, cp)];
        b[JH][0][G::of(qw, I)] = make_float2(v.x, v.y);
        b[JH][1][G::of(qw, I)] = make_float2(v.z, v.w);
    }
}
// One round of the transposed regrouping: column parity X of the writers' pairs.
template <int I, int X>
RD_HD void pw_tr_write(const float2 (&a)[2][2][16], float4 *Lw, int lane)
{
    const int cp = lane & 7, q = lane >> 3;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int jp = 0; jp < 8; ++jp)
        Lw[pw_unit_tr(cp, q, jp)] = make_float4(a[I][X][2 * jp].x, a[I][X][2 * jp].y, a[I][X][2 * jp + 1].x, a[I][X][2 * jp + 1].y);
}
template <typename G, int I, int X>
RD_HD void pw_tr_read(float2 (&b)[2][2][16], const float4 *Lw, int lane)
{
    const int sp = lane & 7, qq = lane >> 3;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int qw = 0; qw < 8; ++qw) { // group G(qw, I), rows 2 sp and 2 sp + 1 of it, column 2 qq + X
        const float4 v = Lw[pw_unit_tr(qq, qw, sp)];
        b[X][0][G::of(qw, I)] = make_float2(v.x, v.y);
        b[X][1][G::of(qw, I)] = make_float2(v.z, v.w);
    }
}

// ---- where the registers of the two phases live in memory (cf32 offsets from the tile's origin; each is the sum of a lane part,
// f(q, cp, 0, 0), and a wave-uniform part, f(0, 0, i, j): the device code adds the two so that the uniform part stays scalar) -----
// in-place pass on rows m_lo apart: phase A loads row 16 (8 i + q) + j, plain phase B stores row (q + 8 x) + 16 j; columns 2 cp, 2 cp + 1
RD_HD constexpr long pw_mid_ld(long m_lo, int q, int cp, int i, int j) { return m_lo * (16 * (8 * i + q) + j) + 2 * cp; }
RD_HD constexpr long pw_mid_st(long m_lo, int q, int cp, int x, int j) { return m_lo * ((q + 8 * x) + 16 * j) + 2 * cp; }
// gather pass of a 4^L-point transform (S = 4^L / 256: source row stride): working row 16 g + j is source row 16 rev2(j) + rev2(g),
// g = 8 i + q: rev2(g) = 4 (q & 3) + (q >> 2) + 2 i
RD_HD constexpr long pw_first_ld(long S, int q, int cp, int i, int j) { return S * (16 * pw_rev2(j) + 4 * (q & 3) + (q >> 2) + 2 * i) + 2 * cp; }
// ... and its store: source column r = 16 c + gamma (gamma = 2 qq + x) is column h = digit reversal of r over L - 4 digits of the
// working array = (gamma & 3) 4^(L-5) + (gamma >> 2) 4^(L-6) + rev(c); rows 2 sp, 2 sp + 1 (+ 16 j) of it are contiguous.
// rc = rev(c) is added by the caller (wave-uniform).
RD_HD constexpr long pw_first_st(int L, int qq, int sp, int x, int j)
{
    return 256l * ((long)((2 * (qq & 1) + x)) * (1l << (2 * (L - 5))) + (long)(qq >> 1) * (1l << (2 * (L - 6)))) + 16 * j + 2 * sp;
}


// =============================================================================================================================
// G128: the gather pass of N = 2 * 4^L' points (32768, 131072 ...) with FOUR stages -- kissfft's radix-2 stage (sub-length 1) and the
// radix-4 stages of sub-length 2, 8 and 32 -- on tiles of 128 rows x 32 source columns, one wavefront per tile, in the pair layout.
// (fftbig_first2_kernel does the first three of them, 32 rows; with the fourth in the same pass 2^15 points are this pass plus ONE
// four-stage in-place pass, 2^17 points this pass plus one five-stage pass, and the overlap-save of 32768-point blocks gets the
// three-pass scheme of the 65536-point blocks: the forward in-place pass, the spectrum product and this pass of the INVERSE transform
// on one tile.)
//
// Leaf position P = 128 h + (b0 + 2 d1 + 8 d2 + 32 d3) holds input n = column + S (d3 + 4 d2 + 16 d1 + 64 b0), S = N / 128, h = digit
// reversal of the column.  lane = cp + 16 q: columns 2 cp, 2 cp + 1 (cp = 0..15) of source rows with d3 = q.
//   phase A  a[e][d2][b0 + 2 d1]: the stages on b0, d1, d2 run inside the lane (the register program of fftbig_first2_kernel)
//   phase B  lane = kp + 8 cg: b[r][x][e][d3] = position kk = 16 r + 2 kp + e (+ 32 d3) of column cg + 8 x   (r = 0, 1; x = 0..3):
//            the stage on d3, then every store instruction writes eight 128-byte runs (rows kk, kk + 1 of eight lanes are contiguous)
// The regrouping goes through a 16 KiB wave-private image in two rounds (kk below 16, then the rest), b128 both ways, swizzled
// conflict-free for the documented lane groups of ds_read_b128 and for sixteen consecutive lanes on the write side.
// =============================================================================================================================
constexpr int PW_G_UNITS = 1024;
// unit of (column gamma = 0..31, d3, kp = 0..7): rows 2 kp, 2 kp + 1 (+ 16 r) of that column and d3
RD_HD constexpr int pw_unit_g(int gamma, int d3, int kp) { return 16 * ((gamma & 1) + 2 * kp + 16 * d3) + (((gamma >> 1) ^ (((kp & 3) << 2) | ((gamma & 1) << 1))) & 15); }
template <int R>
RD_HD void pw_g_write(const float2 (&a)[2][4][8], float4 *Lw, int lane)
{
    const int cp = lane & 15, q = lane >> 4;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int e = 0; e < 2; ++e)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int kp = 0; kp < 8; ++kp) { // kk = 16 R + 2 kp (+1) = position (kk & 7) of d2 = kk >> 3
            const float2 v0 = a[e][2 * R + (kp >> 2)][2 * (kp & 3)], v1 = a[e][2 * R + (kp >> 2)][2 * (kp & 3) + 1];
            Lw[pw_unit_g(2 * cp + e, q, kp)] = make_float4(v0.x, v0.y, v1.x, v1.y);
        }
}
template <int R>
RD_HD void pw_g_read(float2 (&b)[2][4][2][4], const float4 *Lw, int lane)
{
    const int kp = lane & 7, cg = lane >> 3;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int x = 0; x < 4; ++x)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int d3 = 0; d3 < 4; ++d3) {
            const float4 v = Lw[pw_unit_g(cg + 8 * x, d3, kp)];
            b[R][x][0][d3] = make_float2(v.x, v.y);
            b[R][x][1][d3] = make_float2(v.z, v.w);
        }
}
// the plan's ordered twiddle copy for this pass, 128 entries: [0] tw[0] (radix-2 stage); [2 + (n-1) 2 + k] sub-length 2 (table stride N / 8);
// [8 + (n-1) 8 + k] sub-length 8 (N / 32); [32 + (n-1) 32 + k] sub-length 32 (N / 128)
constexpr int PW_G_TABLE = 128;
RD_HD void pw_g_table_entry(const float2 *tw, unsigned N, int i, float2 &out)
{
    if (i < 2) { out = tw[0]; return; }
    const int m = i < 8 ? 2 : i < 32 ? 8 : 32, r = i - m, n = r / m + 1, k = r - (n - 1) * m;
    out = tw[(size_t)n * k * (N / (4u * m))];
}
// the three stages a lane runs on its own 32 rows of one column: a[d2][b0 + 2 d1] -> position (b0 + 2 d1) + 8 d2 in a[d2'][k]
template <bool INV>
RD_HD void pw_g_inlane(float2 (&a)[4][8], const float2 *Tg)
{
    const float2 w0 = Tg[0], w1 = Tg[2 + 1], w2 = Tg[2 + 2 + 1], w3 = Tg[2 + 4 + 1]; // sub-length 2, k = b0 = 1: n = 1, 2, 3
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int d2 = 0; d2 < 4; ++d2) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int d1 = 0; d1 < 4; ++d1) bfly2(a[d2][2 * d1], a[d2][2 * d1 + 1], w0);
        bfly4x2<INV>(a[d2][0], a[d2][2], a[d2][4], a[d2][6], w0, w0, w0, a[d2][1], a[d2][3], a[d2][5], a[d2][7], w1, w2, w3);
    }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int k = 0; k < 8; k += 2) // stage on d2: sub-length 8, k = b0 + 2 d1
        bfly4x2<INV>(a[0][k], a[1][k], a[2][k], a[3][k], Tg[8 + k], Tg[16 + k], Tg[24 + k],
                     a[0][k + 1], a[1][k + 1], a[2][k + 1], a[3][k + 1], Tg[8 + k + 1], Tg[16 + k + 1], Tg[24 + k + 1]);
}
// the stage on d3 (sub-length 32, twiddle index kk = 16 r + 2 kp + e) on the lane's sixteen four-point groups
template <bool INV>
RD_HD void pw_g_last(float2 (&b)[2][4][2][4], const float2 *Tg, int kp)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 2; ++r) {
        const float4 q1 = *reinterpret_cast<const float4 *>(Tg + 32 + 16 * r + 2 * kp), q2 = *reinterpret_cast<const float4 *>(Tg + 64 + 16 * r + 2 * kp),
                     q3 = *reinterpret_cast<const float4 *>(Tg + 96 + 16 * r + 2 * kp);
        const float2 a1 = make_float2(q1.x, q1.y), b1 = make_float2(q1.z, q1.w), a2 = make_float2(q2.x, q2.y), b2 = make_float2(q2.z, q2.w),
                     a3 = make_float2(q3.x, q3.y), b3 = make_float2(q3.z, q3.w);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int x = 0; x < 4; ++x)
            bfly4x2<INV>(b[r][x][0][0], b[r][x][0][1], b[r][x][0][2], b[r][x][0][3], a1, a2, a3, b[r][x][1][0], b[r][x][1][1], b[r][x][1][2], b[r][x][1][3], b1, b2, b3);
    }
}
// source of phase A (cf32 offset from in_blk + 32 ctile): row S (q + 4 d2 + 16 d1 + 64 b0), jb = b0 + 2 d1; lane part f(q, cp, 0, 0) + uniform part f(0, 0, d2, jb)
RD_HD constexpr long pw_g_ld(long S, int q, int cp, int d2, int jb) { return S * (q + 4 * d2 + 16 * (jb >> 1) + 64 * (jb & 1)) + 2 * cp; }
// destination of phase B, stand-alone pass: column 32 ctile + (cg + 8 x) of nd base-4 digits -> 128 h + 32 d3 + 16 r + 2 kp; the caller adds
// 128 * (2 (ctile & 1) 4^(nd-3) + rev(ctile >> 1)).  Lane part f(cg, kp, 0, 0, 0) + uniform part f(0, 0, x, r, d3)
RD_HD constexpr long pw_g_st(int nd, int cg, int kp, int x, int r, int d3)
{
    return 128l * ((long)(cg & 3) * (1l << (2 * (nd - 1))) + (long)((cg >> 2) + 2 * (x & 1)) * (1l << (2 * (nd - 2))) + (long)(x >> 1) * (1l << (2 * (nd - 3)))) +
           32 * d3 + 16 * r + 2 * kp;
}


// =============================================================================================================================
// Fifth stage of the five-stage passes in the pair layout.  A tile of 1024 rows x 16 columns belongs to a workgroup of four wavefronts:
// wavefront n runs the four-stage pair program on the quarter of the rows that forms one 256-row sub-transform, then the fifth stage
// combines position r of the four quarters.  The regrouping goes through a workgroup image of 32 KiB (2048 units of 16 bytes) in four
// rounds; round (x, jh) moves rows j = 8 jh .. 8 jh + 7 of phase B's slot x: wavefront n WRITES its eight units (n, q, jj, cp) -- two
// neighbouring columns (rows, on the transposed side of the gather pass) of one row each -- and wavefront w READS, for the rows
// jj = 2 w, 2 w + 1, the four quarters: v[jj'][n][e].  Writer and reader use the same lane -> (cp, q) map, so one swizzle (the q & 1
// bit picks the half of a 256-byte bank line) keeps both ds_write_b128 and ds_read_b128 conflict-free.  Two images alternate, so a round
// needs ONE workgroup barrier (between its writes and its reads); image 0 is the union of the four wave-private 8 KiB images of the
// four-stage program (wavefront n's slice is its own private image: it is written again only after its owner is done with it).
// =============================================================================================================================
constexpr int PW_X5_UNITS = 2048;
RD_HD constexpr int pw_unit_x5(int n, int q, int jj, int cp) { return 16 * (32 * n + 8 * (q >> 1) + jj) + 8 * (q & 1) + cp; }
template <int X, int JH>
RD_HD void pw_x5_write(const float2 (&b)[2][2][16], float4 *Xi, int lane, int n)
{
    const int cp = lane & 7, q = lane >> 3;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int jj = 0; jj < 8; ++jj) {
        const int j = 8 * JH + jj;
        Xi[pw_unit_x5(n, q, jj, cp)] = make_float4(b[X][0][j].x, b[X][0][j].y, b[X][1][j].x, b[X][1][j].y);
    }
}
RD_HD void pw_x5_read(float2 (&v)[2][4][2], const float4 *Xi, int lane, int w)
{
    const int cp = lane & 7, q = lane >> 3;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int jp = 0; jp < 2; ++jp)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int n = 0; n < 4; ++n) {
            const float4 t = Xi[pw_unit_x5(n, q, 2 * w + jp, cp)];
            v[jp][n][0] = make_float2(t.x, t.y);
            v[jp][n][1] = make_float2(t.z, t.w);
        }
}
// the stage itself on the lane's four groups: v[jp][.][e] with the twiddles of index k (e = 0) and k + 1 (e = 1), k = k0 + dk jp
template <bool INV, typename TP>
RD_HD void pw_x5_stage(float2 (&v)[2][4][2], TP t4, unsigned k0, unsigned dk)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int jp = 0; jp < 2; ++jp) {
        float2 a1, a2, a3, b1, b2, b3;
        t4.get3x2(k0 + dk * (unsigned)jp, a1, a2, a3, b1, b2, b3);
        bfly4x2<INV>(v[jp][0][0], v[jp][1][0], v[jp][2][0], v[jp][3][0], a1, a2, a3, v[jp][0][1], v[jp][1][1], v[jp][2][1], v[jp][3][1], b1, b2, b3);
    }
}
// gather pass with five stages (4^L points): source column 16 c + (2 qq + x) of the N / 1024 source columns -> column h (L - 5 digits
// reversed) of the working array, 1024 positions each: 256 n + (2 sp + e) + 16 j.  Lane part f(qq, sp, 0, 0, 0) + uniform part f(0, 0, x, n, j); the
// caller adds 1024 * rev(c)
RD_HD constexpr long pw_first5_st(int L, int qq, int sp, int x, int n, int j)
{
    return 1024l * ((long)(2 * (qq & 1) + x) * (1l << (2 * (L - 6))) + (long)(qq >> 1) * (1l << (2 * (L - 7)))) + 256 * n + 16 * j + 2 * sp;
}


// =============================================================================================================================
// G512 (lane program and CPU emulation this round; the device kernel is next): the gather pass of N = 2 * 4^L' points with FIVE stages --
// G128's four and the radix-4 stage of sub-length 128 -- on tiles of 512 rows x 32 source columns, one workgroup of four wavefronts per
// tile.  Leaf position P = 512 h + (b0 + 2 d1 + 8 d2 + 32 d3 + 128 d4) holds input n = column + S5 (d4 + 4 d3 + 16 d2 + 64 d1 + 256 b0),
// S5 = N / 512, h = digit reversal of the column: wavefront n = d4 runs the G128 program on the columns shifted by S5 n (its row stride
// N / 128 is 4 S5), then the fifth stage combines position p of the four wavefronts with the twiddles n k N / 512, k = p.  With this pass
// 2^19 points are TWO passes (this one and a five-stage in-place pass on rows 512 apart) instead of three.
// The regrouping reuses the five-stage passes' workgroup image and unit map (pw_unit_x5, conflict-free both ways): round (r, xh) moves
// b[r][2 xh + xl][.][d3] as unit jj = 4 xl + d3 of lane (cp = kp, q = cg); wavefront w reads jj = 2 w, 2 w + 1 of the four wavefronts.
// After phase B of G128 the two images can both lie where the four wave-private G128 images were (4 x 16 KiB >= 2 x 32 KiB).
// =============================================================================================================================
constexpr int PW_G5_TABLE = PW_G_TABLE + 3 * 128; // G128's 128 entries, then the ordered copy of sub-length 128: [128 + (n - 1) 128 + k] = tw[n k N / 512]
RD_HD void pw_g5_table_entry(const float2 *tw, unsigned N, int i, float2 &out)
{
    if (i < PW_G_TABLE) { pw_g_table_entry(tw, N, i, out); return; }
    const int r = i - PW_G_TABLE, n = r / 128 + 1, k = r % 128;
    out = tw[(size_t)n * k * (N / 512u)];
}
template <int R, int XH>
RD_HD void pw_g5_write(const float2 (&b)[2][4][2][4], float4 *Xi, int lane, int n)
{
    const int kp = lane & 7, cg = lane >> 3;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int jj = 0; jj < 8; ++jj) {
        const float2 v0 = b[R][2 * XH + (jj >> 2)][0][jj & 3], v1 = b[R][2 * XH + (jj >> 2)][1][jj & 3];
        Xi[pw_unit_x5(n, cg, jj, kp)] = make_float4(v0.x, v0.y, v1.x, v1.y);
    }
}
// what wavefront w holds after pw_x5_read in round (r, xh): v[jp][n][e] = position 16 r + 2 kp + e + 32 d3 of column cg + 8 x, quarter n
RD_HD constexpr int pw_g5_x(int xh, int w) { return 2 * xh + (w >> 1); }
RD_HD constexpr int pw_g5_d3(int w, int jp) { return 2 * (w & 1) + jp; }
RD_HD constexpr unsigned pw_g5_k0(int r, int kp, int w) { return (unsigned)(16 * r + 2 * kp + 32 * pw_g5_d3(w, 0)); } // twiddle index of jp = 0, e = 0; jp adds 32

// ---- input of the four-wave kernels (16384 = 4 x 4096, 8192 = 4 x 2048 points; fft_kernels.hip f16k_deal_load / f8k_deal_load) ---------
// Wave q transforms x[4 n + q].  The four waves read a round (half a block) in 512-byte runs -- wave w, load t: samples 256 t + 64 w + lane
// of the round -- and DEAL them through LDS: sample e = 4 n + q goes to plane q, cell n; wave q then reads its positions lane-contiguous.
// The plane stride PS is 8 mod 16 cells, so the four planes start 16 banks apart: the 16 lanes of a ds_write_b64 group (4 cells in each
// plane) and the 32 lanes of a half-wave (8 in each) write different banks; the reads are consecutive cells.  Run lane by lane on the CPU
// in tests/emu (tests/test_emu_lane_programs.py::test_four_wave_deal).
constexpr int F16K_PS = 2056, F8K_PS = 1032;
RD_HD int deal_write_cell(int PS, int w, int lane, int t) { return PS * (lane & 3) + 16 * w + (lane >> 2) + 64 * t; }
RD_HD int deal_read_cell(int PS, int q, int lane, int pos) { return PS * q + pos + lane; } // pos: position in the round, a multiple of 64
// position (in the wave's sub-sequence) of register a[i][j] of the 4096-point program, and of a[d2][j] of the 2048-point program
RD_HD int f4k_reg_pos(int i, int j) { return 1024 * (j & 3) + 256 * (j >> 2) + 64 * i; }
RD_HD int f2k_reg_pos(int d2, int j) { return 64 * (d2 + 4 * (j >> 1) + 16 * (j & 1)); }
} // namespace redio
