// pfb_core.h -- lane programs of the 64-channel polyphase channelizer (BASELINE.json configs[3]).
//
// A new composition of the two reference primitives: per branch m the strict
// left fold of dsputils::convolve (src/dsputils/src/dsputils.rs:31) over P taps g_m[p] = h[64 p + m]
// on rows x_t[m] = x[64 t + m], then kissfft's 64-point forward transform (4 x 4 x 4, published
// butterfly order) across the branches of each row.  Bit-identical to oracle orc_pfb_channelizer.
//
// One wavefront works on 16 consecutive rows: lane = branch for the FIRs (coalesced row loads, the
// P-row window lives in registers), then two padded LDS exchanges turn the 16 x 64 tile so that a
// lane owns 16 points of one row's transform: stages over d2, d1 in registers | exchange | stage d0.
// Host-compilable (tests/emu).
#pragma once
#include "fft_core.h"

namespace redio {

constexpr int PFB_M = 64;
constexpr int PFB_TILE = 16;  // rows per wave tile
constexpr int PFB_ROW = 68;   // LDS row stride (float2): 64 + 4 keeps the first exchange conflict-free
constexpr int PFB_LDS = PFB_TILE * PFB_ROW;

// exchange 1: FIR lane (= branch m) stores row ti; transform lane (tt = lane>>2, d0 = lane&3) loads
// element e = d1 + 4 d2 of row tt, i.e. branch m = d0 + 4 e
RD_HD int pfb_x1_store(int ti, int lane) { return ti * PFB_ROW + lane; }
RD_HD int pfb_x1_load(int lane, int e) { return (lane >> 2) * PFB_ROW + (lane & 3) + 4 * e; }

// stages over d2 (m = 1) and d1 (m = 4, fstride 4, k = k2): v[d1 + 4 d2] -> v[k1 + 4 k2]
template <bool INV, typename TwPtr>
RD_HD void pfb_fft64_passAB(float2 (&v)[16], TwPtr tw)
{
    const float2 one = tw[0];
#pragma unroll
    for (int d1 = 0; d1 < 4; ++d1) bfly4<INV>(v[d1], v[d1 + 4], v[d1 + 8], v[d1 + 12], one, one, one);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2)
        bfly4<INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3], tw[4 * k2], tw[8 * k2], tw[12 * k2]);
}

// the same two stages with the twiddles already in registers (wave-uniform: a1[k2] = tw[4 k2], a2[k2] = tw[8 k2], a3[k2] = tw[12 k2]): the
// kernel loads them ONCE per wavefront -- fetched inside the tile loop they are scalar loads whose wait (lgkmcnt(0)) also waits for LDS traffic
template <bool INV>
RD_HD void pfb_fft64_passAB_pre(float2 (&v)[16], float2 one, const float2 (&a1)[4], const float2 (&a2)[4], const float2 (&a3)[4])
{
#pragma unroll
    for (int d1 = 0; d1 < 4; ++d1) bfly4<INV>(v[d1], v[d1 + 4], v[d1 + 8], v[d1 + 12], one, one, one);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) bfly4<INV>(v[4 * k2], v[4 * k2 + 1], v[4 * k2 + 2], v[4 * k2 + 3], a1[k2], a2[k2], a3[k2]);
}

// exchange 2: lane (tt, d0) stores value (k1, k2); lane (tt, k1) loads element f = d0 + 4 k2
RD_HD int pfb_x2_store(int lane, int k1, int k2) { return (lane >> 2) * PFB_ROW + 16 * k1 + 4 * k2 + (lane & 3); }
RD_HD int pfb_x2_load(int lane, int f) { return (lane >> 2) * PFB_ROW + 16 * (lane & 3) + f; }

// stage over d0 (m = 16, fstride 1, k = k2 + 4 k1): w[d0 + 4 k2] -> w[k0 + 4 k2] = X[k2 + 4 k1 + 16 k0]
template <bool INV, typename TwPtr>
RD_HD void pfb_fft64_passC(float2 (&w)[16], int lane, TwPtr tw)
{
    const int k1 = lane & 3;
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        const int k = k2 + 4 * k1;
        bfly4<INV>(w[4 * k2], w[4 * k2 + 1], w[4 * k2 + 2], w[4 * k2 + 3], tw[k], tw[2 * k], tw[3 * k]);
    }
}
// the same stage with the lane's twiddles already in registers (w1[k2] = tw[k], w2[k2] = tw[2 k], w3[k2] = tw[3 k], k = k2 + 4 (lane & 3)):
// fetched inside the tile loop they are ten vector loads per tile whose wait (vmcnt(0): loads return in order) also waits for the NEXT tile's
// rows requested at the top of the loop -- the prefetch then has to land within the tile that issued it (round 5)
template <bool INV>
RD_HD void pfb_fft64_passC_pre(float2 (&w)[16], const float2 (&w1)[4], const float2 (&w2)[4], const float2 (&w3)[4])
{
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) bfly4<INV>(w[4 * k2], w[4 * k2 + 1], w[4 * k2 + 2], w[4 * k2 + 3], w1[k2], w2[k2], w3[k2]);
}
// channel index of element (k0, k2) held by lane after pass C
RD_HD int pfb_out_channel(int lane, int k0, int k2) { return k2 + 4 * (lane & 3) + 16 * k0; }

// output addressing: natural [row][64], or grouped for the multi-GPU exchange:
// [group][row][cpg] with cpg = 64 / ngroups channels per group (ngroups = 1 is the natural layout)
RD_HD long pfb_out_index(long row, int ch, long rows_total, int ngroups)
{
    const int cpg = PFB_M / ngroups;
    return (long)(ch / cpg) * rows_total * cpg + row * cpg + (ch % cpg);
}

} // namespace redio
