// fft_core.h -- butterfly arithmetic and lane/LDS index maps of the gfx950 FFT kernels.
//
// Arithmetic contract: kissfft::fft (src/kissfft/src/kissfft.rs:18-31) hands each block to
// kiss_fft(), whose published algorithm (kissfft 1.3.0 kiss_fft.c) is decimation in time over the
// factor list "4s, then 2s, then 3, 5, odd primes" with a float twiddle table.  The butterflies
// below keep that operation order, one rounding per multiply/add and no FMA contraction, so a
// transform is bit-identical to the CPU algorithm; only the data movement is redesigned for a
// 64-lane wavefront and LDS.
//
// Host-compilable (tests/emu runs the same lane programs on the CPU, one lane at a time).
#pragma once
#include "redio_device.h"
#include <math.h>

namespace redio {

// bfly4 with the packed-f32 instructions picked by hand: 3 per complex product, 1 per complex sum = 17 per butterfly and no
// register shuffles (the compiler's own packing of bfly4 spends about 30).  Every product and sum is rounded on its own, in
// the same association as bfly4, so the results are the same bits.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(REDIO_NO_PK_BFLY)
#define REDIO_PK_BFLY 1
typedef float redio_pk2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ redio_pk2 pk_cmul(redio_pk2 a, redio_pk2 t)
{
    redio_pk2 p, r, o;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(t));               // (a.x t.x, a.y t.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t));   // (a.y t.y, a.x t.y)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(o) : "v"(p), "v"(r));                   // (p.x - r.x, p.y + r.y)
    return o;
}
// the three instructions of pk_cmul on their own, so that a butterfly can keep dependent packed operations two issue slots
// apart (gfx950 needs a wait state between a packed-f32 result and its use; the assembler-level order below avoids the
// s_nop the compiler would otherwise insert after nearly every pair)
__device__ __forceinline__ redio_pk2 pk_cmul_p(redio_pk2 a, redio_pk2 t)
{
    redio_pk2 p;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(t));
    return p;
}
__device__ __forceinline__ redio_pk2 pk_cmul_r(redio_pk2 a, redio_pk2 t)
{
    redio_pk2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(t));
    return r;
}
__device__ __forceinline__ redio_pk2 pk_cmul_o(redio_pk2 p, redio_pk2 r)
{
    redio_pk2 o;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(o) : "v"(p), "v"(r));
    return o;
}
__device__ __forceinline__ redio_pk2 pk_add_rot_a(redio_pk2 a, redio_pk2 b) // (a.x + b.y, a.y - b.x)
{
    redio_pk2 o;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
__device__ __forceinline__ redio_pk2 pk_add_rot_b(redio_pk2 a, redio_pk2 b) // (a.x - b.y, a.y + b.x)
{
    redio_pk2 o;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
__device__ __forceinline__ redio_pk2 pk_add2(redio_pk2 a, redio_pk2 b)
{
    redio_pk2 o;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
__device__ __forceinline__ redio_pk2 pk_sub2(redio_pk2 a, redio_pk2 b)
{
    redio_pk2 o;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o) : "v"(a), "v"(b));
    return o;
}
#endif

// ---- radix-4 butterfly (kf_bfly4 order) ------------------------------------------------------
template <bool INV, bool PK = true> // PK = false: leave the instruction selection to the compiler (same arithmetic)
RD_HD void bfly4(float2 &a0, float2 &a1, float2 &a2, float2 &a3, float2 t1, float2 t2, float2 t3)
{
#if defined(REDIO_PK_BFLY)
    if constexpr (PK) {
    const redio_pk2 x0 = {a0.x, a0.y}, x1 = {a1.x, a1.y}, x2 = {a2.x, a2.y}, x3 = {a3.x, a3.y};
    const redio_pk2 w1 = {t1.x, t1.y}, w2 = {t2.x, t2.y}, w3 = {t3.x, t3.y};
    const redio_pk2 p1 = pk_cmul_p(x2, w2), r1 = pk_cmul_r(x2, w2);
    const redio_pk2 p0 = pk_cmul_p(x1, w1), r0 = pk_cmul_r(x1, w1);
    const redio_pk2 p2 = pk_cmul_p(x3, w3), r2 = pk_cmul_r(x3, w3);
    const redio_pk2 s1 = pk_cmul_o(p1, r1), s0 = pk_cmul_o(p0, r0), s2 = pk_cmul_o(p2, r2);
    const redio_pk2 s5 = pk_sub2(x0, s1), y0 = pk_add2(x0, s1);
    const redio_pk2 s3 = pk_add2(s0, s2), s4 = pk_sub2(s0, s2);
    const redio_pk2 o2 = pk_sub2(y0, s3), o0 = pk_add2(y0, s3);
    const redio_pk2 o1 = INV ? pk_add_rot_b(s5, s4) : pk_add_rot_a(s5, s4);
    const redio_pk2 o3 = INV ? pk_add_rot_a(s5, s4) : pk_add_rot_b(s5, s4);
    a0 = make_float2(o0.x, o0.y); a1 = make_float2(o1.x, o1.y); a2 = make_float2(o2.x, o2.y); a3 = make_float2(o3.x, o3.y);
    return;
    }
#endif
    float2 s0 = cmul_rn(a1, t1);
    float2 s1 = cmul_rn(a2, t2);
    float2 s2 = cmul_rn(a3, t3);
    float2 s5 = csub_rn(a0, s1);
    a0 = cadd_rn(a0, s1);
    float2 s3 = cadd_rn(s0, s2);
    float2 s4 = csub_rn(s0, s2);
    a2 = csub_rn(a0, s3);
    a0 = cadd_rn(a0, s3);
    if (INV) {
        a1 = make_float2(sub_rn(s5.x, s4.y), add_rn(s5.y, s4.x));
        a3 = make_float2(add_rn(s5.x, s4.y), sub_rn(s5.y, s4.x));
    } else {
        a1 = make_float2(add_rn(s5.x, s4.y), sub_rn(s5.y, s4.x));
        a3 = make_float2(sub_rn(s5.x, s4.y), add_rn(s5.y, s4.x));
    }
}

// two independent radix-4 butterflies with their packed operations interleaved (A, B, A, B ...): every result is used at
// least three issue slots after it is produced, so no wait states are needed between the dependent packed operations
template <bool INV>
RD_HD void bfly4x2(float2 &a0, float2 &a1, float2 &a2, float2 &a3, float2 ta1, float2 ta2, float2 ta3,
                   float2 &b0, float2 &b1, float2 &b2, float2 &b3, float2 tb1, float2 tb2, float2 tb3)
{
#if defined(REDIO_PK_BFLY)
    const redio_pk2 xa0 = {a0.x, a0.y}, xa1 = {a1.x, a1.y}, xa2 = {a2.x, a2.y}, xa3 = {a3.x, a3.y};
    const redio_pk2 xb0 = {b0.x, b0.y}, xb1 = {b1.x, b1.y}, xb2 = {b2.x, b2.y}, xb3 = {b3.x, b3.y};
    const redio_pk2 wa1 = {ta1.x, ta1.y}, wa2 = {ta2.x, ta2.y}, wa3 = {ta3.x, ta3.y};
    const redio_pk2 wb1 = {tb1.x, tb1.y}, wb2 = {tb2.x, tb2.y}, wb3 = {tb3.x, tb3.y};
    const redio_pk2 pa1 = pk_cmul_p(xa2, wa2), pb1 = pk_cmul_p(xb2, wb2), ra1 = pk_cmul_r(xa2, wa2), rb1 = pk_cmul_r(xb2, wb2);
    const redio_pk2 pa0 = pk_cmul_p(xa1, wa1), pb0 = pk_cmul_p(xb1, wb1), ra0 = pk_cmul_r(xa1, wa1), rb0 = pk_cmul_r(xb1, wb1);
    const redio_pk2 pa2 = pk_cmul_p(xa3, wa3), pb2 = pk_cmul_p(xb3, wb3), ra2 = pk_cmul_r(xa3, wa3), rb2 = pk_cmul_r(xb3, wb3);
    const redio_pk2 sa1 = pk_cmul_o(pa1, ra1), sb1 = pk_cmul_o(pb1, rb1);
    const redio_pk2 sa0 = pk_cmul_o(pa0, ra0), sb0 = pk_cmul_o(pb0, rb0);
    const redio_pk2 sa2 = pk_cmul_o(pa2, ra2), sb2 = pk_cmul_o(pb2, rb2);
    const redio_pk2 sa5 = pk_sub2(xa0, sa1), sb5 = pk_sub2(xb0, sb1);
    const redio_pk2 ya0 = pk_add2(xa0, sa1), yb0 = pk_add2(xb0, sb1);
    const redio_pk2 sa3 = pk_add2(sa0, sa2), sb3 = pk_add2(sb0, sb2);
    const redio_pk2 sa4 = pk_sub2(sa0, sa2), sb4 = pk_sub2(sb0, sb2);
    const redio_pk2 oa2 = pk_sub2(ya0, sa3), ob2 = pk_sub2(yb0, sb3);
    const redio_pk2 oa0 = pk_add2(ya0, sa3), ob0 = pk_add2(yb0, sb3);
    const redio_pk2 oa1 = INV ? pk_add_rot_b(sa5, sa4) : pk_add_rot_a(sa5, sa4), ob1 = INV ? pk_add_rot_b(sb5, sb4) : pk_add_rot_a(sb5, sb4);
    const redio_pk2 oa3 = INV ? pk_add_rot_a(sa5, sa4) : pk_add_rot_b(sa5, sa4), ob3 = INV ? pk_add_rot_a(sb5, sb4) : pk_add_rot_b(sb5, sb4);
    a0 = make_float2(oa0.x, oa0.y); a1 = make_float2(oa1.x, oa1.y); a2 = make_float2(oa2.x, oa2.y); a3 = make_float2(oa3.x, oa3.y);
    b0 = make_float2(ob0.x, ob0.y); b1 = make_float2(ob1.x, ob1.y); b2 = make_float2(ob2.x, ob2.y); b3 = make_float2(ob3.x, ob3.y);
#else
    bfly4<INV>(a0, a1, a2, a3, ta1, ta2, ta3);
    bfly4<INV>(b0, b1, b2, b3, tb1, tb2, tb3);
#endif
}

// ---- radix-2 butterfly (kf_bfly2 order) ------------------------------------------------------
RD_HD void bfly2(float2 &a0, float2 &a1, float2 t1)
{
    float2 t = cmul_rn(a1, t1);
    a1 = csub_rn(a0, t);
    a0 = cadd_rn(a0, t);
}

// ---- radix-3 butterfly (kf_bfly3 order; HALF_OF multiplies by a double literal) ---------------
RD_HD void bfly3(float2 &a0, float2 &a1, float2 &a2, float2 t1, float2 t2, float2 epi3)
{
    float2 s1 = cmul_rn(a1, t1);
    float2 s2 = cmul_rn(a2, t2);
    float2 s3 = cadd_rn(s1, s2);
    float2 s0 = csub_rn(s1, s2);
    a1.x = (float)((double)a0.x - (double)s3.x * .5);
    a1.y = (float)((double)a0.y - (double)s3.y * .5);
    s0.x = mul_rn(s0.x, epi3.y);
    s0.y = mul_rn(s0.y, epi3.y);
    a0 = cadd_rn(a0, s3);
    a2.x = add_rn(a1.x, s0.y);
    a2.y = sub_rn(a1.y, s0.x);
    a1.x = sub_rn(a1.x, s0.y);
    a1.y = add_rn(a1.y, s0.x);
}

// ---- radix-5 butterfly (kf_bfly5 order) ------------------------------------------------------
RD_HD void bfly5(float2 &a0, float2 &a1, float2 &a2, float2 &a3, float2 &a4, float2 t1, float2 t2,
                 float2 t3, float2 t4, float2 ya, float2 yb)
{
    float2 s0 = a0;
    float2 s1 = cmul_rn(a1, t1);
    float2 s2 = cmul_rn(a2, t2);
    float2 s3 = cmul_rn(a3, t3);
    float2 s4 = cmul_rn(a4, t4);
    float2 s7 = cadd_rn(s1, s4), s10 = csub_rn(s1, s4);
    float2 s8 = cadd_rn(s2, s3), s9 = csub_rn(s2, s3);
    a0.x = add_rn(a0.x, add_rn(s7.x, s8.x));
    a0.y = add_rn(a0.y, add_rn(s7.y, s8.y));
    float2 s5, s6, s11, s12;
    s5.x = add_rn(add_rn(s0.x, mul_rn(s7.x, ya.x)), mul_rn(s8.x, yb.x));
    s5.y = add_rn(add_rn(s0.y, mul_rn(s7.y, ya.x)), mul_rn(s8.y, yb.x));
    s6.x = add_rn(mul_rn(s10.y, ya.y), mul_rn(s9.y, yb.y));
    s6.y = sub_rn(-mul_rn(s10.x, ya.y), mul_rn(s9.x, yb.y));
    a1 = csub_rn(s5, s6);
    a4 = cadd_rn(s5, s6);
    s11.x = add_rn(add_rn(s0.x, mul_rn(s7.x, yb.x)), mul_rn(s8.x, ya.x));
    s11.y = add_rn(add_rn(s0.y, mul_rn(s7.y, yb.x)), mul_rn(s8.y, ya.x));
    s12.x = add_rn(-mul_rn(s10.y, yb.y), mul_rn(s9.y, ya.y));
    s12.y = sub_rn(mul_rn(s10.x, yb.y), mul_rn(s9.x, ya.y));
    a2 = cadd_rn(s11, s12);
    a3 = csub_rn(s11, s12);
}

// ============================================================================================
// 1024-point transform by ONE wavefront: 64 lanes x 16 points, three register passes
// (stages m=1,4 | m=16,64 | m=256 of the 4^5 factorisation) with two LDS exchanges.
//
// Input index n = d0 + 4 d1 + 16 d2 + 64 d3 + 256 d4.  Stage s (m = 4^s) transforms digit d(4-s)
// into output digit k(4-s) of weight 4^s, so X[k4 + 4 k3 + 16 k2 + 64 k1 + 256 k0].
//
// LDS image (one per transform, 16 rows x FFT1K_ROW float2 = 8704 B; the 4-float2 row pad makes
// every exchange conflict-free for ds_write_b64 16-lane groups and ds_read_b64 32-lane groups):
//   L1 (after pass A):  row = d1 + 4 d2,   col = d0 + 4 k4 + 16 k3
//   L2 (after pass B):  row = 4 k1 + d0,   col = k4 + 4 k3 + 16 k2
// ============================================================================================
constexpr int FFT1K_ROW = 68;
constexpr int FFT1K_LDS = 16 * FFT1K_ROW; // float2 elements per transform

// Pass A.  lane = d0 + 4 d1 + 16 d2 holds v[t] = x[lane + 64 t], t = d3 + 4 d4.
// tw points at the 1024-entry table of the plan (forward or inverse).
template <bool INV, typename TwPtr>
RD_HD void fft1k_passA(float2 (&v)[16], TwPtr tw)
{
    const float2 one = tw[0];
#pragma unroll
    for (int d3 = 0; d3 < 4; ++d3) // stage m=1 (fstride 256): k = 0, every twiddle is tw[0]
        bfly4<INV>(v[d3], v[d3 + 4], v[d3 + 8], v[d3 + 12], one, one, one);
#pragma unroll
    for (int k4 = 0; k4 < 4; ++k4) // stage m=4 (fstride 64): k = k4
        bfly4<INV>(v[4 * k4], v[4 * k4 + 1], v[4 * k4 + 2], v[4 * k4 + 3], tw[64 * k4], tw[128 * k4], tw[192 * k4]);
    // now v[k3 + 4 k4]
}
RD_HD int fft1k_A_store(int lane, int k3, int k4) { return (lane & 3) + 4 * k4 + 16 * k3 + FFT1K_ROW * (lane >> 2); }

// Pass B.  lane = d0 + 4 k4 + 16 k3 (the L1 column) holds u[e], e = d1 + 4 d2 (the L1 row).
RD_HD int fft1k_B_load(int lane, int e) { return lane + FFT1K_ROW * e; }
// The 27 lane-dependent twiddles of passes B and C (pass A's are wave-uniform).  A persistent
// kernel loads them once per lane and keeps them in registers.
struct Fft1kTw {
    float2 b[15]; // [0..2] stage m=16; [3 + 3*k2 .. ] stage m=64
    float2 c[12]; // [3*q ..] stage m=256
};
template <typename TwPtr>
RD_HD void fft1k_load_tw(Fft1kTw &t, int lane, TwPtr tw)
{
    const int k = lane >> 2; // pass B lane = d0 + 4 k4 + 16 k3  ->  k = k4 + 4 k3
    t.b[0] = tw[16 * k]; t.b[1] = tw[32 * k]; t.b[2] = tw[48 * k];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        const int kk = k + 16 * k2;
        t.b[3 + 3 * k2] = tw[4 * kk]; t.b[4 + 3 * k2] = tw[8 * kk]; t.b[5 + 3 * k2] = tw[12 * kk];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { // pass C lane = k4 + 4 k3 + 16 k2  ->  k = lane + 64 q
        const int kc = lane + 64 * q;
        t.c[3 * q] = tw[kc]; t.c[3 * q + 1] = tw[2 * kc]; t.c[3 * q + 2] = tw[3 * kc];
    }
}

template <bool INV>
RD_HD void fft1k_passB(float2 (&u)[16], const Fft1kTw &t)
{
#pragma unroll
    for (int d1 = 0; d1 < 4; ++d1) // stage m=16, fstride 16: k = k4 + 4 k3
        bfly4<INV>(u[d1], u[d1 + 4], u[d1 + 8], u[d1 + 12], t.b[0], t.b[1], t.b[2]);
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) // stage m=64, fstride 4: k = k4 + 4 k3 + 16 k2
        bfly4<INV>(u[4 * k2], u[4 * k2 + 1], u[4 * k2 + 2], u[4 * k2 + 3], t.b[3 + 3 * k2], t.b[4 + 3 * k2], t.b[5 + 3 * k2]);
    // now u[k1 + 4 k2]
}
RD_HD int fft1k_B_store(int lane, int k1, int k2) { return (lane >> 2) + 16 * k2 + FFT1K_ROW * (4 * k1 + (lane & 3)); }

// Pass C.  lane = k4 + 4 k3 + 16 k2 (the L2 column) holds w[4 q + j], q = k1, j = d0 (the L2 row).
RD_HD int fft1k_C_load(int lane, int q, int j) { return lane + FFT1K_ROW * (4 * q + j); }
template <bool INV>
RD_HD void fft1k_passC(float2 (&w)[16], const Fft1kTw &t)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) // stage m=256, fstride 1: k = lane + 64 q
        bfly4<INV>(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3], t.c[3 * q], t.c[3 * q + 1], t.c[3 * q + 2]);
    // w[4 q + j] is X[lane + 64 q + 256 j]
}

// ============================================================================================
// 1024-point transform in the FIR's own register layout ("native" form, used by the fused chain):
// the wavefront already holds the block as a[4 s + r] = x[256 s + 4 lane + r] (s = FIR sub-tile,
// r = the lane's four consecutive outputs), i.e. with n = d0 + 4 d1 + 16 d2 + 64 d3 + 256 d4:
// r = d0, lane = d1 + 4 d2 + 16 d3, s = d4.  Stage m=1 (over d4) and stage m=256 (over d0) are then
// register-only, and the transform needs no input load at all:
//     stage 0 | exchange X1 | stages 1,2 | exchange X2 | stages 3,4 -> X[l2 + 64 k1 + 256 k0]
// X1 image: row = 4 k4 + d0 (stride 68), col = d1 + 4 d2 + 16 d3;  reader lane l1 = d1 + 4 d0 + 16 k4
// X2 image: row = d0 + 4 d1 (stride 66), col = k4 + 4 k3 + 16 k2;  reader lane l2 = the column
// Both strides make the b64 writes (16-lane groups) and reads (32-lane groups) conflict-free.
// ============================================================================================
constexpr int FFT1KN_ROW1 = 68, FFT1KN_ROW2 = 66;
constexpr int FFT1KN_LDS = 16 * FFT1KN_ROW1; // float2 per wave

template <bool INV, typename TwPtr>
RD_HD void fft1kn_stage0(float2 (&a)[16], TwPtr tw)
{
    const float2 one = tw[0]; // stage m=1: k = 0
#pragma unroll
    for (int r = 0; r < 4; ++r) bfly4<INV>(a[r], a[4 + r], a[8 + r], a[12 + r], one, one, one);
    // now a[4 k4 + d0]
}
RD_HD int fft1kn_x1_store(int lane, int k4, int d0) { return (4 * k4 + d0) * FFT1KN_ROW1 + lane; }
// lane l1 = d1 + 4 d0 + 16 k4 reads element e = d2 + 4 d3
RD_HD int fft1kn_x1_load(int lane, int e) { return (4 * (lane >> 4) + ((lane >> 2) & 3)) * FFT1KN_ROW1 + (lane & 3) + 4 * e; }

struct Fft1knTw12 { float2 a[3]; float2 b[12]; };
struct Fft1knTw34 { float2 c[3]; float2 d[12]; };
template <typename TwPtr>
RD_HD void fft1kn_load_tw12(Fft1knTw12 &t, int lane, TwPtr tw)
{
    const int k4 = lane >> 4;
    t.a[0] = tw[64 * k4]; t.a[1] = tw[128 * k4]; t.a[2] = tw[192 * k4];
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) {
        const int k = k4 + 4 * k3;
        t.b[3 * k3] = tw[16 * k]; t.b[3 * k3 + 1] = tw[32 * k]; t.b[3 * k3 + 2] = tw[48 * k];
    }
}
template <typename TwPtr>
RD_HD void fft1kn_load_tw34(Fft1knTw34 &t, int lane, TwPtr tw)
{
    t.c[0] = tw[4 * lane]; t.c[1] = tw[8 * lane]; t.c[2] = tw[12 * lane];
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        const int k = lane + 64 * k1;
        t.d[3 * k1] = tw[k]; t.d[3 * k1 + 1] = tw[2 * k]; t.d[3 * k1 + 2] = tw[3 * k];
    }
}
// u[d2 + 4 d3] -> u[k2 + 4 k3]
template <bool INV>
RD_HD void fft1kn_pass12(float2 (&u)[16], const Fft1knTw12 &t)
{
#pragma unroll
    for (int d2 = 0; d2 < 4; ++d2) // stage m=4 (over d3), k = k4
        bfly4<INV>(u[d2], u[d2 + 4], u[d2 + 8], u[d2 + 12], t.a[0], t.a[1], t.a[2]);
#pragma unroll
    for (int k3 = 0; k3 < 4; ++k3) // stage m=16 (over d2), k = k4 + 4 k3
        bfly4<INV>(u[4 * k3], u[4 * k3 + 1], u[4 * k3 + 2], u[4 * k3 + 3], t.b[3 * k3], t.b[3 * k3 + 1], t.b[3 * k3 + 2]);
}
// lane l1 = d1 + 4 d0 + 16 k4 stores value (k2, k3)
RD_HD int fft1kn_x2_store(int lane, int k2, int k3)
{
    return (((lane >> 2) & 3) + 4 * (lane & 3)) * FFT1KN_ROW2 + (lane >> 4) + 4 * k3 + 16 * k2;
}
// lane l2 = k4 + 4 k3 + 16 k2 reads element f = d1 + 4 d0
RD_HD int fft1kn_x2_load(int lane, int f) { return ((f >> 2) + 4 * (f & 3)) * FFT1KN_ROW2 + lane; }
// w[d1 + 4 d0] -> w[k1 + 4 k0] = X[lane + 64 k1 + 256 k0]
template <bool INV>
RD_HD void fft1kn_pass34(float2 (&w)[16], const Fft1knTw34 &t)
{
#pragma unroll
    for (int d0 = 0; d0 < 4; ++d0) // stage m=64 (over d1), k = lane
        bfly4<INV>(w[4 * d0], w[4 * d0 + 1], w[4 * d0 + 2], w[4 * d0 + 3], t.c[0], t.c[1], t.c[2]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) // stage m=256 (over d0), k = lane + 64 k1
        bfly4<INV>(w[k1], w[k1 + 4], w[k1 + 8], w[k1 + 12], t.d[3 * k1], t.d[3 * k1 + 1], t.d[3 * k1 + 2]);
}

// ============================================================================================
// Generic mixed-radix stage (any nfft): butterfly b of the stage with radix p, sub-length m and
// twiddle stride fstride, operating in place on F (LDS or global).  Positions g*p*m + k + j*m.
// Radix > 5 is done out of place (src -> dst) because kf_bfly_generic needs all p inputs.
// ============================================================================================
struct FftStage { int p, m, fstride; };

// position P of the decimation-in-time leaf copy holds input index n: digits of P over (m_s) map to
// digits of n over (fstride_s)
RD_HD int fft_leaf_source(int P, const FftStage *st, int nstages)
{
    int n = 0;
    for (int s = 0; s < nstages; ++s) {
        int d = P / st[s].m;
        P -= d * st[s].m;
        n += d * st[s].fstride;
    }
    return n;
}

// Stage list of a plan: the factor order of the published kissfft (4s, then 2s, then 3, 5, 7, ...;
// a trial factor above floor(sqrt(n)) is replaced by what is left).  Host side; returns the number
// of stages or -1 when max_stages is too small.
inline int fft_plan_stages(int n, FftStage *st, int max_stages)
{
    int p = 4, cnt = 0, fstride = 1;
    const double floor_sqrt = floor(sqrt((double)n));
    do {
        while (n % p) {
            switch (p) {
            case 4: p = 2; break;
            case 2: p = 3; break;
            default: p += 2; break;
            }
            if (p > floor_sqrt) p = n;
        }
        n /= p;
        if (cnt >= max_stages) return -1;
        st[cnt].p = p; st[cnt].m = n; st[cnt].fstride = fstride;
        fstride *= p;
        ++cnt;
    } while (n > 1);
    return cnt;
}

// butterfly (g, k) of a stage: group g of p*m positions, index k inside the sub-length m
template <bool INV, typename Ptr, typename TwPtr>
RD_HD void fft_stage_butterfly_gk(Ptr F, TwPtr tw, FftStage s, int g, int k)
{
    const int base = g * s.p * s.m + k;
    const int m = s.m, fs = s.fstride;
    if (s.p == 4) {
        float2 a0 = F[base], a1 = F[base + m], a2 = F[base + 2 * m], a3 = F[base + 3 * m];
        bfly4<INV>(a0, a1, a2, a3, tw[k * fs], tw[2 * k * fs], tw[3 * k * fs]);
        F[base] = a0; F[base + m] = a1; F[base + 2 * m] = a2; F[base + 3 * m] = a3;
    } else if (s.p == 2) {
        float2 a0 = F[base], a1 = F[base + m];
        bfly2(a0, a1, tw[k * fs]);
        F[base] = a0; F[base + m] = a1;
    } else if (s.p == 3) {
        float2 a0 = F[base], a1 = F[base + m], a2 = F[base + 2 * m];
        bfly3(a0, a1, a2, tw[k * fs], tw[2 * k * fs], tw[fs * m]);
        F[base] = a0; F[base + m] = a1; F[base + 2 * m] = a2;
    } else if (s.p == 5) {
        float2 a0 = F[base], a1 = F[base + m], a2 = F[base + 2 * m], a3 = F[base + 3 * m], a4 = F[base + 4 * m];
        bfly5(a0, a1, a2, a3, a4, tw[k * fs], tw[2 * k * fs], tw[3 * k * fs], tw[4 * k * fs], tw[fs * m], tw[fs * 2 * m]);
        F[base] = a0; F[base + m] = a1; F[base + 2 * m] = a2; F[base + 3 * m] = a3; F[base + 4 * m] = a4;
    } // p == 1 (nfft == 1): kf_bfly_generic with one input is the identity
}

template <bool INV, typename Ptr, typename TwPtr>
RD_HD void fft_stage_butterfly(Ptr F, TwPtr tw, FftStage s, int b)
{
    const int g = b / s.m;
    fft_stage_butterfly_gk<INV>(F, tw, s, g, b - g * s.m);
}

// kf_bfly_generic, one OUTPUT element per call: out position base + q1*m of butterfly (g,u).
// Fout[k] = scratch[0] + sum_{q=1}^{p-1} scratch[q] * tw[(q * fstride * k) mod nfft], accumulated
// in that order with the running twidx of the published code.
template <typename SrcPtr, typename TwPtr>
RD_HD float2 fft_generic_output(SrcPtr src, TwPtr tw, FftStage s, int nfft, int g, int u, int q1)
{
    const int base = g * s.p * s.m;
    const int k = u + q1 * s.m; // index inside this sub-transform, as in the published loop
    float2 acc = src[base + u];
    int twidx = 0;
    for (int q = 1; q < s.p; ++q) {
        twidx += s.fstride * k;
        if (twidx >= nfft) twidx -= nfft;
        float2 t = cmul_rn(src[base + u + q * s.m], tw[twidx]);
        acc = cadd_rn(acc, t);
    }
    return acc;
}

} // namespace redio

// ============================================================================================
// 65536-point transform (4^8) in two passes over global memory, four radix-4 stages each, same
// butterflies and twiddle table as the one-stage-per-launch path (bit-identical results).
// Position P = 256 G + p after the decimation-in-time leaf copy holds input n = rev4(G) + 256 rev4(p)
// (rev4 = reversal of four base-4 digits).
//   pass 0: stages m = 1, 4, 16, 64 act inside each group G of 256 consecutive positions;
//   pass 1: stages m = 256 ... 16384 act, for each k0 = P mod 256, on the 256 positions k0 + 256 G.
// A workgroup takes a tile of 256 rows x F64K_COLS columns whose rows are F64K_COLS contiguous
// samples in memory (pass 0: columns = consecutive rev4(G); pass 1: columns = consecutive k0), keeps
// it in LDS with a row stride of F64K_LD float2 and runs the four stages along the row index.
// ============================================================================================
namespace redio {

constexpr int F64K_N = 65536;
constexpr int F64K_COLS = 16;
constexpr int F64K_LD = F64K_COLS + 1;

RD_HD int rev4_of_8bit(int v) { return ((v & 3) << 6) | (((v >> 2) & 3) << 4) | (((v >> 4) & 3) << 2) | ((v >> 6) & 3); }

// global float2 index of tile element (row, col) of tile `c` (0..15) of a transform
//   pass 0 load : source n = (16 c + col) + 256 * row                 -> LDS row rev4(row)
//   pass 0 store: position P = 256 * rev4(16 c + col) + row
//   pass 1 load/store: position P = (16 c + col) + 256 * row
RD_HD int f64k_p0_src(int c, int row, int col) { return F64K_COLS * c + col + 256 * row; }
RD_HD int f64k_p0_dst(int c, int row, int col) { return 256 * rev4_of_8bit(F64K_COLS * c + col) + row; }
RD_HD int f64k_p1_pos(int c, int row, int col) { return F64K_COLS * c + col + 256 * row; }

// butterfly b (0..63) of in-tile stage t (sub-length 4^t along the row index) for column `col`
template <bool INV, typename Ptr, typename TwPtr>
RD_HD void f64k_tile_butterfly(Ptr L, TwPtr tw, int pass, int t, int col, int b, int k0)
{
    const int m = 1 << (2 * t);
    const int grp = b >> (2 * t), kk = b & (m - 1);
    const int base = grp * 4 * m + kk;
    const int k = pass == 0 ? kk : k0 + 256 * kk;
    const int fs = pass == 0 ? (16384 >> (2 * t)) : (64 >> (2 * t));
    float2 a0 = L[(base)*F64K_LD + col], a1 = L[(base + m) * F64K_LD + col];
    float2 a2 = L[(base + 2 * m) * F64K_LD + col], a3 = L[(base + 3 * m) * F64K_LD + col];
    bfly4<INV>(a0, a1, a2, a3, tw[k * fs], tw[2 * k * fs], tw[3 * k * fs]);
    L[(base)*F64K_LD + col] = a0; L[(base + m) * F64K_LD + col] = a1;
    L[(base + 2 * m) * F64K_LD + col] = a2; L[(base + 3 * m) * F64K_LD + col] = a3;
}

// two in-tile stages t and t+1 (t = 0 or 2) on the 16 rows base + j*m of one column held in registers
// (grp = 0..15, blk = grp >> 2t, kk = grp & (m - 1), base = blk*16m + kk)
template <bool INV, typename TwPtr>
RD_HD void f64k_macro_regs(float2 (&a)[16], TwPtr tw, int pass, int t, int kk, int k0)
{
    const int m = 1 << (2 * t);
    {
        const int k = pass == 0 ? kk : k0 + 256 * kk;
        const int fs = pass == 0 ? (16384 >> (2 * t)) : (64 >> (2 * t));
        const float2 t1 = tw[(unsigned)(k * fs)], t2 = tw[(unsigned)(2 * k * fs)], t3 = tw[(unsigned)(3 * k * fs)];
        for (int q = 0; q < 4; q += 2)
            bfly4x2<INV>(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3], t1, t2, t3, a[4 * q + 4], a[4 * q + 5], a[4 * q + 6], a[4 * q + 7], t1, t2, t3);
    }
    {
        const int fs = pass == 0 ? (16384 >> (2 * t + 2)) : (64 >> (2 * t + 2));
        for (int u = 0; u < 4; u += 2) {
            const int k1 = kk + u * m, k1b = k1 + m; // index inside the 4m-block
            const int k = pass == 0 ? k1 : k0 + 256 * k1, kb = pass == 0 ? k1b : k0 + 256 * k1b;
            bfly4x2<INV>(a[u], a[u + 4], a[u + 8], a[u + 12], tw[(unsigned)(k * fs)], tw[(unsigned)(2 * k * fs)], tw[(unsigned)(3 * k * fs)],
                         a[u + 1], a[u + 5], a[u + 9], a[u + 13], tw[(unsigned)(kb * fs)], tw[(unsigned)(2 * kb * fs)], tw[(unsigned)(3 * kb * fs)]);
        }
    }
}

} // namespace redio
