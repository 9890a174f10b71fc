// src_core.h -- lane program of the register-blocked uniform-phase resampler kernels (src_kernels.hip),
// host-compilable so that tests/emu runs it one lane at a time against the plain per-output sums.
//
// samplerate::resample (src/samplerate/src/samplerate.rs:59-87) at a constant ratio 1/S, zero phase: output o of a tile is
//     left_o  = sum_{t=0..cl} L[t] * x[S*o + t]                (far end first, t = cl multiplies the centre sample)
//     right_o = sum_{t=0..cr} Rt[t] * x[S*o + c - t]           (far end first; c = cl + 1 + cr)
// each a strictly ordered double sum with separately rounded product and sum (calc_output_single of the published
// libsamplerate 0.1.8).  A lane owns ONE wing of R CONSECUTIVE outputs.  Walked by sample, in the order the wing
// visits them, sample i of the lane (i counted from the first sample of its first output) feeds accumulator r with
// coefficient index i - S*r (left) -- so one LDS read and one v_cvt_f64_f32 serve R taps, the coefficients stay
// wave-uniform (scalar loads of R runs of the table, S apart), and every accumulator still receives exactly its own
// products in its own order: the bits of the one-output-per-lane kernel and of oracle/oracle_src.c.  The right wing is
// the same program mirrored (m' = top - i, accumulator R-1-r).
//
// The walk goes in U-sample steps from the first sample any accumulator uses to the last, software-pipelined.
// In the (R-1)*S samples at either end some accumulators have no tap ("ramps"): those steps replace the product by
// +0.0 (a wave-uniform select; acc + 0.0 == acc bit for bit because a sum that started at +0.0 is never -0.0), so a
// sample outside an output's window can never reach it, whatever it holds (Inf, NaN, unwritten LDS).  The coefficient
// tables carry zero guard zones of at least (R-1)*S + 2*U entries on both sides (src_host.hip, prepare_uniform), so
// the scalar loads of those steps stay inside the allocation.
//
// LDS image: sample n of the tile sits at xs[n + P*(n / B)], B = R*S samples per lane, P = 0 or 4 pad floats per B
// chosen so that the lane stride (B + P)/4 is odd: a 16-lane group of ds_read_b128 then covers all 64 banks once.
// B must be a multiple of 4 (16-byte aligned groups of four samples that never straddle a pad).
#pragma once
#include "redio_device.h"

namespace redio {

RD_HD int src_rb_pad(int B) { return (B % 8 == 0) ? 4 : 0; }

#if defined(__HIP_DEVICE_COMPILE__)
#define RD_SRC_LAND_V(v) asm volatile("" : "+v"(v))
#define RD_SRC_LAND_S(s) asm volatile("" : "+s"(s))
// tells the compiler that a pointer / an int is wave-uniform (it is: built from kernel arguments and block indices)
__device__ __forceinline__ const double *src_uniform(const double *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const double *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int src_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
#define RD_SRC_LAND_V(v) ((void)0)
#define RD_SRC_LAND_S(s) ((void)0)
inline const double *src_uniform(const double *p) { return p; }
inline int src_uniform(int v) { return v; }
#endif

// The walk of one wing.  lane_xs: the lane's sample 0 in the LDS image (xs + q*(B + P)); i0: lane-relative index of the
// wing's first sample (DIR = +1: 0, walked upwards; DIR = -1: (R-1)*S + c, walked downwards); T: the wing's table, far end
// first, n entries.  acc[r'] belongs to output r' (DIR = +1) or R-1-r' (DIR = -1) of the lane.
template <int R, int DIR, bool PAD>
struct SrcRbWalk {
    const float *xp;                 // the lane's next group of four samples (low address)
    const char *__restrict__ tb;     // the table minus a bias that keeps every offset non-negative (wave-uniform)
    unsigned toff[R];                // byte offset of the next coefficient of every accumulator (wave-uniform)
    int B, P, S, n;
    int rem;                         // PAD: position of the next group inside its B-block (wave-uniform)
    double acc[R];

    template <int UU>
    RD_D void fetch(float4 (&x)[UU / 4], double (&k)[R][UU])
    {
#pragma unroll
        for (int g = 0; g < UU / 4; ++g) {
            x[g] = *reinterpret_cast<const float4 *>(xp);
            xp += DIR > 0 ? 4 : -4;
            if (PAD) { // the pad floats behind every B samples of the image
                if (DIR > 0) { rem += 4; if (rem >= B) { rem -= B; xp += P; } }
                else { rem -= 4; if (rem < 0) { rem += B; xp -= P; } }
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j = 0; j < UU; ++j) k[r][j] = *reinterpret_cast<const double *>(tb + toff[r] + 8u * j);
#if !defined(REDIO_EXP_SRC_FIXEDK) // timing-only experiment (results wrong): every coefficient load hits the same cache line
            toff[r] += 8u * UU;
#endif
        }
    }
    template <int UU>
    RD_D void land(float4 (&x)[UU / 4], double (&k)[R][UU]) // the step's one wait lands here, before the next issue
    {
#pragma unroll
        for (int g = 0; g < UU / 4; ++g) { RD_SRC_LAND_V(x[g].x); RD_SRC_LAND_V(x[g].y); RD_SRC_LAND_V(x[g].z); RD_SRC_LAND_V(x[g].w); }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < UU; ++j) RD_SRC_LAND_S(k[r][j]);
    }
    // RAMP = false: every (sample, accumulator) pair of the step has a tap.  RAMP = true: accumulator r has taps for the
    // samples j in [S*r - m, n + S*r - m) of the step only -- all of them (plain), none (skipped) or some: a pair without
    // a tap adds +0.0.  The three cases are wave-uniform branches.
    template <int UU, bool RAMP>
    RD_D void compute(int m, const float4 (&x)[UU / 4], const double (&k)[R][UU])
    {
        double xd[UU]; // the step's samples in the order the wing visits them
#pragma unroll
        for (int g = 0; g < UU / 4; ++g) {
            xd[4 * g + 0] = (double)(DIR > 0 ? x[g].x : x[g].w);
            xd[4 * g + 1] = (double)(DIR > 0 ? x[g].y : x[g].z);
            xd[4 * g + 2] = (double)(DIR > 0 ? x[g].z : x[g].y);
            xd[4 * g + 3] = (double)(DIR > 0 ? x[g].w : x[g].x);
        }
        if (!RAMP) {
#pragma unroll
            for (int j = 0; j < UU; ++j)
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] += k[r][j] * xd[j];
            return;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int lo = S * r - m, hi = n + S * r - m;
            if (lo <= 0 && hi >= UU) {
#pragma unroll
                for (int j = 0; j < UU; ++j) acc[r] += k[r][j] * xd[j];
            } else if (lo < UU && hi > 0) {
#pragma unroll
                for (int j = 0; j < UU; ++j) {
                    double p = k[r][j] * xd[j];
                    if (j < lo || j >= hi) p = 0.0;
                    acc[r] += p;
                }
            }
        }
    }
};

// U: samples per step.  ONE continuous two-deep software pipeline over the whole walk: the fetch of step i + 1 (LDS reads,
// scalar coefficient loads) is issued before the arithmetic of step i, also across the seams between ramp and body, and
// the walk's last step prefetches one step past the end (inside the table guards and the image slack).  No branch inside
// a step: ramp-up, body and ramp-down are three loops that differ in their compute only.  The body loop is written as
// two steps that use the two register sets alternately, so that nothing is copied between steps; the few ramp steps (and
// an odd body step) hand the prefetched set over by copy.
template <int R, int U, int DIR, bool PAD>
RD_D void src_rb_wing_t(const float *lane_xs, int B, int P, int i0, const double *__restrict__ T, int n, int S, double (&acc)[R])
{
    static_assert(U % 4 == 0 && U >= 4, "steps are whole 16-byte groups");
    const int mlo = DIR > 0 ? 0 : -((3 - (i0 & 3)) & 3); // the first group of a downward walk ends on an aligned top
    const int mend = n + (R - 1) * S;                     // one past the last m' any accumulator has a tap for
    SrcRbWalk<R, DIR, PAD> w;
    w.B = B; w.P = P; w.S = S; w.n = n;
    {
        const int ilo = DIR > 0 ? i0 + mlo : i0 - mlo - 3; // low index of the first group
        w.rem = ilo % B;
        w.xp = lane_xs + ilo + P * (ilo / B);
    }
    w.tb = reinterpret_cast<const char *>(T - ((R - 1) * S + 8));
#pragma unroll
    for (int r = 0; r < R; ++r) { w.acc[r] = acc[r]; w.toff[r] = 8u * (unsigned)(mlo - S * r + (R - 1) * S + 8); }
    const int nup = ((R - 1) * S - mlo + U - 1) / U;     // steps that start before every accumulator has begun
    const int mbody = mlo + nup * U;
    const int nbody = n > mbody ? (n - mbody) / U : 0;   // whole steps before the first accumulator ends
    const int mdown = mbody + nbody * U;
    const int ndown = (mend - mdown + U - 1) / U;
    float4 xA[U / 4], xB[U / 4];
    double kA[R][U], kB[R][U];
    int m = mlo;
    w.template fetch<U>(xA, kA);
    w.template land<U>(xA, kA);
#define RD_SRC_HALF(RAMP, XC, KC, XN, KN) /* compute the step in (XC, KC) while (XN, KN) are on their way */ \
    {                                                       \
        w.template fetch<U>(XN, KN);                        \
        RD_SCHED_BARRIER();                                 \
        w.template compute<U, RAMP>(m, XC, KC);             \
        RD_SCHED_BARRIER();                                 \
        w.template land<U>(XN, KN);                         \
        m += U;                                             \
    }
#define RD_SRC_STEP(RAMP)                                   \
    {                                                       \
        RD_SRC_HALF(RAMP, xA, kA, xB, kB)                   \
        _Pragma("unroll") for (int g = 0; g < U / 4; ++g) xA[g] = xB[g]; \
        _Pragma("unroll") for (int r = 0; r < R; ++r)       \
            _Pragma("unroll") for (int j = 0; j < U; ++j) kA[r][j] = kB[r][j]; \
    }
    for (int i = 0; i < nup; ++i) RD_SRC_STEP(true)
    for (int i = 0; i + 1 < nbody; i += 2) {
        RD_SRC_HALF(false, xA, kA, xB, kB)
        RD_SRC_HALF(false, xB, kB, xA, kA)
    }
    if (nbody & 1) RD_SRC_STEP(false)
    for (int i = 0; i < ndown; ++i) RD_SRC_STEP(true)
#undef RD_SRC_STEP
#undef RD_SRC_HALF
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = w.acc[r];
}

template <int R, int U, int DIR>
RD_D void src_rb_wing(const float *lane_xs, int B, int P, int i0, const double *__restrict__ T, int n, int S, double (&acc)[R])
{
    if (P) src_rb_wing_t<R, U, DIR, true>(lane_xs, B, P, i0, T, n, S, acc);
    else src_rb_wing_t<R, U, DIR, false>(lane_xs, B, P, i0, T, n, S, acc); // no pad floats: the image is linear, no cursor
}

// LDS floats of a tile of NO outputs (image with pads + slack behind it)
RD_HD long src_rb_tile_floats(int NO, int R, int S, int cl, int cr)
{
    const int B = R * S, P = src_rb_pad(B);
    const long span = (long)(NO - 1) * S + cl + cr + 2;
    return span + (long)P * (span / B + 2) + 8 + 16; // the pipeline prefetches one step (<= 8 samples) past a walk's end
}

} // namespace redio
