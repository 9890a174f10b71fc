// src_kernels.hip -- gfx950 kernels behind samplerate::resample (src/samplerate/src/samplerate.rs:59-87)
// and the src_* C symbols it binds (:32-42): band-limited interpolation over a tabulated half window
// (the published libsamplerate 0.1.8 sinc converter, mono).
//
// Output k of a call is a pure function of the stream and of (position, start_filter_index,
// increment, scale)_k, which the host state machine (src_host.cpp) produces with the same double
// recurrence the CPU library runs.  The kernel evaluates
//     left  = sum_{i=cl..0} c(start + i*inc) * x[pos - i]          (far end first)
//     right = sum_{i=cr..0} c(inc - start + i*inc) * x[pos + 1 + i]
//     out   = (float)(scale * (left + right))
// with c() = linear interpolation between adjacent float table entries at a 12-bit fixed-point
// index, everything accumulated in double in exactly that order -- one lane per output, so results
// are bit-identical to oracle/oracle_src.c.  Channels are independent streams (grid.y).
//
// Bound: VALU (about 91 multiply-adds plus two table reads and a lerp per input sample at ratio
// 1/50); the window and table reads are served by L1/L2.
#include "redio_internal.h"
#include "src_core.h"
#include <type_traits>

namespace redio {

constexpr int SRC_SHIFT_BITS = 12;

__device__ __forceinline__ double src_wing(const float *__restrict__ coeffs, const float *__restrict__ x,
                                           int filter_index, int increment, int data_index, int step, bool inclusive_zero)
{
    double acc = 0.0;
    const double inv_fp_one = 1.0 / (double)(1 << SRC_SHIFT_BITS);
    do {
        const double fraction = (double)(filter_index & ((1 << SRC_SHIFT_BITS) - 1)) * inv_fp_one;
        const int indx = filter_index >> SRC_SHIFT_BITS;
        const float c0 = coeffs[indx];
        const float dc = coeffs[indx + 1] - c0;
        const double icoeff = (double)c0 + fraction * (double)dc;
        acc += icoeff * (double)x[data_index];
        filter_index -= increment;
        data_index += step;
    } while (inclusive_zero ? filter_index >= 0 : filter_index > 0);
    return acc;
}

// win: [nchan][win_stride] floats, the stream window of this call (history + new input)
// pos/start/inc/scale: per output (shared by all channels)
__global__ __launch_bounds__(256) void src_sinc_exact_kernel(const float *__restrict__ win, long win_stride,
                                                             const float *__restrict__ coeffs, int coeff_half_len,
                                                             const int *__restrict__ pos, const int *__restrict__ start,
                                                             const int *__restrict__ inc, const double *__restrict__ scale,
                                                             float *__restrict__ out, long out_stride, long nout)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nout) return;
    const float *x = win + (long)blockIdx.y * win_stride;
    const int increment = inc[k];
    const int start_filter_index = start[k];
    const int b_current = pos[k];
    const int max_filter_index = coeff_half_len << SRC_SHIFT_BITS;

    int filter_index = start_filter_index;
    int coeff_count = (max_filter_index - filter_index) / increment;
    filter_index += coeff_count * increment;
    const double left = src_wing(coeffs, x, filter_index, increment, b_current - coeff_count, +1, true);

    filter_index = increment - start_filter_index;
    coeff_count = (max_filter_index - filter_index) / increment;
    filter_index += coeff_count * increment;
    const double right = src_wing(coeffs, x, filter_index, increment, b_current + 1 + coeff_count, -1, false);

    out[(long)blockIdx.y * out_stride + k] = (float)(scale[k] * (left + right));
}

hipError_t launch_src_exact(const float *win, long win_stride, const float *coeffs, int coeff_half_len,
                            const int *pos, const int *start, const int *inc, const double *scale,
                            float *out, long out_stride, long nout, int nchan, hipStream_t s)
{
    if (nout <= 0 || nchan <= 0) return hipSuccess;
    dim3 grid((unsigned)((nout + 255) / 256), (unsigned)nchan);
    hipLaunchKernelGGL(src_sinc_exact_kernel, grid, dim3(256), 0, s, win, win_stride, coeffs, coeff_half_len, pos, start, inc,
                       scale, out, out_stride, nout);
    return hipGetLastError();
}

// ---- periodic-phase path: rational ratios whose (start index, position step) sequence repeats --------------
// With a constant ratio p/q the fractional input position of output k + P equals that of output k (P outputs per
// Q input samples: 160 / 147 for 44.1 -> 48 kHz, 2 / 1 for ratio 2.0, 3 / 10 for 0.3 ...), so only P different sets of
// interpolated coefficients exist.  The host (src_host.hip, flush_epoch) verifies on the library's own per-output
// recurrence that the epoch IS periodic and builds, per phase, the coefficient of every tap the library's two wing
// loops would visit -- in double, with the library's expression -- as two tables [tap][phase].  A thread then owns
// ONE phase and G outputs that many periods apart: per tap it loads one coefficient (coalesced across the phases)
// and feeds G strictly ordered double accumulators from an LDS tile of the buffer image.  Same products in the
// same order as src_sinc_exact_kernel: bit-identical; no per-tap table interpolation, samples read from LDS.
struct SrcPeriodic {
    const double *Lc;  // [NL][P] left wing, far end first; row NL-1 multiplies x[pos]
    const double *Rc;  // [NR][P] right wing, far end first; row t multiplies x[pos + 1 + (NR-1) - t]
    const int *dpos;   // [P] buffer position of phase p relative to phase 0 of the same period (non-decreasing)
    const int *skipL;  // [P] leading rows of Lc / Rc that do not exist for this phase (its wing is shorter)
    const int *skipR;
    int P, Q, NL, NR, maxskipL, maxskipR, dpos_max;
};

template <int G>
__global__ __launch_bounds__(256) void src_sinc_periodic_kernel(const float *__restrict__ win, long win_stride, SrcPeriodic t, int pos0, int NT,
                                                                double scale, float *__restrict__ out, long out_stride, long nout)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, ch = blockIdx.y;
    const long k0 = (long)blockIdx.x * G * NT; // a whole number of periods
    const int periods = G * NT / t.P;
    const long tile_base = (long)pos0 + (k0 / t.P) * t.Q - (t.NL - 1); // buffer index of xs[0]
    const int span = (periods - 1) * t.Q + t.dpos_max + t.NL + t.NR + 1;
    const float *row = win + (long)ch * win_stride;
    for (int n = tid; n < span; n += 256) {
        long a = tile_base + n; // rows past the image belong to outputs that do not exist (or to skipped taps)
        a = a < 0 ? 0 : (a < win_stride ? a : win_stride - 1);
        xs[n] = row[a];
    }
    __syncthreads();
    if (tid >= NT) return;
    const int p = tid % t.P, mg = tid / t.P, per_g = NT / t.P;
    int base[G]; // xs index of x[pos] of item g
    double l[G], r[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        base[g] = (mg + g * per_g) * t.Q + t.dpos[p] + (t.NL - 1);
        l[g] = 0.0; r[g] = 0.0;
    }
    const int sl = t.skipL[p], sr = t.skipR[p];
    const double *lc = t.Lc + p, *rc = t.Rc + p;
    int tt = 0;
    for (; tt < t.maxskipL; ++tt) {
        const double c = lc[(long)tt * t.P];
        if (tt >= sl)
#pragma unroll
            for (int g = 0; g < G; ++g) l[g] += c * (double)xs[base[g] - (t.NL - 1) + tt];
    }
#pragma unroll 4
    for (; tt < t.NL; ++tt) {
        const double c = lc[(long)tt * t.P];
#pragma unroll
        for (int g = 0; g < G; ++g) l[g] += c * (double)xs[base[g] - (t.NL - 1) + tt];
    }
    for (tt = 0; tt < t.maxskipR; ++tt) {
        const double c = rc[(long)tt * t.P];
        if (tt >= sr)
#pragma unroll
            for (int g = 0; g < G; ++g) r[g] += c * (double)xs[base[g] + 1 + (t.NR - 1) - tt];
    }
#pragma unroll 4
    for (; tt < t.NR; ++tt) {
        const double c = rc[(long)tt * t.P];
#pragma unroll
        for (int g = 0; g < G; ++g) r[g] += c * (double)xs[base[g] + 1 + (t.NR - 1) - tt];
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const long k = k0 + tid + (long)g * NT;
        if (k < nout) out[(long)ch * out_stride + k] = (float)(scale * (l[g] + r[g]));
    }
}

// P phases (<= 256), Q input samples per period; *lds_bytes / *NT_out describe the launch; false if a tile cannot fit
bool src_periodic_shape(int P, int Q, int NL, int NR, int dpos_max, int G, int *NT_out, size_t *lds_bytes)
{
    if (P < 1 || P > 256) return false;
    const int NT = P * (256 / P);
    const long span = (long)(G * NT / P - 1) * Q + dpos_max + NL + NR + 1;
    if (span * 4 > 60 * 1024) return false;
    *NT_out = NT; *lds_bytes = (size_t)span * 4;
    return true;
}

hipError_t launch_src_periodic(const float *win, long win_stride, const double *Lc, const double *Rc, const int *dpos, const int *skipL,
                               const int *skipR, int P, int Q, int NL, int NR, int maxskipL, int maxskipR, int dpos_max, int pos0,
                               double scale, float *out, long out_stride, long nout, int nchan, hipStream_t s)
{
    if (nout <= 0 || nchan <= 0) return hipSuccess;
    const SrcPeriodic t = {Lc, Rc, dpos, skipL, skipR, P, Q, NL, NR, maxskipL, maxskipR, dpos_max};
    int NT = 0; size_t lds = 0;
    const int Gs[3] = {8, 4, 1};
    for (int i = 0; i < 3; ++i) {
        const int G = Gs[i];
        if (!src_periodic_shape(P, Q, NL, NR, dpos_max, G, &NT, &lds)) continue;
        if (G > 1 && nout < (long)G * NT) continue; // short epochs: less work per thread, more workgroups
        const dim3 grid((unsigned)((nout + (long)G * NT - 1) / ((long)G * NT)), (unsigned)nchan);
        if (G == 8) hipLaunchKernelGGL(src_sinc_periodic_kernel<8>, grid, dim3(256), lds, s, win, win_stride, t, pos0, NT, scale, out, out_stride, nout);
        else if (G == 4) hipLaunchKernelGGL(src_sinc_periodic_kernel<4>, grid, dim3(256), lds, s, win, win_stride, t, pos0, NT, scale, out, out_stride, nout);
        else hipLaunchKernelGGL(src_sinc_periodic_kernel<1>, grid, dim3(256), lds, s, win, win_stride, t, pos0, NT, scale, out, out_stride, nout);
        return hipGetLastError();
    }
    return hipErrorNotSupported;
}

// ---- the stream window of a single-launch call: [old buffer image | new input] by absolute index
struct SrcWindow {
    const float *old_img; long old_stride; // [nchan][old_stride]
    const float *input; long in_stride;    // [nchan][in_stride]
    long a_in0;                            // first absolute index served by `input`
};
__device__ __forceinline__ float win_load(const SrcWindow &w, int ch, long a)
{
    return a < w.a_in0 ? w.old_img[(long)ch * w.old_stride + a] : w.input[(long)ch * w.in_stride + (a - w.a_in0)];
}

// Stage a tile of the stream window into LDS: sample n of the tile goes to xs[n + P*(n / B)] (P pad
// floats after every B samples, P == 0: plain).  U independent loads per thread are issued before any of
// them is stored, the source select (old image / new input) is a pointer select so the loads carry no
// branch, and samples at or past `need` (they belong to no valid output and may not exist) read a
// clamped address and store zero.
template <int NT, int U, typename XT = float>
__device__ __forceinline__ void src_tile_load(XT *xs, const SrcWindow &w, int ch, long tile_base, int span, int need, int B, int P)
{
    const int tid = threadIdx.x;
    const float *old_row = w.old_img + (long)ch * w.old_stride;
    const float *in_row = w.input + (long)ch * w.in_stride - w.a_in0;
    int q = P ? tid / B : 0, r = P ? tid - q * B : 0; // n = q*B + r for this thread's next sample
    const int dq = NT / B, dr = NT - dq * B;
    for (int n0 = tid; n0 < span; n0 += NT * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            int n = n0 + u * NT;
            n = n < need ? n : need - 1;
            const long a = tile_base + n;
            const float *p = a < w.a_in0 ? old_row + a : in_row + a;
            v[u] = *p;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int n = n0 + u * NT;
            if (n < span) xs[n + P * q] = n < need ? (XT)v[u] : (XT)0;
            if (P) { q += dq; r += dr; if (r >= B) { r -= B; ++q; } }
        }
    }
}

// ---- general phase, LDS tile: any constant ratio (0.0213, 3.7 ...) without a table per phase ----------------------------------
// src_sinc_exact_kernel gives an output to one lane and lets it walk ~2 * half_len / increment taps with dependent loads from
// global memory: with a few thousand outputs per channel the launch has little parallelism and is bound by load latency (8 ms for
// 64 channels x 2^18 frames at ratio 0.0213).  Here a workgroup takes NT consecutive outputs of one channel, stages the span of
// the buffer image behind them in LDS, and gives the two wings of an output to two threads (they are independent sums that meet
// only in scale * (left + right)).  A wing is a counted loop (its tap count is known up front), walked U taps per step with the
// next step's table entries (global, L1 / L2-resident) and samples (LDS) requested before the current step's arithmetic.  Per tap
// exactly the expression of calc_output_single -- fraction from the low 12 bits, the float difference of two adjacent entries,
// icoeff in double, the product rounded, the sum in ascending order -- so the result is bit-identical to the per-lane kernel.
// Samples come through a SrcWindow: an epoch's buffer image (everything "old image"), or -- for a whole call in ONE launch -- the
// window [old image | new input] addressed by absolute index, with the positions the host's dry run of the library's control
// flow produced (try_general_window, src_host.hip).
template <int NT, int U>
__global__ __launch_bounds__(2 * NT) void src_sinc_tile_kernel(SrcWindow w, long a_limit, const float *__restrict__ coeffs,
                                                               int coeff_half_len, const int *__restrict__ pos, const int *__restrict__ start,
                                                               int increment, double scale, float *__restrict__ out, long out_stride, long nout)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    __shared__ double rsum[NT];
    const int tid = threadIdx.x, ch = blockIdx.y;
    const long k0 = (long)blockIdx.x * NT;
    const int nvalid = (int)((nout - k0 < NT) ? nout - k0 : NT);
    const int max_filter_index = coeff_half_len << SRC_SHIFT_BITS;
    const int cmax = max_filter_index / increment; // no wing has more than cmax + 1 taps
    const int lo = pos[k0] - cmax, hi = pos[k0 + nvalid - 1] + 1 + cmax; // positions are non-decreasing inside an epoch (checked by the host)
    for (int n = tid; n <= hi - lo; n += 2 * NT) {
        long a = (long)lo + n;
        a = a < 0 ? 0 : (a < a_limit ? a : a_limit - 1); // clamped indices are never read by a tap that exists
        xs[n] = win_load(w, ch, a);
    }
    __syncthreads();
    const bool right = tid >= NT; // wave-uniform
    const int o = right ? tid - NT : tid;
    double acc = 0.0;
    if (o < nvalid) {
        const int p = pos[k0 + o] - lo, st = start[k0 + o];
        int fi = right ? increment - st : st;
        const int cc = (max_filter_index - fi) / increment;
        fi += cc * increment;                     // the far end of the wing
        int di = right ? p + 1 + cc : p - cc;
        const int step = right ? -1 : 1;
        const int ntaps = cc + 1;
        const double inv_fp_one = 1.0 / (double)(1 << SRC_SHIFT_BITS);
        float c0[U], c1[U], xv[U];
        int fr[U];
        auto fetch = [&](int t) { // taps t .. t + U - 1 (clamped: a tap past the end is fetched again and not used)
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const int tt = t + j < ntaps ? t + j : ntaps - 1;
                const int f = fi - tt * increment;
                const int indx = f >> SRC_SHIFT_BITS;
                fr[j] = f & ((1 << SRC_SHIFT_BITS) - 1);
                c0[j] = coeffs[indx];
                c1[j] = coeffs[indx + 1];
                xv[j] = xs[di + tt * step];
            }
        };
        fetch(0);
        for (int t = 0; t < ntaps; t += U) {
            float k0v[U], k1v[U], xc[U];
            int fc[U];
#pragma unroll
            for (int j = 0; j < U; ++j) { k0v[j] = c0[j]; k1v[j] = c1[j]; xc[j] = xv[j]; fc[j] = fr[j]; }
            if (t + U < ntaps) fetch(t + U);
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (t + j < ntaps) {
                    const double fraction = (double)fc[j] * inv_fp_one;
                    const float dc = k1v[j] - k0v[j];
                    const double icoeff = (double)k0v[j] + fraction * (double)dc;
                    acc += icoeff * (double)xc[j];
                }
        }
    }
    if (right) rsum[o] = acc;
    __syncthreads();
    if (!right && o < nvalid) out[(long)ch * out_stride + k0 + o] = (float)(scale * (acc + rsum[o]));
}

// LDS floats the tile of outputs [k0, k0 + nt) needs, from the host's copy of the positions; 0 if some tile cannot fit
size_t src_tile_lds_bytes(const int *pos_host, long nout, int nt, int coeff_half_len, int increment)
{
    const int cmax = (coeff_half_len << SRC_SHIFT_BITS) / increment;
    long worst = 0;
    for (long k0 = 0; k0 < nout; k0 += nt) {
        const long k1 = k0 + nt < nout ? k0 + nt : nout;
        const long span = (long)pos_host[k1 - 1] - pos_host[k0] + 2 * (long)cmax + 2;
        worst = span > worst ? span : worst;
    }
    const size_t b = (size_t)worst * sizeof(float);
    return b <= 60 * 1024 ? b : 0; // two or three workgroups per CU
}

// old_img / input / a_in0: the window (an epoch launch passes its image as old_img, input = nullptr, a_in0 = a_limit = its length)
hipError_t launch_src_tile(const float *old_img, long old_stride, const float *input, long in_stride, long a_in0, long a_limit,
                           const float *coeffs, int coeff_half_len, const int *pos, const int *start,
                           int increment, double scale, float *out, long out_stride, long nout, int nchan, size_t lds_bytes, hipStream_t s)
{
    if (nout <= 0 || nchan <= 0) return hipSuccess;
    constexpr int NT = 128;
    const SrcWindow w = {old_img, old_stride, input ? input : old_img, input ? in_stride : old_stride, a_in0};
    auto kern = src_sinc_tile_kernel<NT, 4>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    dim3 grid((unsigned)((nout + NT - 1) / NT), (unsigned)nchan);
    hipLaunchKernelGGL(kern, grid, dim3(2 * NT), lds_bytes, s, w, a_limit, coeffs, coeff_half_len, pos, start, increment, scale, out, out_stride, nout);
    return hipGetLastError();
}

// The two wings of one output at zero phase: left = sum_t L[t]*x[t], right = sum_t R[t]*x[c-t], each a
// strictly ordered double sum (far end first, multiply and add rounded separately) exactly as
// calc_output_single runs them.  The wings are independent chains, so they run side by side, U taps
// per step; the LDS reads and scalar coefficient loads of step i+1 are issued before the arithmetic of
// step i (software pipeline: one lgkmcnt(0) per step with a full step of arithmetic to hide it).
// At zero phase R[t] == L[t] bit for bit for t <= cr when cr == cl - 1 (the right wing starts one
// increment in; prepare_uniform builds both from the same expression), so the shared part walks ONE
// coefficient stream and the double-buffered taps fit the SGPR file.
struct WingCursor { int lrem, lquo, rrem, rquo; };

template <int U, bool PAD, typename XT>
__device__ __forceinline__ void wings_fetch(const XT *xs, int lbase, int c, int S, const double *__restrict__ tab, int t,
                                            WingCursor &w, XT (&xl)[U], XT (&xr)[U], double (&k)[U])
{
    if (!PAD) { // plain layout: one base address per wing, the U reads differ by immediate offsets
        const XT *pl = xs + lbase + t, *pr = xs + lbase + (c - t - (U - 1));
#pragma unroll
        for (int j = 0; j < U; ++j) {
            xl[j] = pl[j];
            xr[j] = pr[U - 1 - j];
            k[j] = tab[t + j];
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
        xl[j] = xs[lbase + t + j + w.lquo];
        xr[j] = xs[lbase + (c - t - j) + w.rquo];
        k[j] = tab[t + j];
        if (++w.lrem == S) { w.lrem = 0; ++w.lquo; }
        if (--w.rrem < 0) { w.rrem += S; --w.rquo; }
    }
}

// float tiles: one pad float per S samples only when the lane stride S would put 8 or more lanes on a bank
// (S % 8 == 0); a 2- or 4-way ds_read_b32 conflict costs less than the f64 arithmetic it feeds.
// double tiles (ds_read_b64): an odd lane stride is conflict-free, so an even S gets one pad double per S samples.
__host__ __device__ __forceinline__ int src_tile_pad(int S, bool f64_tile = false) { return f64_tile ? ((S % 2 == 0) ? 1 : 0) : ((S % 8 == 0) ? 1 : 0); }

template <int U, bool PAD, typename XT = float>
__device__ __forceinline__ void sinc_wings(const XT *xs, int lbase, int S, const double *__restrict__ cl_rev,
                                           const double *__restrict__ cr_rev, int cl, int cr, double &left, double &right)
{
    constexpr int pad = PAD ? 1 : 0;
    const int c = cl + 1 + cr;
    WingCursor w;
    w.lrem = 0; w.lquo = 0;                  // left data index  t      = lquo*S + lrem
    w.rquo = c / S; w.rrem = c - w.rquo * S; // right data index c - t  = rquo*S + rrem
    const int both = cl < cr ? cl : cr;
    int t = 0;
    if (cr == cl - 1) {
        const int nsteps = (both + 1) / U;
        // two register sets used alternately (the loop body is the pipeline step written twice): the fetch of step i + 1 goes
        // into the set that step i - 1 has finished with, so nothing is copied between steps (a rotating single set costs
        // 32 v_mov_b32 per step: 5.1 vector instructions per tap instead of the 3 that do arithmetic)
        XT xlA[U], xrA[U], xlB[U], xrB[U];
        double kA[U], kB[U];
        auto land = [&](XT (&xl)[U], XT (&xr)[U], double (&k)[U]) { // the one lgkmcnt(0) of a step lands here, not after the next issue
#pragma unroll
            for (int j = 0; j < U; ++j) {
                asm volatile("" : "+v"(xl[j]), "+v"(xr[j]));
                asm volatile("" : "+s"(k[j]));
            }
        };
        auto compute = [&](const XT (&xl)[U], const XT (&xr)[U], const double (&k)[U]) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                left += k[j] * (double)xl[j];
                right += k[j] * (double)xr[j];
            }
        };
        if (nsteps > 0) { wings_fetch<U, PAD, XT>(xs, lbase, c, S, cl_rev, 0, w, xlA, xrA, kA); land(xlA, xrA, kA); }
        int i = 0;
        for (; i + 1 < nsteps; i += 2) {
            wings_fetch<U, PAD, XT>(xs, lbase, c, S, cl_rev, (i + 1) * U, w, xlB, xrB, kB);
            RD_SCHED_BARRIER();
            compute(xlA, xrA, kA);
            RD_SCHED_BARRIER();
            land(xlB, xrB, kB);
            if (i + 2 < nsteps) wings_fetch<U, PAD, XT>(xs, lbase, c, S, cl_rev, (i + 2) * U, w, xlA, xrA, kA);
            RD_SCHED_BARRIER();
            compute(xlB, xrB, kB);
            RD_SCHED_BARRIER();
            if (i + 2 < nsteps) land(xlA, xrA, kA);
        }
        if (i < nsteps) compute(xlA, xrA, kA); // an odd number of steps: the last one is already fetched
        t = nsteps * U;
    }
    const double *__restrict__ kr_tab = (cr == cl - 1) ? cl_rev : cr_rev;
    for (; t <= both; ++t) {
        const XT xl = xs[lbase + t + (pad ? w.lquo : 0)];
        const XT xr = xs[lbase + (c - t) + (pad ? w.rquo : 0)];
        left += cl_rev[t] * (double)xl;
        right += kr_tab[t] * (double)xr;
        if (++w.lrem == S) { w.lrem = 0; ++w.lquo; }
        if (--w.rrem < 0) { w.rrem += S; --w.rquo; }
    }
    for (; t <= cl; ++t) {
        left += cl_rev[t] * (double)xs[lbase + t + (pad ? w.lquo : 0)];
        if (++w.lrem == S) { w.lrem = 0; ++w.lquo; }
    }
    for (; t <= cr; ++t) {
        right += cr_rev[t] * (double)xs[lbase + (c - t) + (pad ? w.rquo : 0)];
        if (--w.rrem < 0) { w.rrem += S; --w.rquo; }
    }
}

// ---- uniform-phase fast path -------------------------------------------------------------------
// When 1/ratio is an integer S and the phase is zero (decimation by S, or ratio 1), every output of an
// epoch has start_filter_index 0 and the same increment, so the interpolated coefficients are the same
// for every output: c_left[i] = c(i*inc), c_right[i] = c((i+1)*inc), computed once on the host in
// double exactly as the per-tap expression would.  One wavefront then produces 64 consecutive outputs
// from an LDS tile of the 63*S + cl + cr + 2 input samples behind them (coalesced load, one pad float
// per S samples so the lane stride S+1 is odd -> conflict-free ds_read_b32), each lane running the two
// wings as two independent, strictly ordered double accumulations -- the library's own order, so the
// result is bit-identical to the general kernel and to the oracle.  Coefficients are wave-uniform and
// come through the scalar cache.
template <int NT> // NT threads = NT consecutive outputs per workgroup tile
__global__ __launch_bounds__(NT) void src_sinc_uniform_kernel(const float *__restrict__ win, long win_stride,
                                                              const double *__restrict__ cl_rev, int ncl, // far end first
                                                              const double *__restrict__ cr_rev, int ncr,
                                                              int pos0, int S, double scale, float *__restrict__ out,
                                                              long out_stride, long nout)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x;
    const long k0 = (long)blockIdx.x * NT;
    const int pad = src_tile_pad(S);
    const int cl = ncl - 1, cr = ncr - 1;
    const long tile_base = (long)pos0 + (long)S * k0 - cl; // buffer index of tile-relative sample 0
    const int span = (NT - 1) * S + cl + cr + 2;
    // the last tile may reach past the outputs that exist: clamp the load to what the valid outputs need
    const long nvalid = (nout - k0 < NT) ? nout - k0 : NT;
    const int need = (int)((nvalid - 1) * S) + cl + cr + 2;
    {
        const SrcWindow w = {win, win_stride, win, win_stride, 1L << 40}; // everything is 'old image': the live buffer
        src_tile_load<NT, 8>(xs, w, blockIdx.y, tile_base, span, need, S, pad);
    }
    __syncthreads();
    if (k0 + tid >= nout) return;
    const int lbase = (S + pad) * tid;
    double left = 0.0, right = 0.0;
    if (pad) sinc_wings<8, true>(xs, lbase, S, cl_rev, cr_rev, cl, cr, left, right);
    else sinc_wings<8, false>(xs, lbase, S, cl_rev, cr_rev, cl, cr, left, right);
    out[(long)blockIdx.y * out_stride + k0 + tid] = (float)(scale * (left + right));
}

template <int NT>
static hipError_t launch_uniform_t(const float *win, long win_stride, const double *cl_rev, int ncl, const double *cr_rev, int ncr,
                                   int pos0, int S, double scale, float *out, long out_stride, long nout, int nchan, size_t lds_bytes,
                                   hipStream_t s)
{
    auto kern = src_sinc_uniform_kernel<NT>;
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    dim3 grid((unsigned)((nout + NT - 1) / NT), (unsigned)nchan);
    hipLaunchKernelGGL(kern, grid, dim3(NT), lds_bytes, s, win, win_stride, cl_rev, ncl, cr_rev, ncr, pos0, S, scale, out, out_stride, nout);
    return hipGetLastError();
}

// LDS bytes of a tile of nt outputs; 0 if it cannot fit
size_t src_uniform_lds(int nt, int S, int cl, int cr)
{
    const long span = (long)(nt - 1) * S + cl + cr + 2;
    const size_t b = (size_t)(span + span / S + 2) * sizeof(float);
    return b <= 150 * 1024 ? b : 0;
}

hipError_t launch_src_uniform(const float *win, long win_stride, const double *cl_rev, int ncl, const double *cr_rev, int ncr,
                              int pos0, int S, double scale, float *out, long out_stride, long nout, int nchan, hipStream_t s)
{
    if (nout <= 0 || nchan <= 0) return hipSuccess;
    const int cl = ncl - 1, cr = ncr - 1;
    // widest tile that still lets two workgroups share a CU, else the widest that fits at all
    const int cand[3] = {256, 128, 64};
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < 3; ++i) {
            const size_t b = src_uniform_lds(cand[i], S, cl, cr);
            if (!b || (pass == 0 && b > 78 * 1024)) continue;
            switch (cand[i]) {
            case 256: return launch_uniform_t<256>(win, win_stride, cl_rev, ncl, cr_rev, ncr, pos0, S, scale, out, out_stride, nout, nchan, b, s);
            case 128: return launch_uniform_t<128>(win, win_stride, cl_rev, ncl, cr_rev, ncr, pos0, S, scale, out, out_stride, nout, nchan, b, s);
            default: return launch_uniform_t<64>(win, win_stride, cl_rev, ncl, cr_rev, ncr, pos0, S, scale, out, out_stride, nout, nchan, b, s);
            }
        }
    return hipErrorNotSupported;
}

// ---- single-launch uniform-phase path ------------------------------------------------------------
// The whole call as ONE launch: the stream window is [old buffer image | new input] addressed by an
// absolute index a (a < a_in0 -> old image, else input), output k sits at a0 + S*k.  Same arithmetic
// as src_sinc_uniform_kernel (bit-identical), no per-refill launches.
// XT is the element type of the LDS tile.  XT = double (each sample converted once while the tile is staged, no v_cvt_f64_f32 per
// tap) is bit-identical but was measured SLOWER at 1/50 -- the double tile of 256 outputs fills a CU's LDS: 12.9 ms with one
// thread per output (four waves per CU), 8.3 ms with the two wings of an output on two threads (eight waves), against 3.6 ms
// for the float tile at two workgroups per CU -- so every launch uses XT = float (profiles/r02_c3_experiments.txt).
template <int NT, typename XT = float>
__global__ __launch_bounds__(NT) void src_window_exact_kernel(SrcWindow w, const double *__restrict__ cl_rev, int ncl,
                                                              const double *__restrict__ cr_rev, int ncr, long a0, int S, double scale,
                                                              float *__restrict__ out, long out_stride, long nout)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    XT *xs = reinterpret_cast<XT *>(smem);
    const int tid = threadIdx.x, ch = blockIdx.y;
    const long k0 = (long)blockIdx.x * NT;
    const int pad = src_tile_pad(S, sizeof(XT) == 8);
    const int cl = ncl - 1, cr = ncr - 1;
    const long tile_base = a0 + (long)S * k0 - cl;
    const int span = (NT - 1) * S + cl + cr + 2;
    const long nvalid = (nout - k0 < NT) ? nout - k0 : NT;
    const int need = (int)((nvalid - 1) * S) + cl + cr + 2;
    src_tile_load<NT, 8, XT>(xs, w, ch, tile_base, span, need, S, pad);
    __syncthreads();
    if (k0 + tid >= nout) return;
    const int lbase = (S + pad) * tid;
    double left = 0.0, right = 0.0;
    if (pad) sinc_wings<8, true, XT>(xs, lbase, S, cl_rev, cr_rev, cl, cr, left, right);
    else sinc_wings<8, false, XT>(xs, lbase, S, cl_rev, cr_rev, cl, cr, left, right);
    out[(long)ch * out_stride + k0 + tid] = (float)(scale * (left + right));
}

// Tile loader of the register-blocked kernel: sample n of the tile goes to xs[n + P*(n / B)] (PAD) or xs[n].  The
// window [old image | new input] changes its source at ONE tile index, so the tile is three runs -- old image, new
// input, zeros behind `need` (samples that belong to no valid output and may not exist) -- each walked with a
// wave-uniform base pointer and a 32-bit lane index: per sample one address add, one coalesced load, one LDS store
// (src_tile_load pays a 64-bit pointer select, a clamp and a pad cursor per sample: 12 % of the kernel's vector
// instructions at 1/50).  U loads per thread are in flight before the first store.
template <int NT, int U, bool PAD>
__device__ __forceinline__ void src_rb_tile_load(float *xs, const SrcWindow &w, int ch, long tile_base, int span, int need, int B, int P)
{
    const int tid = threadIdx.x;
    const long ns = w.a_in0 - tile_base;
    const int nsplit = ns < 0 ? 0 : (ns < need ? (int)ns : need);
    const float *src[2] = {w.old_img + (long)ch * w.old_stride + tile_base, w.input + (long)ch * w.in_stride + (tile_base - w.a_in0)};
    const int lo[3] = {0, nsplit, need}, hi[3] = {nsplit, need, span};
#pragma unroll
    for (int part = 0; part < 3; ++part) {
        const int n1 = hi[part];
        int n = lo[part] + tid;
        int q = PAD ? n / B : 0, r = PAD ? n - q * B : 0; // n = q*B + r
        const int dq = PAD ? NT / B : 0, dr = PAD ? NT - dq * B : 0;
        const float *base = src[part < 2 ? part : 0];
        auto step_pad = [&]() { if (PAD) { q += dq; r += dr; if (r >= B) { r -= B; ++q; } } };
        for (; n + (U - 1) * NT < n1; n += NT * U) { // whole rounds: U loads in flight, no per-sample bound check
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = part < 2 ? base[n + u * NT] : 0.0f;
#pragma unroll
            for (int u = 0; u < U; ++u) { xs[n + u * NT + (PAD ? P * q : 0)] = v[u]; step_pad(); }
        }
        for (; n < n1; n += NT) { xs[n + (PAD ? P * q : 0)] = part < 2 ? base[n] : 0.0f; step_pad(); }
    }
}

// ---- register-blocked single-launch uniform-phase kernel (round 3) -------------------------------------------------
// src_window_exact_kernel gives a lane ONE output: every tap pays its own ds_read_b32 and its own v_cvt_f64_f32 although
// neighbouring outputs share all but S samples of their windows, and the lane stride S (50 floats) puts pairs of lanes
// on one bank.  Here a lane owns ONE WING of R consecutive outputs (src_core.h): one ds_read_b128 and four conversions
// feed 4*R taps.  The LDS holds about 730 outputs' worth of samples per CU whatever the lane assignment, so R outputs
// per lane would leave 1/R of the waves; the two wings of an output are independent sums (they meet only in
// scale * (left + right)), so they go to two lanes and the wave count stays: LW lanes per wing, tile of LW*R outputs,
// waves [0, LW/64) run left wings, the rest right wings, the right sums cross through LDS.  Bit-identical to
// src_window_exact_kernel and to oracle/oracle_src.c (tests/test_gpu_resample.py; lane program on the CPU:
// tests/test_emu_lane_programs.py).
template <int R, int U, int LW>
__global__ __launch_bounds__(2 * LW) void src_window_rb_kernel(SrcWindow w, const double *__restrict__ Lt, int ncl,
                                                               const double *__restrict__ Rt, int ncr, long a0, int S, double scale,
                                                               float *__restrict__ out, long out_stride, long nout)
{
    constexpr int NO = LW * R, NT = 2 * LW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *rsum = reinterpret_cast<double *>(smem);             // [NO] right-wing sums
    float *xs = reinterpret_cast<float *>(smem + NO * sizeof(double));
    const int tid = threadIdx.x, ch = blockIdx.y;
    const long k0 = (long)blockIdx.x * NO;
    const int cl = ncl - 1, cr = ncr - 1;
    const int B = R * S, P = src_rb_pad(B);
    const long tile_base = a0 + (long)S * k0 - cl;
    const int span = (NO - 1) * S + cl + cr + 2;
    const long nvalid = (nout - k0 < NO) ? nout - k0 : NO;
    const int need = (int)((nvalid - 1) * S) + cl + cr + 2;
    if (P) src_rb_tile_load<NT, 8, true>(xs, w, ch, tile_base, span, need, B, P);
    else src_rb_tile_load<NT, 8, false>(xs, w, ch, tile_base, span, need, B, P);
    __syncthreads();
    const bool right = tid >= LW; // wave-uniform
    const int q = right ? tid - LW : tid;
    const float *lane_xs = xs + q * (B + P);
    double acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.0;
    if ((long)(q & ~63) * R < nvalid) { // a wave whose outputs all lie past the end has nothing to do
        if (right) src_rb_wing<R, U, -1>(lane_xs, B, P, (R - 1) * S + cl + 1 + cr, Rt, ncr, S, acc);
        else src_rb_wing<R, U, +1>(lane_xs, B, P, 0, Lt, ncl, S, acc);
    }
    if (right) {
#pragma unroll
        for (int r = 0; r < R; ++r) rsum[q * R + (R - 1 - r)] = acc[r];
    }
    __syncthreads();
    if (!right) {
        float *o = out + (long)ch * out_stride + k0 + (long)q * R;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if ((long)q * R + r < nvalid) o[r] = (float)(scale * (acc[r] + rsum[q * R + r]));
    }
}

// which register blocking serves (S, taps): 0 = none (the one-output-per-lane kernel runs)
int src_rb_choose(int S, int ncl, int ncr)
{
    const int forced = measure_env("REDIO_SRC_RB") ? atoi(measure_env("REDIO_SRC_RB")) : -1; // measurement only: 0 = off, 2 / 4 = that blocking
    if (forced == 0) return 0;
    const int nmin = ncl < ncr ? ncl : ncr;
    const int cand[2] = {forced == 4 ? 4 : 2, forced == 2 ? 2 : 4};
    for (int i = 0; i < 2; ++i) {
        const int R = cand[i];
        if ((R * S) % 4 != 0) continue;
        if ((R - 1) * S + 24 > nmin) continue; // the ramps must be short next to the wing
        if ((size_t)src_rb_tile_floats(256, R, S, ncl - 1, ncr - 1) * sizeof(float) + 256 * sizeof(double) > 78 * 1024) continue; // two tiles per CU
        return R;
    }
    return 0;
}

typedef float src_v2f __attribute__((ext_vector_type(2))); // one 64-bit register pair (v_pk_fma_f32 operand)

// ---- f32 polyphase decimator (REDIO_SRC_FAST): the same uniform-phase filter as ONE real FIR
// H[j] = (float)(scale * icoeff_j), j = 0 .. KH-1 (left wing far end first, then the right wing near
// end first), evaluated with f32 FMAs.  A lane owns two consecutive outputs (2l, 2l+1) that share every
// LDS read: acc.xy += (x, x) * (H[m], H[m-S]) -- one v_pk_fma_f32 per sample with the tap pair as a
// wave-uniform SGPR operand from the packed table T2[m] = (H[m], H[m-S]), m = 0 .. nm-1 (nm = KH + S
// rounded up to a multiple of 128, zero filled).  The eight waves of a workgroup split the tap range
// (K-split) over one shared 128-output tile and are summed in wave order.  Eight taps per step, the
// next step's LDS reads and scalar loads in flight under the current step's FMAs.
// LDS layout: lane stride 2S floats; ds_read_b64 is conflict-free when (stride/2) is odd, so an even S
// gets two pad floats per 2S samples (PAD), an odd S none.
// Tolerance vs the exact path: |d| <= (KH + 1) * 2^-24 * sum|H| * max|x| (tested).
enum { SRC_FAST_WAVES = 8, SRC_FAST_STEP = 8 }; // K-split width and taps per pipeline step; nm % (WAVES*STEP) == 0

template <bool PAD>
__global__ __launch_bounds__(64 * SRC_FAST_WAVES) void src_window_fast_kernel(SrcWindow w, const float2 *__restrict__ T2, int nm, int KH,
                                                                              int cl, long a0, int S, float *__restrict__ out,
                                                                              long out_stride, long nout)
{
    constexpr int NW = SRC_FAST_WAVES, ST = SRC_FAST_STEP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    const int tid = threadIdx.x, ch = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const long k0 = (long)blockIdx.x * 128;
    const long tile_base = a0 + (long)S * k0 - cl;
    const int S2 = 2 * S;
    const int span = 126 * S + nm;
    const long nvalid = (nout - k0 < 128) ? nout - k0 : 128;
    const int need = (int)((nvalid - 1) * S) + KH; // samples past this belong to no valid output (and may not exist)
    src_tile_load<64 * NW, 8>(xs, w, ch, tile_base, span, need, S2, PAD ? 2 : 0);
    float2 *red = reinterpret_cast<float2 *>(xs + ((span + (PAD ? 2 * (span / S2) : 0) + 5) & ~1)); // [NW][64] partial sums
    __syncthreads();
    const int per = nm / NW; // multiple of ST
    const int m0 = wave * per, nsteps = per / ST;
    const int lbase = (S2 + (PAD ? 2 : 0)) * lane;
    int quo = PAD ? m0 / S2 : 0, rem = PAD ? m0 - quo * S2 : 0;
    src_v2f acc = {0.f, 0.f};
    src_v2f x[ST / 2], t[ST];
    auto fetch = [&](int m, src_v2f(&fx)[ST / 2], src_v2f(&ft)[ST]) {
#pragma unroll
        for (int h = 0; h < ST / 4; ++h) { // four samples never straddle a pad (S2 % 4 == 0 when PAD)
            const float *p = xs + lbase + m + 4 * h + 2 * quo;
            fx[2 * h] = *reinterpret_cast<const src_v2f *>(p);
            fx[2 * h + 1] = *reinterpret_cast<const src_v2f *>(p + 2);
            if (PAD) { rem += 4; if (rem >= S2) { rem -= S2; ++quo; } }
        }
#pragma unroll
        for (int j = 0; j < ST; ++j) { const float2 v = T2[m + j]; ft[j] = src_v2f{v.x, v.y}; }
    };
    if (nsteps > 0) fetch(m0, x, t);
#pragma unroll
    for (int j = 0; j < ST / 2; ++j) asm volatile("" : "+v"(x[j]));
#pragma unroll
    for (int j = 0; j < ST; ++j) asm volatile("" : "+s"(t[j]));
    for (int i = 0; i < nsteps; ++i) {
        src_v2f nx[ST / 2], nt[ST];
        if (i + 1 < nsteps) fetch(m0 + ST * (i + 1), nx, nt);
        RD_SCHED_BARRIER();
#pragma unroll
        for (int j = 0; j < ST / 2; ++j) {
            acc = __builtin_elementwise_fma(src_v2f{x[j].x, x[j].x}, t[2 * j], acc);
            acc = __builtin_elementwise_fma(src_v2f{x[j].y, x[j].y}, t[2 * j + 1], acc);
        }
        RD_SCHED_BARRIER();
#pragma unroll
        for (int j = 0; j < ST / 2; ++j) { asm volatile("" : "+v"(nx[j])); x[j] = nx[j]; }
#pragma unroll
        for (int j = 0; j < ST; ++j) { asm volatile("" : "+s"(nt[j])); t[j] = nt[j]; }
    }
    red[wave * 64 + lane] = make_float2(acc.x, acc.y);
    __syncthreads();
    if (wave == 0) {
        float2 s = red[lane];
        for (int q = 1; q < NW; ++q) { s.x = add_rn(s.x, red[q * 64 + lane].x); s.y = add_rn(s.y, red[q * 64 + lane].y); }
        const long k = k0 + 2 * lane;
        if (k < nout) out[(long)ch * out_stride + k] = s.x;
        if (k + 1 < nout) out[(long)ch * out_stride + k + 1] = s.y;
    }
}

// ---- f32 polyphase decimator, phase-split (round 3; REDIO_SRC_FAST, tolerance-tested, NOT bit-identical) ------------
// out[o] = sum_m H[m] * x[S*o + m] with m = S*j + p is, for every phase p, a NON-decimating FIR over the phase's own
// sample stream X_p[n] = x[S*n + p] with the taps H_p[j] = H[S*j + p] (about 92 of them at any S: the filter stretches
// with S).  The sum over the S phases is order-free here, so the K-split goes over PHASES: the W wavefronts of a
// workgroup share one tile of 512 outputs and take the phases p = w, w + W, ... each; no wavefront has a ramp, none
// reads a tap that is not its own.  A lane owns R = 8 consecutive outputs.  Per phase it reads the 8 + taps samples its
// outputs see into registers ONCE and runs the taps as v_pk_fma_f32 with the tap as a wave-uniform SGPR operand
// broadcast to both halves: one scalar register feeds 8 multiply-adds (the tap-range K-split of
// src_window_fast_kernel needs a scalar register PAIR per packed multiply-add and is bound by scalar-load latency).
// Packed operands must be even-aligned register pairs E[a] = (X[2a], X[2a+1]): even taps run on the output pairs
// (0,1) .. (6,7) (accA), odd taps on the pairs (-1,0) .. (7,8) (accB, whose two outer halves are discarded): 9 packed
// instructions per 16 multiply-adds.
// A workgroup is persistent over TPW consecutive tiles of one channel: the NEXT tile's samples are requested into registers
// (PF per thread) before the current tile's arithmetic and stored to the LDS image after it, so the HBM time of a tile --
// 120 KB at the 10 B/clk a CU gets, as long as half the arithmetic -- runs under the arithmetic of the tile before it (the
// image fills the LDS: a second workgroup per CU cannot provide that overlap).  Eight wavefronts = two per SIMD.
// LDS image: tile sample n = (8*q + k)*S + p (k < 8) sits in float cell 2*(((k/2)*S + p)*NCOL + q) + k%2: the samples
// X_p[8*q + i], X_p[8*q + i + 1] (i even) a lane multiplies as one packed operand are ONE aligned 8-byte cell, column
// q + i/8 of row (i%8/2)*S + p -- four row bases per phase, the column step an immediate, consecutive lanes on consecutive
// bank pairs: conflict-free ds_read_b64, no pad, no per-read address arithmetic.
template <int NPAIR> // tap pairs per phase (the table rows are zero filled to whole chunks of 16 pairs)
struct SrcFastP {
    static constexpr int R = 8, W = 8, PF = 64, NC = (NPAIR + 15) / 16, NTAP = 32 * NC, NE = NPAIR + R / 2, NI = 2 * NE; // NE packed pairs = NI window samples per lane and phase
    static constexpr int NCOL = 64 + (NI + R - 1) / R;                                                // columns of the image
    static constexpr int NGROUPS = 63 * R + NI;                                                       // groups of S samples (n = g*S + p, g = 8*q + k) a lane can read from
    static size_t lds_bytes(int S) { return ((size_t)R * S * NCOL + (size_t)W * 64 * R + 4) * sizeof(float); } // image, partial sums, a spare cell
    static bool fits(int S) { return S >= 2 && S <= 64 * W && (NGROUPS + (64 * W) / S - 1) / ((64 * W) / S) <= PF && lds_bytes(S) <= 160 * 1024; }
};

template <int NPAIR>
__global__ __launch_bounds__(64 * SrcFastP<NPAIR>::W) void src_window_fastp_kernel(SrcWindow w, const float *__restrict__ Hp, int KH, int cl, long a0, int S,
                                                                               float *__restrict__ out, long out_stride, long nout, int TPW)
{
    using G = SrcFastP<NPAIR>;
    constexpr int R = G::R, W = G::W, PF = G::PF, NC = G::NC, NO = 64 * R, NT = 64 * W, NTAP = G::NTAP, NE = G::NE, NCOL = G::NCOL, NGROUPS = G::NGROUPS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);   // [4*S rows][NCOL] cells of two floats
    const int tid = threadIdx.x, ch = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    float *red = xs + R * S * NCOL;                // [W][NO] partial sums
    const src_v2f *xs2 = reinterpret_cast<const src_v2f *>(xs);
    const long ntiles = (nout + NO - 1) / NO;
    const long t0 = (long)blockIdx.x * TPW;
    const int ntile = (int)(ntiles - t0 < TPW ? ntiles - t0 : TPW);
    // loader mapping: a thread keeps ONE phase.  With GI = NT / S groups per round, thread t < GI*S holds sample pl = t % S of
    // group g = t / S + GI*round, round < PF: n and the cell advance by constants, (q, k) = (g / 8, g % 8) by (GI / 8, GI % 8)
    const int GI = NT / S, pl = tid % S, g0 = tid / S;
    const bool active = tid < GI * S;
    float pf[PF];
    // Requests go through buffer descriptors whose range check returns 0.0f for everything at or behind `need` (samples that
    // belong to no valid output and may not exist), for the other source of a tile that straddles [old image | new input], and
    // for the rounds past the image: no per-sample branch, one 32-bit offset per request.
    auto prefetch = [&](long tile) {
        const long k0 = tile * NO, tile_base = a0 + (long)S * k0 - cl;
        const long nvalid = (nout - k0 < NO) ? nout - k0 : NO;
        const int need = (int)((nvalid - 1) * S) + KH;
        const long ns = w.a_in0 - tile_base;
        const int nsplit = ns < 0 ? 0 : (ns < need ? (int)ns : need); // [0, nsplit): old image, [nsplit, need): new input
        const float *src_old = w.old_img + (long)ch * w.old_stride + tile_base, *src_new = w.input + (long)ch * w.in_stride + (tile_base - w.a_in0) + nsplit;
        const auto r_old = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src_old), 0, nsplit * 4, 0x00020000);
        const auto r_new = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(src_new), 0, (need - nsplit) * 4, 0x00020000);
        unsigned off0 = active ? 4u * (unsigned)(g0 * S + pl) : 0x80000000u;
        const unsigned doff = 4u * (unsigned)(GI * S), split4 = 4u * (unsigned)nsplit;
        int gl = g0;
        asm volatile("" : "+v"(off0), "+v"(gl)); // per tile: the 64 request offsets are recomputed from this one, not kept alive across the tile loop
        if (nsplit == 0 || nsplit >= need) { // ONE source
            // round u reads doff*u bytes further on: the advance goes into the descriptor (scalar arithmetic: base + doff*u, range - doff*u),
            // every round uses the lane's one offset register -- no vector instruction per request.  A round past the image reads
            // behind `need` (+0.0f) or a sample nobody uses; its store goes to the spare cell either way.
            const char *srcp = reinterpret_cast<const char *>(nsplit == 0 ? src_new : src_old);
            const int range = 4 * need;
            int dstep = (int)doff;
            asm volatile("" : "+s"(dstep)); // per tile: the 64 advances are recomputed, not kept in 64 scalar registers across the tile loop
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                const int adv = dstep * u, rem = range - adv;
                const auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(srcp + adv), 0, rem > 0 ? rem : 0, 0x00020000);
                pf[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off0, 0, 0));
                RD_SCHED_BARRIER(); // a descriptor is built where it is used (hoisted together the 64 of them spill)
            }
        } else { // the one tile of a call that straddles the two sources: both descriptors, eight samples at a time
#pragma unroll
            for (int u0 = 0; u0 < PF; u0 += 8) {
                unsigned a_[8], b_[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const unsigned off = gl + GI * (u0 + u) < NGROUPS ? off0 + doff * (u0 + u) : 0x80000000u;
                    a_[u] = __builtin_amdgcn_raw_buffer_load_b32(r_old, off, 0, 0);
                    b_[u] = __builtin_amdgcn_raw_buffer_load_b32(r_new, off - split4, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) pf[u0 + u] = __builtin_bit_cast(float, a_[u] | b_[u]); // one of the two is out of its range: +0.0f
                RD_SCHED_BARRIER();
            }
        }
    };
    auto store_image = [&]() { // the requested samples -> the LDS image (waits for them)
        if (!active) return;
        // group g = 8*q + k -> byte address 8*(((k/2)*S + pl)*NCOL + q) + 4*(k%2).  g advances by GI per round, so k returns after
        // eight rounds with q advanced by GI: eight addresses from g alone (24-bit multiply-add, shifts), then one add per store;
        // a round past the image stores into the spare cell behind the partial sums
        unsigned gg = (unsigned)g0;
        asm volatile("" : "+v"(gg)); // per tile: the addresses are recomputed from this, not kept alive across the tile loop
        const unsigned rowbytes = 8u * (unsigned)(S * NCOL), base = 8u * (unsigned)(pl * NCOL), spare = (unsigned)((R * S * NCOL + W * NO) * sizeof(float));
        unsigned a8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned g = gg + (unsigned)(GI * u);
            a8[u] = __umul24((g >> 1) & 3u, rowbytes) + base + ((g >> 3) << 3) + ((g & 1u) << 2);
        }
        char *lds = smem;
        const unsigned step = 8u * (unsigned)GI, glimit = (unsigned)NGROUPS - gg; // store u is inside the image iff GI*u < glimit
        // rounds u0 .. u0+7 are inside the image for every active lane (gg < GI) if GI*(u0+8) <= NGROUPS: a wave-uniform test per eight
        // stores, and one vector add per store; only the last rounds pay the per-lane test and the select
#pragma unroll
        for (int u0 = 0; u0 < PF; u0 += 8) {
            if (GI * (u0 + 8) <= NGROUPS) {
#pragma unroll
                for (int u = u0; u < u0 + 8; ++u) *reinterpret_cast<float *>(lds + (a8[u % 8] + step * (unsigned)(u / 8))) = pf[u];
            } else {
#pragma unroll
                for (int u = u0; u < u0 + 8; ++u) {
                    const unsigned addr = (unsigned)(GI * u) < glimit ? a8[u % 8] + step * (unsigned)(u / 8) : spare;
                    *reinterpret_cast<float *>(lds + addr) = pf[u];
                }
            }
        }
    };
    if (ntile > 0) { prefetch(t0); store_image(); }
    __syncthreads();
    for (int ti = 0; ti < ntile; ++ti) {
        const long k0 = (t0 + ti) * NO;
        const long nvalid = (nout - k0 < NO) ? nout - k0 : NO;
        if (ti + 1 < ntile) prefetch(t0 + ti + 1); // in flight during this tile's arithmetic
        src_v2f accA[R / 2], accB[R / 2 + 1];
#pragma unroll
        for (int c = 0; c < R / 2; ++c) accA[c] = src_v2f{0.f, 0.f};
#pragma unroll
        for (int c = 0; c <= R / 2; ++c) accB[c] = src_v2f{0.f, 0.f};
        // One software pipeline over all (phase, chunk of 32 taps) steps of the wavefront: the next step's 16 tap pairs are
        // requested (scalar loads) before the current step's 144 packed multiply-adds and land after them; a window register is
        // dead once the tap pair with its own index has run, and is refilled on the spot with the NEXT phase's sample pair, so
        // the LDS reads of a phase travel under the arithmetic of the phase before it at no cost in registers.
        if (wave < S) {
            src_v2f E[NE], hc[16];
            {
                const src_v2f *hp = reinterpret_cast<const src_v2f *>(Hp + (long)wave * NTAP);
#pragma unroll
                for (int m = 0; m < 16; ++m) hc[m] = hp[m];
#pragma unroll
                for (int a = 0; a < NE; ++a) E[a] = xs2[((a % (R / 2)) * S + wave) * NCOL + lane + a / (R / 2)];
#pragma unroll
                for (int m = 0; m < 16; ++m) asm volatile("" : "+s"(hc[m]));
            }
            for (int p = wave; p < S; p += W) {
                const int pn = p + W < S ? p + W : p; // the phase after this one (the last refills with itself: never used)
                const src_v2f *rown[R / 2];           // (X_pn[8*q + i], X_pn[8*q + i + 1]) = rown[i % 8 / 2][i / 8], i even
#pragma unroll
                for (int kk = 0; kk < R / 2; ++kk) rown[kk] = xs2 + (kk * S + pn) * NCOL + lane;
                const src_v2f *hp = reinterpret_cast<const src_v2f *>(Hp + (long)p * NTAP), *hpn = reinterpret_cast<const src_v2f *>(Hp + (long)pn * NTAP);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    src_v2f hn[16];
                    const src_v2f *nextc = c + 1 < NC ? hp + 16 * (c + 1) : hpn;
#pragma unroll
                    for (int m = 0; m < 16; ++m) hn[m] = nextc[m];
                    RD_SCHED_BARRIER();
#pragma unroll
                    for (int m = 0; m < 16; ++m) {
                        const int a0i = 16 * c + m; // index of the tap pair (2*a0i, 2*a0i + 1) and of the window pair its first output pair reads
                        if (a0i >= NPAIR) continue; // zero fill of the table row
                        const src_v2f he = __builtin_shufflevector(hc[m], hc[m], 0, 0), ho = __builtin_shufflevector(hc[m], hc[m], 1, 1);
#pragma unroll
                        for (int ca = 0; ca < R / 2; ++ca) accA[ca] = __builtin_elementwise_fma(E[ca + a0i], he, accA[ca]);
#pragma unroll
                        for (int cb = 0; cb <= R / 2; ++cb) accB[cb] = __builtin_elementwise_fma(E[cb + a0i], ho, accB[cb]);
                        E[a0i] = rown[a0i % (R / 2)][a0i / (R / 2)];
                        if (a0i == NPAIR - 1) {
#pragma unroll
                            for (int a = NPAIR; a < NE; ++a) E[a] = rown[a % (R / 2)][a / (R / 2)];
                        }
                        if (m % 4 == 3) RD_SCHED_BARRIER();
                    }
#pragma unroll
                    for (int m = 0; m < 16; ++m) { asm volatile("" : "+s"(hn[m])); hc[m] = hn[m]; }
                }
            }
        }
        float *myred = red + wave * NO + lane * R;
#pragma unroll
        for (int c = 0; c < R / 2; ++c) {
            myred[2 * c] = accA[c].x + accB[c].y;
            myred[2 * c + 1] = accA[c].y + accB[c + 1].x;
        }
        __syncthreads(); // every wavefront is done with this tile's image; the partial sums are complete
        for (int t = tid; t < NO; t += NT) {
            float sum = red[t];
            for (int q = 1; q < W; ++q) sum = add_rn(sum, red[q * NO + t]);
            if (t < nvalid) out[(long)ch * out_stride + k0 + t] = sum;
        }
        if (ti + 1 < ntile) store_image();
        __syncthreads(); // the next tile's image is in place; the partial sums are free
    }
}

#include "src_fastp2.h" // round 5: the same kernel with the loader inside the multiply-add stream (two image halves by phase)

// tap pairs per phase the f32 polyphase phase-split kernel is built for (>= the shape's); 0: the kernel does not serve the shape
#define REDIO_FASTP_SHAPES(X) X(16) X(20) X(24) X(32) X(40) X(46) X(48)
int src_fastp_pairs(int S, int KH)
{
    if (measure_env("REDIO_SRC_FASTP") && !atoi(measure_env("REDIO_SRC_FASTP"))) return 0; // measurement only
    const int npair = ((KH + S - 1) / S + 1) / 2;
#define X(N) if (npair <= N) return SrcFastP<N>::fits(S) ? N : 0;
    REDIO_FASTP_SHAPES(X)
#undef X
    return 0;
}
int src_fastp_row(int npair) { return 32 * ((npair + 15) / 16); } // floats per phase row of the tap table

// rebuild the library's buffer image after a single-launch call: dst[j] = window(A0 + j), j in [j0, j1)
__global__ __launch_bounds__(256) void src_window_image_kernel(SrcWindow w, long A0, long j0, long j1, float *__restrict__ dst, long dst_stride)
{
    const long j = j0 + (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < j1) dst[(long)blockIdx.y * dst_stride + j] = win_load(w, blockIdx.y, A0 + j);
}

hipError_t launch_src_window_image(const float *old_img, long old_stride, const float *input, long in_stride, long a_in0, long A0f, long j0, long j1,
                                   float *new_img, int nchan, hipStream_t s)
{
    if (j1 <= j0 || nchan <= 0) return hipSuccess;
    const SrcWindow w = {old_img, old_stride, input, in_stride, a_in0};
    dim3 grid((unsigned)((j1 - j0 + 255) / 256), (unsigned)nchan);
    hipLaunchKernelGGL(src_window_image_kernel, grid, dim3(256), 0, s, w, A0f, j0, j1, new_img, old_stride);
    return hipGetLastError();
}

hipError_t launch_src_window(const float *old_img, long old_stride, const float *input, long in_stride, long a_in0,
                             const double *cl_rev, int ncl, const double *cr_rev, int ncr, const float2 *T2, int nm, const float *Hp, int fastp_nc,
                             bool fast, long a0, int S, double scale, float *out, long out_stride, long nout, int nchan,
                             long A0f, long j0, long j1, float *new_img, hipStream_t s)
{
    SrcWindow w = {old_img, old_stride, input, in_stride, a_in0};
    const int cl = ncl - 1, cr = ncr - 1;
    if (nout > 0) {
        if (fast && Hp && fastp_nc > 0) {
            const int KH = ncl + ncr;
            const long ntiles = (nout + 511) / 512;
            bool done = false;
            // the round-5 kernel requests everything behind a workgroup's first tile from the NEW input only (src_fastp2.h): the call's second tile must
            // not reach into the old image.  The image holds about two filter half-lengths of a call's window, a tile is 512 x S samples: true for every
            // state the converter can be in at these S -- checked, not assumed; a call that fails the check runs round 3's kernel.
            const bool later_tiles_in_new_input = w.a_in0 <= a0 - cl + 512L * S;
            // round 5: the same arithmetic with the loader inside the multiply-add stream (src_fastp2.h) for the tap counts of the medium and
            // fastest converters at S >= 25; persistent workgroups, about one per CU in all (its prologue loads a tile with nothing to overlap)
#define LAUNCH_FP2(N)                                                                                                             \
    if (!done && fastp_nc == N && SrcFastP2<N, 32>::fits(S) && later_tiles_in_new_input && !measure_env("REDIO_SRC_FASTP_R3")) {                              \
        auto kern = src_window_fastp2_kernel<N, 32, 0>;                                                                           \
        const size_t lds = SrcFastP2<N, 32>::lds_bytes(S);                                                                        \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                            \
        long splits = 256 / nchan;                                                                                                \
        splits = splits < 1 ? 1 : (splits > ntiles ? ntiles : splits);                                                            \
        const long tpw = (ntiles + splits - 1) / splits;                                                                          \
        dim3 grid((unsigned)((ntiles + tpw - 1) / tpw), (unsigned)nchan);                                                         \
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, w, Hp, KH, cl, a0, S, out, out_stride, nout, (int)tpw);                 \
        done = true;                                                                                                              \
    }
            LAUNCH_FP2(20) LAUNCH_FP2(24) LAUNCH_FP2(46) LAUNCH_FP2(48)
#undef LAUNCH_FP2
            if (!done) {
            // tiles per (persistent) workgroup: a channel's tiles in equal runs, about two workgroups per CU in all (one is resident)
            long splits = 512 / nchan;
            splits = splits < 1 ? 1 : (splits > ntiles ? ntiles : splits);
            const long tpw = (ntiles + splits - 1) / splits;
            dim3 grid((unsigned)((ntiles + tpw - 1) / tpw), (unsigned)nchan);
#define LAUNCH_FP(N)                                                                                                              \
    if (fastp_nc == N) {                                                                                                          \
        auto kern = src_window_fastp_kernel<N>;                                                                                   \
        const size_t lds = SrcFastP<N>::lds_bytes(S);                                                                             \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        if (e != hipSuccess) return e;                                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(64 * SrcFastP<N>::W), lds, s, w, Hp, KH, cl, a0, S, out, out_stride, nout, (int)tpw); \
    }
            REDIO_FASTP_SHAPES(LAUNCH_FP)
#undef LAUNCH_FP
            }
        } else if (fast) {
            const int KH = ncl + ncr;
            const bool padded = (S % 2) == 0;
            const long span = 126L * S + nm;
            const size_t lds = (size_t)(span + (padded ? 2 * (span / (2 * S)) : 0) + 8) * sizeof(float) + SRC_FAST_WAVES * 64 * sizeof(float2);
            if (lds > 150 * 1024 || nm % (SRC_FAST_WAVES * SRC_FAST_STEP) != 0) return hipErrorNotSupported;
            dim3 grid((unsigned)((nout + 127) / 128), (unsigned)nchan);
#define LAUNCH_F(P)                                                                                                               \
    {                                                                                                                             \
        auto kern = src_window_fast_kernel<P>;                                                                                    \
        if (lds > 48 * 1024) {                                                                                                    \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            if (e != hipSuccess) return e;                                                                                        \
        }                                                                                                                         \
        hipLaunchKernelGGL(kern, grid, dim3(64 * SRC_FAST_WAVES), lds, s, w, T2, nm, KH, cl, a0, S, out, out_stride, nout);                       \
    }
            if (padded) LAUNCH_F(true) else LAUNCH_F(false)
#undef LAUNCH_F
        } else if (const int RB = src_rb_choose(S, ncl, ncr)) {
            const bool wide = RB == 2 && measure_env("REDIO_SRC_RB_WIDE") && atoi(measure_env("REDIO_SRC_RB_WIDE")) &&
                              (size_t)src_rb_tile_floats(640, 2, S, cl, cr) * sizeof(float) + 640 * sizeof(double) <= 160 * 1024; // measurement only
            const int NOt = wide ? 640 : 256;
            const size_t b = (size_t)src_rb_tile_floats(NOt, RB, S, cl, cr) * sizeof(float) + NOt * sizeof(double);
            dim3 grid((unsigned)((nout + NOt - 1) / NOt), (unsigned)nchan);
#define LAUNCH_RB(R, U, LW)                                                                                                       \
    {                                                                                                                             \
        auto kern = src_window_rb_kernel<R, U, LW>;                                                                               \
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b); \
        if (e != hipSuccess) return e;                                                                                            \
        hipLaunchKernelGGL(kern, grid, dim3(2 * LW), b, s, w, cl_rev, ncl, cr_rev, ncr, a0, S, scale, out, out_stride, nout);     \
    }
            if (RB == 2 && wide) LAUNCH_RB(2, 8, 320) else if (RB == 2) LAUNCH_RB(2, 8, 128) else LAUNCH_RB(4, 4, 64)
#undef LAUNCH_RB
        } else {
            const int cand[3] = {256, 128, 64};
            bool done = false;
            for (int pass = 0; pass < 2 && !done; ++pass)
                for (int i = 0; i < 3 && !done; ++i) {
                    const size_t b = src_uniform_lds(cand[i], S, cl, cr);
                    if (!b || (pass == 0 && b > 78 * 1024)) continue;
                    dim3 grid((unsigned)((nout + cand[i] - 1) / cand[i]), (unsigned)nchan);
#define LAUNCH_W(NT)                                                                                                              \
    {                                                                                                                             \
        auto kern = src_window_exact_kernel<NT>;                                                                                  \
        if (b > 48 * 1024) {                                                                                                      \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)b); \
            if (e != hipSuccess) return e;                                                                                        \
        }                                                                                                                         \
        hipLaunchKernelGGL(kern, grid, dim3(NT), b, s, w, cl_rev, ncl, cr_rev, ncr, a0, S, scale, out, out_stride, nout);         \
    }
                    if (cand[i] == 256) LAUNCH_W(256) else if (cand[i] == 128) LAUNCH_W(128) else LAUNCH_W(64)
#undef LAUNCH_W
                    done = true;
                }
            if (!done) return hipErrorNotSupported;
        }
    }
    if (j1 > j0) {
        dim3 grid((unsigned)((j1 - j0 + 255) / 256), (unsigned)nchan);
        hipLaunchKernelGGL(src_window_image_kernel, grid, dim3(256), 0, s, w, A0f, j0, j1, new_img, old_stride);
    }
    return hipGetLastError();
}

// window maintenance: dst[c][0..keep) = src[c][from..from+keep) (overlapping allowed: goes through
// registers in ascending order per thread block stride, keep <= from is NOT assumed -> two-buffer use)
__global__ __launch_bounds__(256) void src_copy_rows_kernel(const float *__restrict__ src, long src_stride, long src_off,
                                                            float *__restrict__ dst, long dst_stride, long dst_off, long n)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dst[(long)blockIdx.y * dst_stride + dst_off + i] = src[(long)blockIdx.y * src_stride + src_off + i];
}

hipError_t launch_src_copy_rows(const float *src, long src_stride, long src_off, float *dst, long dst_stride, long dst_off,
                                long n, int nchan, hipStream_t s)
{
    if (n <= 0 || nchan <= 0) return hipSuccess;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)nchan);
    hipLaunchKernelGGL(src_copy_rows_kernel, grid, dim3(256), 0, s, src, src_stride, src_off, dst, dst_stride, dst_off, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void src_fill_rows_kernel(float *dst, long dst_stride, long dst_off, long n, float v)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dst[(long)blockIdx.y * dst_stride + dst_off + i] = v;
}

hipError_t launch_src_fill_rows(float *dst, long dst_stride, long dst_off, long n, int nchan, float v, hipStream_t s)
{
    if (n <= 0 || nchan <= 0) return hipSuccess;
    dim3 grid((unsigned)((n + 255) / 256), (unsigned)nchan);
    hipLaunchKernelGGL(src_fill_rows_kernel, grid, dim3(256), 0, s, dst, dst_stride, dst_off, n, v);
    return hipGetLastError();
}

// ---- converters 3 / 4: zero-order hold and linear (src_zoh.c / src_linear.c of the published library) -----------
// The host runs the library's double recurrence and hands over, per output, the index i of the sample before the
// output instant (-1: the value carried from the previous call) and the fractional position; all channels share them.
// out = a (hold) or (float)(a + frac * (b - a)) with the difference b - a formed in float, as the C expression does.
__global__ __launch_bounds__(256) void src_zoh_linear_kernel(const float *__restrict__ in, long in_stride, const float *__restrict__ last,
                                                             const int *__restrict__ idx, const double *__restrict__ frac,
                                                             float *__restrict__ out, long out_stride, long nout, int linear)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nout) return;
    const int ch = blockIdx.y, i = idx[k];
    const float *row = in + (long)ch * in_stride;
    const float a = i < 0 ? last[ch] : row[i];
    float v = a;
    if (linear) {
        const float d = row[i + 1] - a;
        v = (float)((double)a + frac[k] * (double)d);
    }
    out[(long)ch * out_stride + k] = v;
}
hipError_t launch_src_zoh_linear(const float *in, long in_stride, const float *last, const int *idx, const double *frac, float *out,
                                 long out_stride, long nout, int nchan, bool linear, hipStream_t s)
{
    if (nout <= 0 || nchan <= 0) return hipSuccess;
    dim3 grid((unsigned)((nout + 255) / 256), (unsigned)nchan);
    hipLaunchKernelGGL(src_zoh_linear_kernel, grid, dim3(256), 0, s, in, in_stride, last, idx, frac, out, out_stride, nout, linear ? 1 : 0);
    return hipGetLastError();
}

// interleaved frames [frames][nchan] <-> rows [nchan][stride] (the multi-channel form of the src_process drop-in)
__global__ __launch_bounds__(256) void src_deinterleave_kernel(const float *__restrict__ inter, float *__restrict__ rows, long stride, long frames, int nchan, int to_rows)
{
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= frames * nchan) return;
    const long fr = e / nchan;
    const int ch = (int)(e - fr * nchan);
    if (to_rows) rows[(long)ch * stride + fr] = inter[e];
    else const_cast<float *>(inter)[e] = rows[(long)ch * stride + fr];
}
hipError_t launch_src_interleave(float *inter, float *rows, long stride, long frames, int nchan, bool to_rows, hipStream_t s)
{
    if (frames <= 0 || nchan <= 0) return hipSuccess;
    const long total = frames * nchan;
    hipLaunchKernelGGL(src_deinterleave_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, inter, rows, stride, frames, nchan, to_rows ? 1 : 0);
    return hipGetLastError();
}

} // namespace redio
