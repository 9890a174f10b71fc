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
    const float *x = win + (long)blockIdx.y * win_stride;
    const int pad = (S & 1) ? 0 : 1;
    const int cl = ncl - 1, cr = ncr - 1;
    const long tile_base = (long)pos0 + (long)S * k0 - cl; // buffer index of tile-relative sample 0
    const int span = (NT - 1) * S + cl + cr + 2;
    // the last tile may reach past the outputs that exist: clamp the load to what the valid outputs need
    const long nvalid = (nout - k0 < NT) ? nout - k0 : NT;
    const int need = (int)((nvalid - 1) * S) + cl + cr + 2;
    for (int n = tid; n < span; n += NT) xs[n + (pad ? n / S : 0)] = (n < need) ? x[tile_base + n] : 0.0f;
    __syncthreads();
    if (k0 + tid >= nout) return;
    const int lbase = (S + pad) * tid;
    // the two wings are independent sums, each strictly ordered (far end first): run them side by side
    // so that two dependent double-precision chains are in flight per lane
    double left = 0.0, right = 0.0;
    const int c = cl + 1 + cr;
    int lrem = 0, lquo = 0;                  // left data index  t      = lquo*S + lrem
    int rquo = c / S, rrem = c - rquo * S;   // right data index c - t  = rquo*S + rrem
    const int both = cl < cr ? cl : cr;      // cl == cr or cl == cr + 1
    int t = 0;
    for (; t <= both; ++t) {
        const float xl = xs[lbase + t + (pad ? lquo : 0)];
        const float xr = xs[lbase + (c - t) + (pad ? rquo : 0)];
        left += cl_rev[t] * (double)xl;
        right += cr_rev[t] * (double)xr;
        if (++lrem == S) { lrem = 0; ++lquo; }
        if (--rrem < 0) { rrem += S; --rquo; }
    }
    for (; t <= cl; ++t) {
        left += cl_rev[t] * (double)xs[lbase + t + (pad ? lquo : 0)];
        if (++lrem == S) { lrem = 0; ++lquo; }
    }
    for (; t <= cr; ++t) {
        right += cr_rev[t] * (double)xs[lbase + (c - t) + (pad ? rquo : 0)];
        if (--rrem < 0) { rrem += S; --rquo; }
    }
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

} // namespace redio
