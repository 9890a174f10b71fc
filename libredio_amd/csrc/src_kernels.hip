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
