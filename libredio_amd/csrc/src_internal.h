// src_internal.h -- launch entry points of src_kernels.hip
#pragma once
#include <hip/hip_runtime.h>

namespace redio {
hipError_t launch_src_exact(const float *win, long win_stride, const float *coeffs, int coeff_half_len,
                            const int *pos, const int *start, const int *inc, const double *scale,
                            float *out, long out_stride, long nout, int nchan, hipStream_t s);
hipError_t launch_src_uniform(const float *win, long win_stride, const double *cl_rev, int ncl, const double *cr_rev, int ncr,
                              int pos0, int S, double scale, float *out, long out_stride, long nout, int nchan, hipStream_t s);
size_t src_uniform_lds(int nt, int S, int cl, int cr);
// periodic-phase epochs (src_kernels.hip): per-phase coefficient tables [tap][phase], P phases per Q input samples
bool src_periodic_shape(int P, int Q, int NL, int NR, int dpos_max, int G, int *NT_out, size_t *lds_bytes);
hipError_t launch_src_periodic(const float *win, long win_stride, const double *Lc, const double *Rc, const int *dpos, const int *skipL,
                               const int *skipR, int P, int Q, int NL, int NR, int maxskipL, int maxskipR, int dpos_max, int pos0,
                               double scale, float *out, long out_stride, long nout, int nchan, hipStream_t s);
// general phase with an LDS tile of the buffer image and the two wings of an output on two threads (constant increment and scale)
size_t src_tile_lds_bytes(const int *pos_host, long nout, int nt, int coeff_half_len, int increment);
hipError_t launch_src_tile(const float *old_img, long old_stride, const float *input, long in_stride, long a_in0, long a_limit,
                           const float *coeffs, int coeff_half_len, const int *pos, const int *start,
                           int increment, double scale, float *out, long out_stride, long nout, int nchan, size_t lds_bytes, hipStream_t s);
hipError_t launch_src_window_image(const float *old_img, long old_stride, const float *input, long in_stride, long a_in0, long A0f, long j0, long j1,
                                   float *new_img, int nchan, hipStream_t s);
hipError_t launch_src_window(const float *old_img, long old_stride, const float *input, long in_stride, long a_in0,
                             const double *cl_rev, int ncl, const double *cr_rev, int ncr, const float2 *T2, int nm, const float *Hp, int fastp_nc,
                             bool fast, long a0, int S, double scale, float *out, long out_stride, long nout, int nchan,
                             long A0f, long j0, long j1, float *new_img, hipStream_t s);
// f32 polyphase path, phase-split kernel: tap pairs per phase of the instantiation that serves the shape (0: none), floats per table row
int src_fastp_pairs(int S, int KH);
int src_fastp_row(int npair);
hipError_t launch_src_copy_rows(const float *src, long src_stride, long src_off, float *dst, long dst_stride, long dst_off,
                                long n, int nchan, hipStream_t s);
hipError_t launch_src_fill_rows(float *dst, long dst_stride, long dst_off, long n, int nchan, float v, hipStream_t s);
hipError_t launch_src_zoh_linear(const float *in, long in_stride, const float *last, const int *idx, const double *frac, float *out,
                                 long out_stride, long nout, int nchan, bool linear, hipStream_t s);
hipError_t launch_src_interleave(float *inter, float *rows, long stride, long frames, int nchan, bool to_rows, hipStream_t s);
} // namespace redio
