// pfb_api.hip -- C ABI of the polyphase channelizer plan (include/redio.h, redio_pfb_*).
#include "../../include/redio.h"
#include "pfb_core.h"
#include "redio_internal.h"
#include <new>
#include <string.h>
#include <vector>

namespace redio {
bool pfb_supported(int nchan, int taps_per_branch);
hipError_t launch_pfb_u8(const void *bytes, const float *h, const float2 *tw64, float2 *out, long rows, int taps_per_branch, int ngroups, bool fused,
                         hipStream_t s);
hipError_t launch_pfb(const float2 *x, const float *h, const float2 *tw64, float2 *out, long rows, int taps_per_branch,
                      int ngroups, bool fused, hipStream_t s);
// fft_kernels.hip: 32, 128, 256, 512 or 1024 channels with 4, 8 or 16 taps per branch in one kernel (branch filters + the M-point transform in LDS)
bool pfb_p2_supported(int nchan, int taps_per_branch);
hipError_t launch_pfb_p2(const float2 *x, const float *h, const float2 *tw, const float2 *Tord, float2 *out, long rows, int nchan, int taps_per_branch,
                         int ngroups, bool fused, hipStream_t s);

// ---- any channel count / branch length: branch filters, then the plan's M-point transform per row ----------
// v[t][m] = fold_p x[(t + p) M + m] * h[M p + m] (ascending p, the reference's fold).  The transform of each row is then one batched call of the FFT plan for M points.
// Two passes over HBM instead of the fused kernel's one, bit-identical to oracle orc_pfb_channelizer.
constexpr int PFB_BR = 8; // rows per thread in the branch kernel
template <bool FUSED>
__global__ __launch_bounds__(256) void pfb_branch_kernel(const float2 *__restrict__ x, const float *__restrict__ h, float2 *__restrict__ v,
                                                         long rows, int M, int P)
{
    // thread = (branch m, block of PFB_BR consecutive rows): the P + PFB_BR - 1 samples of its column slide through
    // a register window, so each is loaded once per thread; consecutive threads are consecutive m (coalesced rows)
    constexpr int R = PFB_BR;
    const long nblk = (rows + R - 1) / R, total = nblk * M;
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const long tb = e / M;
    const int m = (int)(e - tb * M);
    const long t0 = tb * R;
    const long last = rows + P - 2; // last input row that exists for these outputs
    const float2 *col = x + m;
    float2 win[R], acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long t = t0 + r < last ? t0 + r : last;
        win[r] = col[t * M];
        acc[r] = make_float2(0.f, 0.f);
    }
    for (int p0 = 0; p0 < P; p0 += R) {
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int p = p0 + j;
            if (p < P) { // wave-uniform
                const float g = h[(long)p * M + m];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = mac<FUSED>(win[(r + j) % R], g, acc[r]); // row t0 + r + p
                const long t = t0 + R + p < last ? t0 + R + p : last;                        // slot j now holds row t0 + j + p: done with it
                win[j] = col[t * M];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (t0 + r < rows) v[(t0 + r) * M + m] = acc[r];
}

// [row][M] -> [group][row][M / ngroups] (the layout the multi-GPU exchange sends)
__global__ __launch_bounds__(256) void pfb_regroup_kernel(const float2 *__restrict__ in, float2 *__restrict__ out, long rows, int M, int ngroups)
{
    const long total = rows * M, stride = (long)gridDim.x * blockDim.x;
    const int cpg = M / ngroups;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long row = e / M;
        const int ch = (int)(e - row * M), g = ch / cpg;
        out[(long)g * rows * cpg + row * cpg + (ch - g * cpg)] = in[e];
    }
}

} // namespace redio
using namespace redio;

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? REDIO_OK : REDIO_ERR_HIP_BASE - (int)e; }

struct redio_pfb {
    int device, nchan, taps_per_branch;
    unsigned flags;
    float *d_h;
    redio_fft *fft; // generic shapes: the M-point transform applied to every row
    float2 *d_tw;
    bool fused_kernel;       // the 64-channel kernel of pfb_kernels.hip
    float2 *d_v, *d_w;       // generic shapes: branch outputs / transform outputs before regrouping (grown on first use)
    size_t v_elems, w_elems;
    float2 *d_conv;          // redio_pfb_enqueue_u8 on shapes without the one-kernel form: the converted samples (grown on first use)
    size_t conv_elems;
};

extern "C" int redio_pfb_create(redio_pfb **h, const float *proto, int nchan, int taps_per_branch, unsigned flags)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (!proto || nchan <= 0 || taps_per_branch <= 0) return REDIO_ERR_ARG;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return REDIO_ERR_NO_DEVICE;
    redio_pfb *p = new (std::nothrow) redio_pfb();
    if (!p) return REDIO_ERR_NOMEM;
    p->device = dev; p->nchan = nchan; p->taps_per_branch = taps_per_branch; p->flags = flags; p->d_h = nullptr; p->d_tw = nullptr;
    p->fft = nullptr; p->d_v = p->d_w = nullptr; p->v_elems = p->w_elems = 0; p->d_conv = nullptr; p->conv_elems = 0;
    p->fused_kernel = pfb_supported(nchan, taps_per_branch);
    const size_t nt = (size_t)nchan * taps_per_branch;
    std::vector<float2> tw((size_t)nchan);
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    for (int i = 0; i < nchan; ++i) {
        const double phase = -2 * pi * i / nchan;
        tw[(size_t)i] = make_float2((float)cos(phase), (float)sin(phase));
    }
    hipError_t e = hipMalloc((void **)&p->d_h, nt * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_h, proto, nt * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_tw, (size_t)nchan * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(p->d_tw, tw.data(), (size_t)nchan * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(p->d_h); hipFree(p->d_tw); delete p; return hip_rc(e); }
    if (!p->fused_kernel) {
        const int rc = redio_fft_create(&p->fft, nchan, 0);
        if (rc != REDIO_OK) { hipFree(p->d_h); hipFree(p->d_tw); delete p; return rc; }
    }
    *h = p;
    return REDIO_OK;
}

extern "C" int redio_pfb_destroy(redio_pfb *h)
{
    if (!h) return REDIO_OK;
    hipFree(h->d_h);
    hipFree(h->d_tw);
    hipFree(h->d_v);
    hipFree(h->d_w);
    hipFree(h->d_conv);
    redio_fft_destroy(h->fft);
    delete h;
    return REDIO_OK;
}

extern "C" size_t redio_pfb_nrows(const redio_pfb *h, size_t n_in)
{
    if (!h) return 0;
    const size_t T = n_in / (size_t)h->nchan;
    return T < (size_t)h->taps_per_branch ? 0 : T - (size_t)h->taps_per_branch + 1;
}

// scratch of the two-pass shapes (branch outputs; transform outputs when regrouping) for inputs of up to n_in samples
extern "C" int redio_pfb_reserve(redio_pfb *h, size_t n_in, int ngroups)
{
    if (!h) return REDIO_ERR_ARG;
    if (h->fused_kernel) return REDIO_OK;
    // the one-kernel shapes (32 ... 1024 channels x 4 / 8 / 16 taps) touch the scratch only for an output that is not 16-byte aligned:
    // 2-4 GiB per 2^28-sample message held for nothing otherwise.  REDIO_PFB_RESERVE_TWO_PASS in `ngroups` asks for it all the same
    // (the two-pass form then never allocates, e.g. inside a capture with such an output).
    const bool want_two_pass = (ngroups & REDIO_PFB_RESERVE_TWO_PASS) != 0;
    ngroups &= ~REDIO_PFB_RESERVE_TWO_PASS;
    if (pfb_p2_supported(h->nchan, h->taps_per_branch) && !want_two_pass) return REDIO_OK;
    const size_t total = redio_pfb_nrows(h, n_in) * (size_t)h->nchan;
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return hip_rc(e);
    if (total > h->v_elems) {
        hipFree(h->d_v); h->d_v = nullptr; h->v_elems = 0;
        e = hipMalloc((void **)&h->d_v, total * sizeof(float2));
        if (e != hipSuccess) return hip_rc(e);
        h->v_elems = total;
    }
    if (ngroups > 1 && total > h->w_elems) {
        hipFree(h->d_w); h->d_w = nullptr; h->w_elems = 0;
        e = hipMalloc((void **)&h->d_w, total * sizeof(float2));
        if (e != hipSuccess) return hip_rc(e);
        h->w_elems = total;
    }
    return REDIO_OK;
}
extern "C" int redio_pfb_reserve_two_pass(redio_pfb *h, size_t n_in, int ngroups)
{
    if (ngroups < 1) return REDIO_ERR_ARG;
    return redio_pfb_reserve(h, n_in, ngroups | REDIO_PFB_RESERVE_TWO_PASS);
}
void redio_pfb_shape(const redio_pfb *h, int *nchan, int *taps_per_branch, int *device)
{
    *nchan = h->nchan; *taps_per_branch = h->taps_per_branch; *device = h->device;
}

extern "C" int redio_pfb_enqueue(redio_pfb *h, const void *d_in, size_t n_in, void *d_out, int ngroups, void *stream)
{
    if (!h) return REDIO_ERR_ARG;
    const size_t rows = redio_pfb_nrows(h, n_in);
    if (rows == 0) return REDIO_OK;
    if (!d_in || !d_out || d_in == d_out) return REDIO_ERR_ARG;
    if (ngroups < 1 || h->nchan % ngroups) return REDIO_ERR_ARG;
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return hip_rc(e);
    if (!h->fused_kernel && pfb_p2_supported(h->nchan, h->taps_per_branch)) { // one kernel for these shapes too (16-byte aligned output)
        e = launch_pfb_p2((const float2 *)d_in, h->d_h, h->d_tw, redio_fft_twiddles_pass_dev(h->fft), (float2 *)d_out, (long)rows, h->nchan,
                          h->taps_per_branch, ngroups, (h->flags & REDIO_FIR_FUSED) != 0, (hipStream_t)stream);
        if (e != hipErrorNotSupported) return hip_rc(e);
    }
    if (!h->fused_kernel) {
        const size_t total = rows * (size_t)h->nchan;
        hipStream_t st = (hipStream_t)stream;
        if (total > h->v_elems || (ngroups > 1 && total > h->w_elems)) { // un-reserved: grow on first use, never inside a capture
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return REDIO_ERR_NOT_RESERVED;
            const int rc = redio_pfb_reserve(h, n_in, ngroups | REDIO_PFB_RESERVE_TWO_PASS);
            if (rc != REDIO_OK) return rc;
        }
        long blocks = (long)((total + 255) / 256);
        if (blocks > 65536) blocks = 65536;
        const long bthreads = (long)((rows + PFB_BR - 1) / PFB_BR) * h->nchan;
        const unsigned bgrid = (unsigned)((bthreads + 255) / 256);
        if (h->flags & REDIO_FIR_FUSED)
            hipLaunchKernelGGL(pfb_branch_kernel<true>, dim3(bgrid), dim3(256), 0, st, (const float2 *)d_in, h->d_h, h->d_v, (long)rows,
                               h->nchan, h->taps_per_branch);
        else
            hipLaunchKernelGGL(pfb_branch_kernel<false>, dim3(bgrid), dim3(256), 0, st, (const float2 *)d_in, h->d_h, h->d_v, (long)rows,
                               h->nchan, h->taps_per_branch);
        int rc = redio_fft_enqueue(h->fft, h->d_v, ngroups > 1 ? (void *)h->d_w : d_out, rows, stream);
        if (rc != REDIO_OK) return rc;
        if (ngroups > 1)
            hipLaunchKernelGGL(pfb_regroup_kernel, dim3((unsigned)blocks), dim3(256), 0, st, h->d_w, (float2 *)d_out, (long)rows, h->nchan, ngroups);
        return hip_rc(hipGetLastError());
    }
    e = launch_pfb((const float2 *)d_in, h->d_h, h->d_tw, (float2 *)d_out, (long)rows, h->taps_per_branch, ngroups,
                   (h->flags & REDIO_FIR_FUSED) != 0, (hipStream_t)stream);
    if (e == hipErrorNotSupported) return REDIO_ERR_UNSUPPORTED;
    return hip_rc(e);
}

// sizes what redio_pfb_enqueue_u8 needs beyond the one-kernel form for messages of up to nbytes bytes
extern "C" int redio_pfb_reserve_u8(redio_pfb *h, size_t nbytes, int ngroups)
{
    if (!h) return REDIO_ERR_ARG;
    const size_t n_in = nbytes / 2;
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return hip_rc(e);
    if (n_in > h->conv_elems) {
        hipFree(h->d_conv); h->d_conv = nullptr; h->conv_elems = 0;
        e = hipMalloc((void **)&h->d_conv, n_in * sizeof(float2));
        if (e != hipSuccess) return hip_rc(e);
        h->conv_elems = n_in;
    }
    return redio_pfb_reserve(h, n_in, ngroups);
}

// rtlsdr::data_to_samples (rtlsdr.rs:159-162) -> the channelizer, from the receiver's u8 I/Q bytes
extern "C" int redio_pfb_enqueue_u8(redio_pfb *h, const void *d_bytes, size_t nbytes, void *d_out, int ngroups, void *stream)
{
    if (!h || (nbytes & 1)) return REDIO_ERR_ARG;
    const size_t n_in = nbytes / 2;
    const size_t rows = redio_pfb_nrows(h, n_in);
    if (rows == 0) return REDIO_OK;
    if (!d_bytes || !d_out || d_bytes == d_out) return REDIO_ERR_ARG;
    if (ngroups < 1 || h->nchan % ngroups) return REDIO_ERR_ARG;
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return hip_rc(e);
    if (h->fused_kernel) {
        e = launch_pfb_u8(d_bytes, h->d_h, h->d_tw, (float2 *)d_out, (long)rows, h->taps_per_branch, ngroups, (h->flags & REDIO_FIR_FUSED) != 0,
                          (hipStream_t)stream);
        if (e != hipErrorNotSupported) return hip_rc(e);
    }
    if (n_in > h->conv_elems) { // other shapes, un-reserved (redio_pfb_reserve_u8): grow on first use, never inside a capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return REDIO_ERR_NOT_RESERVED;
        e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) return hip_rc(e);
        const int rc = redio_pfb_reserve_u8(h, nbytes, ngroups);
        if (rc) return rc;
    }
    const int rc = redio_data_to_samples(d_bytes, nbytes, h->d_conv, stream);
    if (rc) return rc;
    return redio_pfb_enqueue(h, h->d_conv, n_in, d_out, ngroups, stream);
}
