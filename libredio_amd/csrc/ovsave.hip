// ovsave.hip -- overlap-save FFT convolution (BASELINE.json configs[4]): the valid-mode correlation of
// dsputils::convolve (src/dsputils/src/dsputils.rs:30-32) on cf32 with real taps, evaluated per block
// of nfft samples with the kissfft-order transforms of fft_kernels.hip:
//     X = FFT(block b at x + b*hop),  Y = X .* conj(H),  y = IFFT(Y),  out[b*hop + i] = y[i]/nfft, i < hop
// with hop = nfft - ntaps + 1 and H = FFT(taps zero-padded).  Bit-identical to oracle
// orc_overlap_save (same transforms, same C_MUL order, same 1/nfft scale).
// Algorithmic bytes per output sample: 8*nfft/hop read + 8 written (17.14 B at nfft 65536, 8193 taps).
#include "../../include/redio.h"
#include "redio_internal.h"
#include <new>
#include <stdlib.h>
#include <vector>

namespace redio {

__global__ __launch_bounds__(256) void ovsave_mul_kernel(float2 *X, const float2 *__restrict__ Hc, long total, int nfft)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    X[i] = cmul_rn(X[i], Hc[i % nfft]);
}

__global__ __launch_bounds__(256) void ovsave_conj_kernel(float2 *H, int nfft)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nfft) H[i].y = -H[i].y;
}

__global__ __launch_bounds__(256) void ovsave_scale_out_kernel(const float2 *__restrict__ y, float2 *__restrict__ out, long nblk,
                                                               int nfft, long hop, float scale)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nblk * hop) return;
    const long b = i / hop, r = i - b * hop;
    const float2 v = y[b * nfft + r];
    out[i] = make_float2(mul_rn(v.x, scale), mul_rn(v.y, scale));
}

} // namespace redio
using namespace redio;

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? REDIO_OK : REDIO_ERR_HIP_BASE - (int)e; }
#define OV_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return hip_rc(_e); } while (0)

struct redio_ovsave {
    int device, nfft;
    size_t ntaps, hop;
    redio_fft *fw, *bw;
    float2 *d_Hc;
    float2 *d_a, *d_b; // work buffers, chunk_blocks * nfft each
    size_t chunk_blocks;
};

// the plan structs live in redio_api.hip; reach the device plan through the public enqueue only
extern "C" int redio_ovsave_create(redio_ovsave **h, const float *taps, size_t ntaps, int nfft)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (!taps || nfft <= 0) return REDIO_ERR_ARG;
    if (ntaps == 0) return REDIO_ERR_ASSERT;
    if (ntaps > (size_t)nfft) return REDIO_ERR_ARG;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return REDIO_ERR_NO_DEVICE;
    redio_ovsave *p = new (std::nothrow) redio_ovsave();
    if (!p) return REDIO_ERR_NOMEM;
    p->device = dev; p->nfft = nfft; p->ntaps = ntaps; p->hop = (size_t)nfft - ntaps + 1;
    p->fw = p->bw = nullptr; p->d_Hc = p->d_a = p->d_b = nullptr;
    // work buffers, only for the block sizes whose passes go through memory (the one-kernel sizes keep a block in
    // registers / LDS): about 64 MiB each, at least one block
    const bool one_kernel = nfft == 1024 || nfft == 2048 || nfft == 4096 || nfft == 8192 || nfft == 16384;
    // 64 MiB per buffer: swept 16 ... 128 MiB at 65536 points in round 2 (profiles/r02_c5_team_experiment.txt): 64 is best, 128 -- past the
    // Infinity Cache -- costs 6 %
    size_t chunk_mib = 64; // at 65536 points: one resident set of waves per pass, three per step launch (round 3: 21 / 56 / 64 MiB 2.76-2.80 ms, 42 / 80 / 96 MiB 2.97-3.05)
    if (const char *e = measure_env("REDIO_OVS_CHUNK_MIB")) { const long v = atol(e); if (v >= 1) chunk_mib = (size_t)v; } // measurement only
    p->chunk_blocks = (chunk_mib << 20) / ((size_t)nfft * sizeof(float2));
    if (p->chunk_blocks < 1) p->chunk_blocks = 1;
    int rc = redio_fft_create(&p->fw, nfft, 0);
    if (rc == REDIO_OK) rc = redio_fft_create(&p->bw, nfft, 1);
    hipError_t e = hipSuccess;
    if (rc == REDIO_OK) {
        float2 *d_pad = nullptr; // the zero-padded taps, transformed once
        e = hipMalloc((void **)&p->d_Hc, (size_t)nfft * sizeof(float2));
        const size_t nbuf = nfft == F64K_N ? 2 : 1; // 65536 points: two chunks per buffer (launch_ovsave64k's step launches)
        if (e == hipSuccess && !one_kernel) e = hipMalloc((void **)&p->d_a, nbuf * p->chunk_blocks * nfft * sizeof(float2));
        if (e == hipSuccess && !one_kernel) e = hipMalloc((void **)&p->d_b, nbuf * p->chunk_blocks * nfft * sizeof(float2));
        if (e == hipSuccess) {
            if (one_kernel) e = hipMalloc((void **)&d_pad, (size_t)nfft * sizeof(float2));
            else d_pad = p->d_a;
        }
        if (e == hipSuccess) {
            std::vector<float2> hp((size_t)nfft, make_float2(0.f, 0.f));
            for (size_t j = 0; j < ntaps; ++j) hp[j].x = taps[j];
            e = hipMemcpy(d_pad, hp.data(), (size_t)nfft * sizeof(float2), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) {
            rc = redio_fft_enqueue(p->fw, d_pad, p->d_Hc, 1, nullptr);
            if (rc == REDIO_OK) {
                hipLaunchKernelGGL(ovsave_conj_kernel, dim3((unsigned)((nfft + 255) / 256)), dim3(256), 0, nullptr, p->d_Hc, nfft);
                e = hipDeviceSynchronize();
            }
        }
        if (one_kernel && d_pad) hipFree(d_pad);
    }
    if (rc != REDIO_OK || e != hipSuccess) {
        redio_ovsave_destroy(p);
        return rc != REDIO_OK ? rc : hip_rc(e);
    }
    *h = p;
    return REDIO_OK;
}

extern "C" int redio_ovsave_destroy(redio_ovsave *h)
{
    if (!h) return REDIO_OK;
    redio_fft_destroy(h->fw); redio_fft_destroy(h->bw);
    hipFree(h->d_Hc);
    if (h->d_a) hipFree(h->d_a);
    if (h->d_b) hipFree(h->d_b);
    delete h;
    return REDIO_OK;
}

void redio_ovsave_shape(const redio_ovsave *h, int *nfft, size_t *hop, int *device) { *nfft = h->nfft; *hop = h->hop; *device = h->device; }

extern "C" size_t redio_ovsave_nout(const redio_ovsave *h, size_t n_in)
{
    if (!h || n_in < (size_t)h->nfft) return 0;
    return ((n_in - (size_t)h->nfft) / h->hop + 1) * h->hop;
}

extern "C" int redio_ovsave_enqueue(redio_ovsave *h, const void *d_in, size_t n_in, void *d_out, void *stream)
{
    if (!h) return REDIO_ERR_ARG;
    const size_t nout = redio_ovsave_nout(h, n_in);
    if (nout == 0) return REDIO_OK;
    if (!d_in || !d_out || d_in == d_out) return REDIO_ERR_ARG;
    OV_TRY(hipSetDevice(h->device));
    hipStream_t st = (hipStream_t)stream;
    const size_t nblk = nout / h->hop;
    const float scale = 1.0f / (float)h->nfft;
    if (h->nfft == 1024) {
        OV_TRY(launch_ovsave1k((const float2 *)d_in, (long)h->hop, redio_fft_twiddles_dev(h->fw), redio_fft_twiddles_dev(h->bw), h->d_Hc,
                               (float2 *)d_out, (long)nblk, scale, st));
        return REDIO_OK;
    }
    if (h->nfft == 2048) {
        OV_TRY(launch_ovsave2k((const float2 *)d_in, (long)h->hop, redio_fft_twiddles_pass_dev(h->fw), redio_fft_twiddles_pass_dev(h->bw), h->d_Hc,
                               (float2 *)d_out, (long)nblk, scale, st));
        return REDIO_OK;
    }
    if (h->nfft == 8192) {
        OV_TRY(launch_ovsave8k((const float2 *)d_in, (long)h->hop, redio_fft_twiddles_pass_dev(h->fw), redio_fft_twiddles_pass_dev(h->bw), h->d_Hc,
                               (float2 *)d_out, (long)nblk, scale, st));
        return REDIO_OK;
    }
    if (h->nfft == 4096) { // block load, both transforms, product, scale and store in one kernel (fft_kernels.hip)
        OV_TRY(launch_ovsave4k((const float2 *)d_in, (long)h->hop, redio_fft_twiddles_pass_dev(h->fw), redio_fft_twiddles_pass_dev(h->bw), h->d_Hc,
                               (float2 *)d_out, (long)nblk, scale, st));
        return REDIO_OK;
    }
    if (h->nfft == 16384) {
        OV_TRY(launch_ovsave16k((const float2 *)d_in, (long)h->hop, redio_fft_twiddles_dev(h->fw), redio_fft_twiddles_dev(h->bw),
                                redio_fft_twiddles_pass_dev(h->fw), redio_fft_twiddles_pass_dev(h->bw), h->d_Hc,
                                (float2 *)d_out, (long)nblk, scale, st));
        return REDIO_OK;
    }
    if (h->nfft == F64K_N) { // three passes per chunk; one launch per step runs the middle pass of chunk k, the last of k - 1 and the first of k + 1 (fft_kernels.hip)
        OV_TRY(launch_ovsave64k((const float2 *)d_in, (long)h->hop, h->d_a, h->d_b, redio_fft_twiddles_dev(h->fw), redio_fft_twiddles_dev(h->bw),
                                redio_fft_twiddles_pass_dev(h->fw), redio_fft_twiddles_pass_dev(h->bw), h->d_Hc, (float2 *)d_out, (long)nblk,
                                (long)h->chunk_blocks, scale, st, true));
        return REDIO_OK;
    }
    for (size_t b0 = 0; b0 < nblk; b0 += h->chunk_blocks) {
        const size_t nb = (nblk - b0 < h->chunk_blocks) ? nblk - b0 : h->chunk_blocks;
        const long total = (long)(nb * (size_t)h->nfft);
        if (ovsave_big_size(h->nfft)) { // the product and the scaled copy ride on the inverse transform's first and last pass
            OV_TRY(launch_ovsave_big(*redio_fft_plan_dev(h->fw), *redio_fft_plan_dev(h->bw), (const float2 *)d_in + b0 * h->hop, (long)h->hop, h->d_a,
                                     h->d_b, h->d_Hc, (float2 *)d_out + b0 * h->hop, (long)nb, scale, st));
            continue;
        }
        int rc = redio_fft_enqueue_strided(h->fw, (const float2 *)d_in + b0 * h->hop, h->d_a, nb, (long)h->hop, st);
        if (rc) return rc;
        hipLaunchKernelGGL(ovsave_mul_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, h->d_a, h->d_Hc, total, h->nfft);
        rc = redio_fft_enqueue(h->bw, h->d_a, h->d_b, nb, st);
        if (rc) return rc;
        const long no = (long)(nb * h->hop);
        hipLaunchKernelGGL(ovsave_scale_out_kernel, dim3((unsigned)((no + 255) / 256)), dim3(256), 0, st, h->d_b,
                           (float2 *)d_out + b0 * h->hop, (long)nb, h->nfft, (long)h->hop, scale);
        OV_TRY(hipGetLastError());
    }
    return REDIO_OK;
}
