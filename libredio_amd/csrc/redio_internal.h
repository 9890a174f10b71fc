// redio_internal.h -- launch entry points shared between the kernel files and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include "fft_core.h"

namespace redio {

// Launch-geometry knobs for measurement (tools/ablate.sh, tools/chain_variants.py ...): read from the environment ONLY in a
// -DREDIO_MEASURE build (make EXTRA=-DREDIO_MEASURE OUT=../_build_measure).  The shipped library reads no environment variable.
#ifdef REDIO_MEASURE
inline const char *measure_env(const char *name) { return getenv(name); }
#else
inline const char *measure_env(const char *) { return nullptr; }
#endif

// fir_kernels.hip
hipError_t launch_fir(const void *x, long n_in, const float *taps, int K, long D, void *y, long n_out,
                      bool cplx, bool fused, hipStream_t s);

// fft_kernels.hip
constexpr int FFT_MAX_STAGES = 32;
struct FftPlanDev {
    int nfft;
    int inverse;
    int nstages;
    FftStage st[FFT_MAX_STAGES];
    const float2 *tw;     // device twiddle table, nfft entries
    const float2 *tw_pass; // powers of two from 32768 up: the same values re-ordered per pass (fftbig_tables_*), else null
    const int *leaf_src;  // device table: leaf position -> input index (digit reversal), nfft entries
    const int *leaf_pos;  // the inverse: input index -> leaf position
    unsigned magic_m[FFT_MAX_STAGES]; // ceil(2^32 / m) per stage and ceil(2^32 / nfft): exact quotients by __umulhi for
    unsigned magic_n;                 // dividends below 2^16 * ... (b * m < 2^32), which LDS-resident sizes satisfy
};
// in != out on the paths that say so (hipErrorNotSupported otherwise: the C-ABI layer stages the input); work
// (nfft*nbatch float2) is needed by the global-memory path when the size has a prime factor above 5
// per-pass twiddle tables of the multi-pass transforms: element count for nfft (0: none needed) and the device-side build
size_t fftbig_tables_elems(int nfft);
hipError_t fftbig_tables_build(const float2 *tw, float2 *tables, int nfft, hipStream_t s);
hipError_t launch_fft(const FftPlanDev &p, const float2 *in, float2 *out, long nbatch, hipStream_t s, long in_stride = 0,
                      float2 *work = nullptr);

// overlap-save at nfft 1024 (one wave per block) and 4096: one kernel, no work buffers; at 4096 / 16384 tw_f / tw_i (4096) and
// Tf / Ti (16384) are the plans' stage-ordered twiddle copies (redio_fft_twiddles_pass_dev)
hipError_t launch_ovsave1k(const float2 *x, long hop, const float2 *tw_f, const float2 *tw_i, const float2 *Hc, float2 *out, long nblk,
                           float scale, hipStream_t s);
hipError_t launch_ovsave2k(const float2 *x, long hop, const float2 *Tf, const float2 *Ti, const float2 *Hc, float2 *out, long nblk,
                           float scale, hipStream_t s); // 2048-point blocks, the same scheme
hipError_t launch_ovsave8k(const float2 *x, long hop, const float2 *Tf, const float2 *Ti, const float2 *Hc, float2 *out, long nblk,
                           float scale, hipStream_t s); // 8192-point blocks: four waves per block
hipError_t launch_ovsave4k(const float2 *x, long hop, const float2 *tw_f, const float2 *tw_i, const float2 *Hc, float2 *out, long nblk,
                           float scale, hipStream_t s);
hipError_t launch_ovsave16k(const float2 *x, long hop, const float2 *tw_f, const float2 *tw_i, const float2 *Tf, const float2 *Ti, const float2 *Hc,
                            float2 *out, long nblk, float scale, hipStream_t s); // the same at nfft 16384
// overlap-save at nfft 65536: x (block b at x + b*hop) -> out (hop valid samples per block), work buffers a, b of `chunk` blocks each
// (doubled: two chunks each -- the three passes of consecutive chunks then share one launch per step)
hipError_t launch_ovsave64k(const float2 *x, long hop, float2 *a, float2 *b, const float2 *tw_f, const float2 *tw_i, const float2 *Tf,
                            const float2 *Ti, const float2 *Hc, float2 *out, long nblk, long chunk, float scale, hipStream_t s, bool doubled);

// chain_kernels.hip : FIR(K taps, decimate D) -> nfft-point forward transform, fused
bool chain_supported(int K, long D, int nfft);
hipError_t launch_chain_u8(const FftPlanDev &p, const void *bytes, const float *taps, int K, long D, float2 *out, long nblocks, bool fused,
                           hipStream_t s);
hipError_t launch_chain(const FftPlanDev &p, const float2 *x, long n_in, const float *taps, int K, long D,
                        float2 *out, long nblocks, bool fused, hipStream_t s, unsigned long long *dbg = nullptr, long dbg_cap = 0);
long chain_v4_blocks_per_wave(long nblocks, int WPS = 2); // chain_v4.hip: consecutive blocks one wavefront of the fused kernel owns (WPS wavefronts per SIMD: the chain 2, the FIR alone 3)
// the fused kernel's name as rocprofv3 prints it (spaces removed), so that a counter file can be tied to the kernel a plan launches
const char *chain_kernel_name(int K, long D, bool fused_math, char *buf, size_t cap);

// misc_kernels.hip
hipError_t launch_synth_iq(float2 *out, uint32_t seed, uint64_t first, long n, hipStream_t s);
hipError_t launch_synth_f32(float *out, uint32_t seed, uint64_t first, long n, hipStream_t s);

// overlap-save with a block size of the multi-pass transform family (32768, 131072 ...): six passes
bool ovsave_big_size(int nfft);
hipError_t launch_ovsave_big(const FftPlanDev &fw, const FftPlanDev &bw, const float2 *x, long hop, float2 *a, float2 *b, const float2 *Hc,
                             float2 *out, long nblk, float scale, hipStream_t s);
// CU count of the device the calling thread is bound to (kept per device: a process may drive several GPUs)
inline int num_cus()
{
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        int n = 0;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        cus[dev] = n > 0 ? n : 256;
    }
    return cus[dev];
}

} // namespace redio

// plan shapes for the carried-history layer (stream_carry.hip); defined next to each plan struct
struct redio_fir; struct redio_chain; struct redio_pfb; struct redio_ovsave;
void redio_fir_shape(const redio_fir *h, size_t *ntaps, size_t *decim, unsigned *flags, int *device);
void redio_chain_shape(const redio_chain *h, size_t *ntaps, size_t *decim, int *nfft, int *device);
void redio_pfb_shape(const redio_pfb *h, int *nchan, int *taps_per_branch, int *device);
void redio_ovsave_shape(const redio_ovsave *h, int *nfft, size_t *hop, int *device);

// redio_api.hip: the device twiddle table behind a public FFT handle (library-internal)
struct redio_fft;
const redio::FftPlanDev *redio_fft_plan_dev(const redio_fft *h);
const float2 *redio_fft_twiddles_dev(const redio_fft *h);
const float2 *redio_fft_twiddles_pass_dev(const redio_fft *h); // the pass-ordered copy (multi-pass sizes), else null
