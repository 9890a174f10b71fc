// fir_run.hip -- dsputils::convolve (dsputils.rs:30-32) on REAL samples in the wave-private run form (fir_run_core.h): one wavefront per
// run of consecutive sub-tiles, halo carried in LDS, register prefetch of the next sub-tile, no workgroup barrier, taps in SGPRs.
// Bit-exact with the oracle (strict left fold per output, fir_core.h).  HBM-bound by design (4 + 4 / D bytes per sample), with the
// multiply-add floor close behind for 63 taps / 1 (126 flops per sample).
#include "fir_run_core.h"
#include "fft_wave.h" // wave_lds_fence
#include "redio_internal.h"

namespace redio {

typedef float run_v4f __attribute__((ext_vector_type(4)));

template <int K, int D, int R, bool FUSED, int WPS>
__global__ __launch_bounds__(64, WPS) void fir_run_real_kernel(const float *__restrict__ x, const float *__restrict__ taps, float *__restrict__ y,
                                                               long nsub, long sub_per_wave)
{
    using U = FirRunReal<K, D, R>;
    using G = typename U::G;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);
    const int lane = threadIdx.x;
    const long s0 = (long)blockIdx.x * sub_per_wave;
    if (s0 >= nsub) return;
    const long s1 = s0 + sub_per_wave < nsub ? s0 + sub_per_wave : nsub;
    const long n = s1 - s0;
    const run_v4f *src0 = reinterpret_cast<const run_v4f *>(x + s0 * U::SUB_NEW) + lane;

    run_v4f pre[U::NLD];
    auto fetch = [&](long j) {
        const run_v4f *src = src0 + U::HALO_A / 4 + j * (U::SUB_NEW / 4);
#pragma unroll
        for (int i = 0; i < U::NLD; ++i) pre[i] = __builtin_nontemporal_load(src + 64 * i); // the stream is read once
    };
    auto park = [&]() {
#pragma unroll
        for (int i = 0; i < U::NLD; ++i) {
            xs[G::lds_index(U::new_sample(lane, i, 0))] = pre[i].x;
            xs[G::lds_index(U::new_sample(lane, i, 1))] = pre[i].y;
            xs[G::lds_index(U::new_sample(lane, i, 2))] = pre[i].z;
            xs[G::lds_index(U::new_sample(lane, i, 3))] = pre[i].w;
        }
    };
    // prologue: the head (the only halo this wave ever fetches) and the first sub-tile
    if (lane < U::HALO_A / 4) {
        const run_v4f q = src0[0];
        xs[G::lds_index(U::head_sample(lane, 0))] = q.x;
        xs[G::lds_index(U::head_sample(lane, 1))] = q.y;
        xs[G::lds_index(U::head_sample(lane, 2))] = q.z;
        xs[G::lds_index(U::head_sample(lane, 3))] = q.w;
    }
    fetch(0);
    park();
    wave_lds_fence();
#pragma unroll 1
    for (long j = 0; j < n; ++j) {
        const bool more = j + 1 < n;
        if (more) fetch(j + 1);
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        int lf = lane; // opaque per sub-tile: keeps LDS address arithmetic out of the loop's live set
        asm volatile("" : "+v"(lf));
        fir_lane<float, K, D, R, FUSED>(xs, lf, taps, acc);
        // R consecutive outputs per lane: whole 16-byte stores, the spectrum-free twin of the chain's FIR_ONLY path
        run_v4f *y4 = reinterpret_cast<run_v4f *>(y + (s0 + j) * U::SUB_OUT + (long)lf * R);
#pragma unroll
        for (int q = 0; q < R / 4; ++q) __builtin_nontemporal_store(run_v4f{acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]}, y4 + q);
        wave_lds_fence(); // window reads done
        // the last HALO_A samples of this image are the first HALO_A of the next one
        float halo[U::NHV];
#pragma unroll
        for (int i = 0; i < U::NHV; ++i) {
            halo[i] = 0.f;
            if (more && lf + 64 * i < U::HALO_A) halo[i] = xs[G::lds_index(U::SUB_NEW + lf + 64 * i)];
        }
        wave_lds_fence();
        if (more) {
#pragma unroll
            for (int i = 0; i < U::NHV; ++i)
                if (lf + 64 * i < U::HALO_A) xs[G::lds_index(lf + 64 * i)] = halo[i];
            park();
        }
        wave_lds_fence();
    }
}

template <int K, int D, int R, int WPS>
static hipError_t launch_run_real_t(const float *x, const float *taps, float *y, long nsub, bool fused, hipStream_t s)
{
    using U = FirRunReal<K, D, R>;
    static_assert(R % 4 == 0, "whole 16-byte stores");
    constexpr size_t LDS_NEED = (size_t)U::lds_floats() * sizeof(float);
    static_assert(4 * WPS * LDS_NEED <= 160 * 1024, "4*WPS waves per CU");
    // exactly 4*WPS waves per CU (chain_v4.hip launch_v4_t: why the LDS request is padded up to a 1/(4*WPS) share)
    constexpr size_t LDS = (160 * 1024 / (4 * WPS)) - 480 > LDS_NEED ? (160 * 1024 / (4 * WPS)) - 480 : LDS_NEED;
    // short runs in dispatch order, at least about six sets of wavefronts per launch (chain_v4_blocks_per_wave's rule), a run
    // no longer than about 4096 outputs x D of stream (the chain's four blocks)
    long waves = 4L * WPS * num_cus();
    long spw = nsub / (6 * waves);
    const long cap = 4096 / U::SUB_OUT > 1 ? 4096 / U::SUB_OUT : 1;
    if (spw < 1) spw = 1;
    if (spw > cap) spw = cap;
    if (const char *e = measure_env("REDIO_FIR_RUN_SPW")) { const long v = atol(e); if (v >= 1) spw = v; }
    const long grid = (nsub + spw - 1) / spw;
    if (fused) hipLaunchKernelGGL((fir_run_real_kernel<K, D, R, true, WPS>), dim3((unsigned)grid), dim3(64), LDS, s, x, taps, y, nsub, spw);
    else hipLaunchKernelGGL((fir_run_real_kernel<K, D, R, false, WPS>), dim3((unsigned)grid), dim3(64), LDS, s, x, taps, y, nsub, spw);
    return hipGetLastError();
}

// whole sub-tiles of a real-sample call (16-byte aligned x and y) in the run form; *done = outputs produced (the caller runs the
// remainder on the tiled kernels).  hipErrorNotSupported: no instantiation for this shape.
hipError_t launch_fir_run_real(int K, int D, const float *x, long n_in, const float *taps, float *y, long n_out, bool fused, hipStream_t s, long *done)
{
    *done = 0;
#ifdef REDIO_MEASURE // bit-identical, measured slower than the tiled kernel (profiles/r06_fir_real_forms.txt): measurement builds only
    if (K == 63 && D == 1) {
        using U = FirRunReal<63, 1, 8>;
        const long nsub = U::whole_subtiles(n_in, n_out);
        if (nsub == 0) return hipSuccess;
        *done = nsub * U::SUB_OUT;
        if (const char *e = measure_env("REDIO_FIR_RUN_WPS")) {
            if (atoi(e) == 3) return launch_run_real_t<63, 1, 8, 3>(x, taps, y, nsub, fused, s);
            if (atoi(e) == 5) return launch_run_real_t<63, 1, 8, 5>(x, taps, y, nsub, fused, s);
            if (atoi(e) == 6) return launch_run_real_t<63, 1, 8, 6>(x, taps, y, nsub, fused, s);
        }
        return launch_run_real_t<63, 1, 8, 4>(x, taps, y, nsub, fused, s); // 120 registers (fmaf build): four wavefronts per SIMD
    }
#endif
    return hipErrorNotSupported;
}

} // namespace redio
