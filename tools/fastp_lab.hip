// fastp_lab.hip -- A/B of the REDIO_SRC_FAST phase-split kernels on BASELINE configs[2]'s shape (256 channels, ratio 1/50, 46 tap pairs
// per phase), stand-alone: random taps and samples, a naive f32 kernel as the check, HIP-event timing after >= 100 ms of warm-up.
//   build (here):  hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-math-errno -Ilibredio_amd/csrc -Iinclude tools/fastp_lab.hip -o gpurun_out/fastp_lab
//   run (gpurun):  gpurun_out/fastp_lab [log2 frames] [S]
#include "../libredio_amd/csrc/src_kernels.hip"
#include <cstdio>
#include <vector>
#include <chrono>
using namespace redio;

__global__ void lab_fill(float *x, long n, unsigned seed)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    x[i] = (float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
}
__global__ void lab_naive(const float *x, long stride, const float *H, int KH, int S, float *out, long ostride, long nout)
{
    const long o = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nout) return;
    const float *p = x + (long)blockIdx.y * stride + S * o;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int m = 0;
    for (; m + 3 < KH; m += 4) { a0 = fmaf(H[m], p[m], a0); a1 = fmaf(H[m + 1], p[m + 1], a1); a2 = fmaf(H[m + 2], p[m + 2], a2); a3 = fmaf(H[m + 3], p[m + 3], a3); }
    for (; m < KH; ++m) a0 = fmaf(H[m], p[m], a0);
    out[(long)blockIdx.y * ostride + o] = (a0 + a1) + (a2 + a3);
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <class F> static float timeit(F f, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 0.15) { for (int i = 0; i < 4; ++i) f(); hipDeviceSynchronize(); }
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int lf = argc > 1 ? atoi(argv[1]) : 20, S = argc > 2 ? atoi(argv[2]) : 50, nch = argc > 3 ? atoi(argv[3]) : 256;
    constexpr int NPAIR = 46;
    const long frames = 1L << lf;
    const int KH = 2 * NPAIR * S - S / 2 - 6;       // a few taps short of whole rows, like the library's 4569 at S = 50
    const int cl = KH / 2;
    const long nout = (frames - KH) / S;
    const int ntap = src_fastp_row(NPAIR);
    std::vector<float> H(KH), P((size_t)S * ntap, 0.f);
    unsigned r = 12345;
    for (int m = 0; m < KH; ++m) { r = r * 1664525u + 1013904223u; H[m] = ((float)(r >> 8) / 8388608.0f - 1.0f) / 64.0f; }
    for (int m = 0; m < KH; ++m) P[(size_t)(m % S) * ntap + m / S] = H[m];
    float *dx, *dH, *dP, *o_ref, *o_old, *o_new;
    CK(hipMalloc(&dx, (size_t)nch * frames * 4)); CK(hipMalloc(&dH, KH * 4)); CK(hipMalloc(&dP, P.size() * 4));
    CK(hipMalloc(&o_ref, (size_t)nch * nout * 4)); CK(hipMalloc(&o_old, (size_t)nch * nout * 4)); CK(hipMalloc(&o_new, (size_t)nch * nout * 4));
    CK(hipMemcpy(dH, H.data(), KH * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dP, P.data(), P.size() * 4, hipMemcpyHostToDevice));
    lab_fill<<<(unsigned)((nch * frames + 255) / 256), 256>>>(dx, nch * frames, 777u);
    CK(hipMemset(o_old, 0xff, (size_t)nch * nout * 4)); CK(hipMemset(o_new, 0xff, (size_t)nch * nout * 4));
    lab_naive<<<dim3((unsigned)((nout + 255) / 256), nch), 256>>>(dx, frames, dH, KH, S, o_ref, nout, nout);
    CK(hipDeviceSynchronize());
    SrcWindow w = {nullptr, 0, dx, frames, 0};
    const long a0 = cl; // tile_base = a0 + S*k0 - cl = S*k0
    const long ntiles = (nout + 511) / 512;
    auto grid_for = [&](long splits, long *tpw) { splits = splits < 1 ? 1 : (splits > ntiles ? ntiles : splits); *tpw = (ntiles + splits - 1) / splits; return dim3((unsigned)((ntiles + *tpw - 1) / *tpw), (unsigned)nch); };
    auto check = [&](const float *o, const char *name) {
        std::vector<float> a((size_t)nch * nout), b((size_t)nch * nout);
        hipMemcpy(a.data(), o_ref, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o, b.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0; long bad = 0, first = -1;
        for (size_t i = 0; i < a.size(); ++i) { const double d = fabs((double)a[i] - b[i]); if (!(d <= 2e-4)) { ++bad; if (first < 0) first = (long)i; } if (d > worst) worst = d; }
        printf("  %s: max |diff| vs naive %.3g, %ld of %zu outputs off by more than 2e-4%s\n", name, worst, bad, a.size(), bad ? "  <-- WRONG" : "");
        if (bad) printf("    first bad output: channel %ld index %ld (tile %ld, in-tile %ld): %g vs %g\n", first / nout, first % nout, (first % nout) / 512, (first % nout) % 512, b[first], a[first]);
    };
    printf("shape: %d channels x 2^%d frames, S = %d, KH = %d taps (%d tap pairs per phase), %ld outputs per channel, %ld tiles\n", nch, lf, S, KH, NPAIR, nout, ntiles);
    const double gflop = 2.0 * KH * (double)nout * nch / 1e9;
    if (SrcFastP<NPAIR>::fits(S)) {
        auto kern = src_window_fastp_kernel<NPAIR>;
        const size_t lds = SrcFastP<NPAIR>::lds_bytes(S);
        CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (long splits : {512L / nch, 256L / nch}) {
            long tpw; dim3 g = grid_for(splits, &tpw);
            auto run = [&] { hipLaunchKernelGGL(kern, g, dim3(512), lds, 0, w, dP, KH, cl, a0, S, o_old, nout, nout, (int)tpw); };
            run(); CK(hipDeviceSynchronize()); CK(hipGetLastError());
            const float ms = timeit(run, 20);
            printf("round-3 kernel, %u workgroups per channel (%ld tiles each): %.4f ms  %.1f TFLOP/s (%.1f %% of 157.3)\n", g.x, tpw, ms, gflop / ms, gflop / ms / 157.3 * 100);
        }
        check(o_old, "round-3 kernel");
    }
    if (SrcFastP2<NPAIR, 32>::fits(S)) {
        const size_t lds = SrcFastP2<NPAIR, 32>::lds_bytes(S);
        auto bench = [&](auto kern, const char *name, bool chk) {
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            for (long splits : {256L / nch}) {
                long tpw; dim3 g = grid_for(splits, &tpw);
                auto run = [&] { hipLaunchKernelGGL(kern, g, dim3(512), lds, 0, w, dP, KH, cl, a0, S, o_new, nout, nout, (int)tpw); };
                hipMemset(o_new, 0xff, (size_t)nch * nout * 4);
                run(); hipDeviceSynchronize();
                if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
                if (chk) check(o_new, name);
                const float ms = timeit(run, 20);
                printf("%s, %u workgroups per channel (%ld tiles each): %.4f ms  %.1f TFLOP/s (%.1f %% of 157.3)\n", name, g.x, tpw, ms, gflop / ms, gflop / ms / 157.3 * 100);
            }
        };
        bench(src_window_fastp2_kernel<NPAIR, 32, 0>, "round-5 kernel", true);
        bench(src_window_fastp2_kernel<NPAIR, 32, 9>, "  ablation: no requests, no parking", false);
        bench(src_window_fastp2_kernel<NPAIR, 32, 1>, "  ablation: parking, no requests", false);
        bench(src_window_fastp2_kernel<NPAIR, 32, 16>, "  ablation: every request reads the workgroup's FIRST tile (cache hits)", false);
        bench(src_window_fastp2_kernel<NPAIR, 32, 2>, "  ablation: no barriers inside the tile", false);
        bench(src_window_fastp2_kernel<NPAIR, 32, 4>, "  ablation: no output stores", false);
    } else printf("round-5 kernel does not serve S = %d\n", S);
    return 0;
}
