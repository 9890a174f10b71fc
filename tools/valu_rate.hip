// valu_rate.hip -- microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (SGPR multiplier) on gfx950,
// at 1/2/4 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 tools/valu_rate.hip -o gpurun_out/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE, int NACC = 8>
__global__ void k(float *out, float h0, float h1, int iters)
{
    v2f a[8];
    for (int i = 0; i < 8; ++i) a[i] = v2f{(float)threadIdx.x + i, (float)i};
    v2f x = v2f{(float)threadIdx.x * 0.001f, 0.5f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                const int i = ii % NACC;
                if (MODE == 0) { // packed: one v_pk_fma_f32 per complex MAC
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "v"(x), "s"(v2f{h0, h1}));
                } else {         // scalar: two v_fmac_f32
                    asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a[i].x) : "v"(x.x), "s"(h0));
                    asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a[i].y) : "v"(x.y), "s"(h0));
                }
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int mode = 0; mode < 2; ++mode) {
            dim3 grid(256), block(256 * wps); // one block per CU, wps waves per SIMD
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, 1.0001f, 0.9999f, iters);
                else hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, 1.0001f, 0.9999f, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double cmacs = (double)iters * 64 * 256.0 * 256 * wps; // complex MACs (2 FMA each)
            printf("waves/SIMD=%d mode=%s  %.3f ms  %.2f T complex-MAC/s  = %.1f TFLOP/s\n", wps, mode == 0 ? "v_pk_fma_f32" : "2x v_fmac_f32", ms, cmacs / ms / 1e9, cmacs * 4 / ms / 1e9);
        }
    }
    // dependent-chain sensitivity: NACC independent accumulators per wave
    for (int wps = 1; wps <= 4; ++wps) {
        dim3 grid(256), block(256 * wps);
        auto run = [&](auto kern, int nacc) {
            for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(kern, grid, block, 0, 0, d, 1.0001f, 0.9999f, iters); hipEventRecord(e1); hipEventSynchronize(e1); }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double cmacs = (double)iters * 64 * 256.0 * 256 * wps;
            printf("waves/SIMD=%d v_pk_fma_f32 nacc=%d  %.2f T complex-MAC/s\n", wps, nacc, cmacs / ms / 1e9);
        };
        run(k<0, 1>, 1); run(k<0, 2>, 2); run(k<0, 4>, 4); run(k<0, 8>, 8);
    }
    return 0;
}
