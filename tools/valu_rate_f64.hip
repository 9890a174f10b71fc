// valu_rate_f64.hip -- microbenchmark: issue cost of the three instructions the bit-exact resampler spends per tap and lane
// (v_cvt_f64_f32, v_mul_f64, v_add_f64; libsamplerate rounds the product, so no FMA) and of v_fma_f64 for reference, on
// gfx950, at 1 / 2 / 4 waves per SIMD, 8 independent chains per wave.  Gives the VALU roofline that C3 is priced against.
//   build (cross-compiles without a GPU): hipcc -O3 --offload-arch=gfx950 tools/valu_rate_f64.hip -o tools/exp/_build_valu/valu_rate_f64
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int MODE>
__global__ void k(double *out, float xf, double c, int iters)
{
    double a[8];
    float x[8];
    for (int i = 0; i < 8; ++i) { a[i] = (double)threadIdx.x + i; x[i] = xf + (float)i + (float)threadIdx.x * 1e-3f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { double d; asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(x[i])); asm volatile("" : "+v"(d)); a[i] = d; }
                if (MODE == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 3) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (MODE == 4) { // the tap: convert, multiply, add
                    double d, p;
                    asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(x[i]));
                    asm volatile("v_mul_f64 %0, %1, %2" : "=v"(p) : "v"(d), "v"(c));
                    asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(p));
                }
            }
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main()
{
    double *d;
    hipMalloc(&d, 256 * 1024 * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    const char *names[5] = {"v_cvt_f64_f32", "v_mul_f64", "v_add_f64", "v_fma_f64", "cvt+mul+add (one tap)"};
    for (int wps = 1; wps <= 4; wps *= 2)
        for (int mode = 0; mode < 5; ++mode) {
            dim3 grid(256), block(256 * wps); // one block per CU, wps waves per SIMD
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                switch (mode) {
                case 0: hipLaunchKernelGGL(k<0>, grid, block, 0, 0, d, 1.5f, 1.0000001, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, grid, block, 0, 0, d, 1.5f, 1.0000001, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, grid, block, 0, 0, d, 1.5f, 1.0000001, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, grid, block, 0, 0, d, 1.5f, 1.0000001, iters); break;
                default: hipLaunchKernelGGL(k<4>, grid, block, 0, 0, d, 1.5f, 1.0000001, iters); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double ops = (double)iters * 64 * wps;          // wave-instructions (or taps) per SIMD
            printf("waves/SIMD=%d %-24s %.3f ms  -> %.2f ns per wave-%s per SIMD = %.1f cycles at 2.4 GHz\n", wps, names[mode], ms,
                   ms * 1e6 / ops, mode == 4 ? "tap" : "instruction", ms * 1e6 / ops * 2.4);
        }
    return 0;
}
