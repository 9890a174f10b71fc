// stream_probe.hip -- what a plain streaming kernel sustains on this MI355X: read-only and
// "read 5 : write 1" (the chain's traffic mix), 16 B per lane, persistent grid-stride loops.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void read_only(const v4f *__restrict__ x, long n4, float *sink)
{
    v4f acc = {0, 0, 0, 0};
    const long stride = (long)gridDim.x * blockDim.x * UNROLL;
    for (long i = (long)blockIdx.x * blockDim.x * UNROLL + threadIdx.x; i < n4; i += stride) {
        v4f v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = (i + u * 256 < n4) ? x[i + u * 256] : v4f{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// read 5 float4, write 1 float4 (sum) : same byte ratio as the chain
__global__ __launch_bounds__(256) void read5_write1(const v4f *__restrict__ x, v4f *__restrict__ y, long nout)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long o = (long)blockIdx.x * blockDim.x + threadIdx.x; o < nout; o += stride) {
        const long blk = o / 256, l = o % 256;
        const v4f *p = x + blk * 1280 + l;
        v4f a = p[0], b = p[256], c = p[512], d = p[768], e = p[1024];
        y[o] = a + b + c + d + e;
    }
}

int main()
{
    const long n = 1L << 28;             // cf32 samples
    const long n4 = n / 2;               // float4
    v4f *x, *y; float *sink;
    hipMalloc(&x, n * 8); hipMalloc(&y, n * 8 / 5 + 4096); hipMalloc(&sink, 4);
    hipMemset(x, 1, n * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto f, const char *name, double bytes) {
        for (int i = 0; i < 3; ++i) f();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
        printf("%-34s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
    };
    for (int g : {512, 1024, 2048, 4096, 8192}) {
        char nm[64];
        snprintf(nm, 64, "read_only<4> grid=%d", g);
        timeit([&] { hipLaunchKernelGGL(read_only<4>, dim3(g), dim3(256), 0, 0, x, n4, sink); }, nm, n * 8.0);
    }
    timeit([&] { hipLaunchKernelGGL(read_only<8>, dim3(2048), dim3(256), 0, 0, x, n4, sink); }, "read_only<8> grid=2048", n * 8.0);
    timeit([&] { hipLaunchKernelGGL(read_only<1>, dim3(8192), dim3(256), 0, 0, x, n4, sink); }, "read_only<1> grid=8192", n * 8.0);
    const long nout = (n4 / 1280) * 256;
    for (int g : {1024, 2048, 4096, 16384})  {
        char nm[64];
        snprintf(nm, 64, "read5_write1 grid=%d", g);
        timeit([&] { hipLaunchKernelGGL(read5_write1, dim3(g), dim3(256), 0, 0, x, y, nout); }, nm, nout * 16.0 * 6);
    }
    return 0;
}
