// stream_probe2.hip -- what the chain's exact TRAFFIC GEOMETRY sustains on this MI355X with no arithmetic at all:
// 64-thread workgroups (one wavefront each, 8 per CU), every wavefront streams its OWN contiguous range of the input
// (10 loads of 16 B per lane = 10 KiB per step, the next step's loads issued before the current step's are consumed) and
// stores 8 KiB of output per four steps as sixteen 512-byte instructions -- against the same bytes moved by a grid-stride
// kernel whose neighbouring workgroups touch neighbouring addresses.  Random (hash-filled) and constant data.
//   build: hipcc -O3 --offload-arch=gfx950 tools/stream_probe2.hip -o tools/exp/_build_valu/stream_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void fill(unsigned *p, long n, int random)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = random ? ((h >> 8) | 0x3f000000u) & 0x3fffffffu : 0x01010101u;
    }
}

// per-wave contiguous ranges (the chain's geometry): wave w owns blocks [w*bpw, (w+1)*bpw), 4 steps of 640 float4 per block
__global__ __launch_bounds__(64, 2) void chain_geometry(const v4f *__restrict__ x, float2 *__restrict__ y, long nblocks, long bpw)
{
    const int lane = threadIdx.x;
    const long b0 = (long)blockIdx.x * bpw;
    if (b0 >= nblocks) return;
    const long b1 = b0 + bpw < nblocks ? b0 + bpw : nblocks;
    const long nsub = 4 * (b1 - b0);
    const v4f *src = x + b0 * 2560 + lane;
    v4f pre[10], cur[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) pre[i] = src[64 * i];
    v4f acc = {0, 0, 0, 0};
    for (long j = 0; j < nsub; ++j) {
#pragma unroll
        for (int i = 0; i < 10; ++i) cur[i] = pre[i];
        if (j + 1 < nsub) {
#pragma unroll
            for (int i = 0; i < 10; ++i) pre[i] = src[(j + 1) * 640 + 64 * i];
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) acc += cur[i];
        if ((j & 3) == 3) {
            float2 *dst = y + (b0 + (j >> 2)) * 1024 + lane;
#pragma unroll
            for (int k = 0; k < 16; ++k) dst[64 * k] = float2{acc.x + k, acc.y};
        }
    }
}

// same bytes, grid-stride: workgroup g of 256 threads takes block g, g + G, ... (5 KiB per wave-load group, neighbours adjacent)
__global__ __launch_bounds__(256) void grid_stride(const v4f *__restrict__ x, float2 *__restrict__ y, long nblocks)
{
    for (long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const v4f *p = x + b * 2560 + threadIdx.x;
        v4f acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 10; ++i) acc += p[256 * i];
        float2 *dst = y + b * 1024 + threadIdx.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[256 * k] = float2{acc.x + k, acc.y};
    }
}

int main()
{
    const long n = 1L << 28, nblocks = (n - 126) / 5120; // as bench.py: 52428 blocks of 5120 input samples
    v4f *x; float2 *y;
    hipMalloc(&x, n * 8); hipMalloc(&y, nblocks * 1024 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = nblocks * (5120.0 * 8 + 1024 * 8);
    for (int random = 1; random >= 0; --random) {
        hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (unsigned *)x, n * 2, random);
        auto timeit = [&](auto f, const char *name) {
            for (int i = 0; i < 200; ++i) f(); // well past the clock transient of a burst
            hipEventRecord(e0);
            for (int i = 0; i < 200; ++i) f();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 200;
            printf("%-8s %-44s %.4f ms  %.0f GB/s = %.1f %% of 8 TB/s\n", random ? "random" : "constant", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
        };
        const long waves = 2048, bpw = (nblocks + waves - 1) / waves, grid = (nblocks + bpw - 1) / bpw;
        timeit([&] { hipLaunchKernelGGL(chain_geometry, dim3((unsigned)grid), dim3(64), 19 * 1024, 0, x, y, nblocks, bpw); }, "per-wave contiguous ranges (chain geometry)");
        for (int g : {2048, 4096, 16384})  {
            char nm[64];
            snprintf(nm, 64, "grid-stride, 256-thread workgroups, grid=%d", g);
            timeit([&] { hipLaunchKernelGGL(grid_stride, dim3(g), dim3(256), 0, 0, x, y, nblocks); }, nm);
        }
    }
    return 0;
}
