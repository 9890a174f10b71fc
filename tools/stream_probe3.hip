// stream_probe3.hip -- where is the streaming ceiling of this MI355X pool, and what reaches it?
// the chain's best empty traffic geometry moved 5.0-5.3 TB/s (63-66 % of 8 TB/s) while
// /opt/skills/guides/MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy and 6.4-6.8 TB/s for LDS-DMA read streams.
// Every kernel here moves bytes and nothing else; each is timed over 100 launches after 100 warm-up launches on random data.
//   read    : 16-byte loads, summed (one store per thread at the end)
//   copy    : 16-byte load -> 16-byte store (bytes counted: read + written, as the guide counts a copy)
//   mix5    : the chain's mix, five 16-byte loads per 16-byte store (bytes counted: read + written)
// variants : plain | nt (non-temporal loads and stores) | glds (global_load_lds_dwordx4 into LDS, default policy and nt, aux = 2;
//            read streams only: the data is never taken out of LDS) | workgroups of 256 / 512 threads | grids of 1024 ... 16384
//   build: hipcc -O3 --offload-arch=gfx950 tools/stream_probe3.hip -o tools/exp/_build_valu/stream_probe3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float v4f __attribute__((ext_vector_type(4)));

__global__ void fill(unsigned *p, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = ((h >> 8) | 0x3f000000u) & 0x3fffffffu;
    }
}

template <bool NT> __device__ __forceinline__ v4f ld(const v4f *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(v4f *p, v4f v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// U independent 16-byte loads in flight per thread, grid-stride over n float4
template <bool NT, int U>
__global__ void k_read(const v4f *__restrict__ x, v4f *__restrict__ y, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    v4f acc = {0, 0, 0, 0};
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    for (; i < n; i += stride) acc += ld<NT>(x + i);
    y[(long)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <bool NT, int U>
__global__ void k_copy(const v4f *__restrict__ x, v4f *__restrict__ y, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld<NT>(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) st<NT>(y + i + u * stride, v[u]);
    }
    for (; i < n; i += stride) st<NT>(y + i, ld<NT>(x + i));
}

// the same copy with 8-byte lanes (float2): what the transform kernels' accesses look like to the memory system
template <int U>
__global__ void k_copy8(const float2 *__restrict__ x, float2 *__restrict__ y, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        float2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) y[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) y[i] = x[i];
}

// the tile passes' pattern: one wavefront moves a 256-row x 16-column tile of float2 (rows 2 KiB apart, 128-byte row segments,
// four segments per wave instruction, 64 loads in flight, then 64 stores), tiles handed out in dispatch order, four per workgroup
__global__ __launch_bounds__(256, 2) void k_tile(const float2 *__restrict__ x, float2 *__restrict__ y, long ntiles)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, col = lane & 15, q = lane >> 4;
    const long tile = (long)blockIdx.x * 4 + w;
    if (tile >= ntiles) return;
    const long blk = tile >> 4, c = tile & 15;
    const float2 *src = x + blk * 65536 + 16 * c + col + 4096 * q;
    float2 *dst = y + blk * 65536 + 16 * c + col + 4096 * q;
    float2 v[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) v[i] = src[256 * ((i >> 4) * 64 + (i & 15))];
#pragma unroll
    for (int i = 0; i < 64; ++i) dst[256 * ((i >> 4) * 64 + (i & 15))] = v[i];
}

// five loads per store: thread t of a workgroup-tile reads x[5*tile*T + t + T*k], k < 5, stores y[tile*T + t]
template <bool NT>
__global__ void k_mix5(const v4f *__restrict__ x, v4f *__restrict__ y, long ntiles)
{
    const int T = blockDim.x;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const v4f *p = x + tile * 5 * T + threadIdx.x;
        v4f v[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) v[k] = ld<NT>(p + k * T);
        st<NT>(y + tile * T + threadIdx.x, v[0] + v[1] + v[2] + v[3] + v[4]);
    }
}

// LDS-DMA read stream: every wave issues U 1-KiB pieces into its own LDS ring slots, waits, repeats; nothing is read back
template <int AUX, int U>
__global__ void k_glds(const v4f *__restrict__ x, v4f *__restrict__ y, long n)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char *slot = smem + wave * U * 1024;
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + i + u * stride),
                                             (__attribute__((address_space(3))) void *)(slot + u * 1024), 16, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (i == -1) y[0] = *reinterpret_cast<v4f *>(slot + lane * 16); // keeps the LDS writes observable
}

int main(int argc, char **argv)
{
    const long n = 1L << 27; // float4: 2 GiB
    v4f *x, *y;
    hipMalloc(&x, n * 16); hipMalloc(&y, n * 16);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, (unsigned *)x, n * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](auto f, double bytes, const char *name) {
        for (int i = 0; i < 100; ++i) f();
        hipEventRecord(e0);
        for (int i = 0; i < 100; ++i) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 100;
        hipError_t e = hipGetLastError();
        printf("%-72s %.4f ms  %6.0f GB/s = %.1f %% of 8 TB/s%s\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0, e == hipSuccess ? "" : "  (launch error)");
        fflush(stdout);
    };
    char nm[128];
    const double rd = n * 16.0;
#define RUN(KERN, T, G, BYTES, LABEL) { snprintf(nm, 128, "%s, %d threads, grid %d", LABEL, T, G); timeit([&] { hipLaunchKernelGGL(KERN, dim3(G), dim3(T), 0, 0, x, y, n); }, BYTES, nm); }
    for (int T : {256, 512}) for (int G : {1024, 2048, 4096, 16384}) {
        RUN((k_read<false, 4>), T, G, rd, "read  plain, 4 loads in flight");
        RUN((k_read<true, 4>), T, G, rd, "read  nt,    4 loads in flight");
    }
    RUN((k_read<false, 8>), 256, 2048, rd, "read  plain, 8 loads in flight");
    RUN((k_read<false, 8>), 256, 4096, rd, "read  plain, 8 loads in flight");
    RUN((k_read<true, 8>), 256, 4096, rd, "read  nt,    8 loads in flight");
    for (int T : {256, 512}) for (int G : {2048, 4096, 16384, 65536}) {
        RUN((k_copy<false, 4>), T, G, 2 * rd, "copy  plain, 4 in flight");
        RUN((k_copy<true, 4>), T, G, 2 * rd, "copy  nt,    4 in flight");
    }
    RUN((k_copy<false, 1>), 256, 131072, 2 * rd, "copy  plain, 1 in flight");
    RUN((k_copy<false, 1>), 256, 524288, 2 * rd, "copy  plain, one float4 per thread");
    RUN((k_copy<true, 1>), 256, 524288, 2 * rd, "copy  nt,    one float4 per thread");
    RUN((k_copy<false, 8>), 256, 4096, 2 * rd, "copy  plain, 8 in flight");
    {
        const long n2 = n * 2; // float2 elements of the same buffers
        for (int G : {4096, 16384, 65536}) {
            snprintf(nm, 128, "copy  8-byte lanes, 4 in flight, 256 threads, grid %d", G);
            timeit([&] { hipLaunchKernelGGL((k_copy8<4>), dim3(G), dim3(256), 0, 0, (const float2 *)x, (float2 *)y, n2); }, 2 * rd, nm);
        }
        snprintf(nm, 128, "copy  8-byte lanes, one float2 per thread, 256 threads, grid %ld", n2 / 256);
        timeit([&] { hipLaunchKernelGGL((k_copy8<1>), dim3((unsigned)(n2 / 256)), dim3(256), 0, 0, (const float2 *)x, (float2 *)y, n2); }, 2 * rd, nm);
        const long ntiles = n2 / 4096; // 256 x 16 tiles of 65536-point blocks
        snprintf(nm, 128, "copy  tile pattern (256 rows x 128 B, 8-byte lanes, 64 in flight), grid %ld", ntiles / 4);
        timeit([&] { hipLaunchKernelGGL(k_tile, dim3((unsigned)(ntiles / 4)), dim3(256), 0, 0, (const float2 *)x, (float2 *)y, ntiles); }, 2 * rd, nm);
        const long chunk_tiles = 2048; // one 64 MiB chunk per launch, as the overlap-save passes run
        snprintf(nm, 128, "copy  tile pattern, ONE 64 MiB chunk per launch (2048 tiles), x%ld launches", ntiles / chunk_tiles);
        timeit([&] { for (long t0 = 0; t0 < ntiles; t0 += chunk_tiles) hipLaunchKernelGGL(k_tile, dim3((unsigned)(chunk_tiles / 4)), dim3(256), 0, 0, (const float2 *)x + t0 * 4096, (float2 *)y + t0 * 4096, chunk_tiles); }, 2 * rd, nm);
    }
    {
        for (int T : {256, 512}) for (int G : {2048, 4096, 16384}) {
            const long ntiles = n / (5L * T);
            const double bytes = ntiles * (double)T * 16.0 * 6;
            snprintf(nm, 128, "mix5  plain, %d threads, grid %d", T, G);
            timeit([&] { hipLaunchKernelGGL((k_mix5<false>), dim3(G), dim3(T), 0, 0, x, y, ntiles); }, bytes, nm);
            snprintf(nm, 128, "mix5  nt,    %d threads, grid %d", T, G);
            timeit([&] { hipLaunchKernelGGL((k_mix5<true>), dim3(G), dim3(T), 0, 0, x, y, ntiles); }, bytes, nm);
        }
    }
#define RUNG(AUX, U, T, G, LABEL) { snprintf(nm, 128, "%s, %d pieces in flight per wave, %d threads, grid %d", LABEL, U, T, G); \
        timeit([&] { hipLaunchKernelGGL((k_glds<AUX, U>), dim3(G), dim3(T), (T / 64) * U * 1024, 0, x, y, n); }, rd, nm); }
    for (int G : {1024, 2048, 4096}) {
        RUNG(0, 4, 256, G, "read  LDS-DMA default");
        RUNG(2, 4, 256, G, "read  LDS-DMA nt     ");
        RUNG(0, 8, 256, G, "read  LDS-DMA default");
        RUNG(2, 8, 256, G, "read  LDS-DMA nt     ");
    }
    RUNG(0, 8, 512, 1024, "read  LDS-DMA default");
    RUNG(2, 8, 512, 1024, "read  LDS-DMA nt     ");
    RUNG(2, 16, 256, 1024, "read  LDS-DMA nt     ");
    return 0;
}
