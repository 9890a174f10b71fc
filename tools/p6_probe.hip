// p6_probe.hip -- would a SIX-stage in-place FFT pass pay?  Its tile is 4096 rows x C columns of float2 with rows m_lo apart;
// 4096 x 4 x 8 B = 128 KiB is what one CU can hold, so a row segment is 32 bytes.  This measures the bare traffic of such a
// pass (read the tile, write it back, no arithmetic) against the passes the library runs today (1024 x 16: 128-byte segments)
// and a plain copy, on 2^26 float2 with the transform size 2^24 and rows 4096 apart.  Sibling tiles (the 128 / (8 C) tiles that
// share cache lines) are placed on the same XCD back to back (workgroup b runs on XCD b % 8) or, for comparison, round-robin.
//   build: hipcc -O3 --offload-arch=gfx950 tools/p6_probe.hip -o tools/exp/_build_valu/p6_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

// one workgroup = one tile of ROWS x C; NT threads; thread t handles column t % C, rows (t / C) + (NT / C) * i
template <int ROWS, int C, int NT, bool SIBLINGS>
__global__ __launch_bounds__(NT) void tile_rw(float2 *data, long ntiles, unsigned m_lo, int lgN)
{
    constexpr int PER = ROWS * C / NT, RSTEP = NT / C, SIB = 16 / C; // tiles per 128-byte line
    unsigned b = blockIdx.x;
    long tile;
    if (SIBLINGS && SIB > 1) { // XCD x gets tiles in runs of SIB siblings
        const unsigned x = b & 7, slot = b >> 3;
        tile = (long)(slot % SIB) + SIB * (x + 8l * (slot / SIB));
    } else tile = b;
    if (tile >= ntiles) return;
    const unsigned tiles_per_block = m_lo / C;               // column groups of one ROWS-row block
    const long H = tile / tiles_per_block, c = tile % tiles_per_block;
    float2 *base = data + H * (long)ROWS * m_lo + c * C;
    const int col = threadIdx.x % C, r0 = threadIdx.x / C;
    float2 v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) v[i] = base[(long)m_lo * (r0 + RSTEP * i) + col];
#pragma unroll
    for (int i = 0; i < PER; ++i) { v[i].x += 1.0f; }
#pragma unroll
    for (int i = 0; i < PER; ++i) base[(long)m_lo * (r0 + RSTEP * i) + col] = v[i];
}

__global__ void copy_rw(float4 *d, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) { float4 v = d[i]; v.x += 1.0f; d[i] = v; }
}

template <typename F> static float timeit(F f, int reps)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(a);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main()
{
    const long n = 1l << 26; const int lgN = 24; const unsigned m_lo = 4096;
    float2 *d; hipMalloc(&d, n * sizeof(float2)); hipMemset(d, 0, n * sizeof(float2));
    const double gb = 2.0 * n * 8 / 1e9;
    float ms = timeit([&] { copy_rw<<<8192, 256>>>((float4 *)d, n / 2); }, 20);
    printf("plain copy (16 B per lane, grid-stride):        %.3f ms  %.2f TB/s\n", ms, gb / ms);
    ms = timeit([&] { tile_rw<1024, 16, 256, false><<<(unsigned)(n / (1024 * 16)), 256>>>(d, n / (1024 * 16), m_lo, lgN); }, 20);
    printf("1024 rows x 16 columns (today's five-stage pass): %.3f ms  %.2f TB/s\n", ms, gb / ms);
    ms = timeit([&] { tile_rw<4096, 4, 1024, true><<<(unsigned)(n / (4096 * 4)), 1024>>>(d, n / (4096 * 4), m_lo, lgN); }, 20);
    printf("4096 rows x 4 columns, siblings on one XCD:      %.3f ms  %.2f TB/s\n", ms, gb / ms);
    ms = timeit([&] { tile_rw<4096, 4, 1024, false><<<(unsigned)(n / (4096 * 4)), 1024>>>(d, n / (4096 * 4), m_lo, lgN); }, 20);
    printf("4096 rows x 4 columns, round-robin:              %.3f ms  %.2f TB/s\n", ms, gb / ms);
    ms = timeit([&] { tile_rw<4096, 8, 1024, true><<<(unsigned)(n / (4096 * 8)), 1024>>>(d, n / (4096 * 8), m_lo, lgN); }, 20);
    printf("4096 rows x 8 columns (256 KiB: does not fit a CU; for the trend), siblings: %.3f ms  %.2f TB/s\n", ms, gb / ms);
    ms = timeit([&] { tile_rw<4096, 2, 1024, true><<<(unsigned)(n / (4096 * 2)), 1024>>>(d, n / (4096 * 2), m_lo, lgN); }, 20);
    printf("4096 rows x 2 columns, siblings:                 %.3f ms  %.2f TB/s\n", ms, gb / ms);
    hipFree(d);
    return 0;
}
