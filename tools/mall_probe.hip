// mall_probe.hip -- does an intermediate that is written and read back within the Infinity Cache's reach (256 MiB) cost HBM time?
// The multi-pass transforms and the overlap-save passes write an intermediate and read it back in the next pass; their work buffers could
// be sized to stay on the die.  Two measurements, bytes and nothing else, random data, 16 bytes per lane:
//   (1) launches : for every chunk k of S bytes  copy X_k -> A  then  copy A -> OUT_k   (A reused: "pingpong"; A_k distinct: "fresh")
//                  moved bytes = 4 S per chunk; S from 16 MiB to 1 GiB; X and OUT are 2 GiB each
//   (2) one launch: a persistent workgroup copies a tile X -> its own slot of A, then slot -> OUT; slot sizes 32 KiB ... 1 MiB
//                  (A = workgroups * slot: 64 MiB ... 2 GiB), against the same loop with the slot write and read left out ("direct")
//   build: hipcc -O3 --offload-arch=gfx950 tools/mall_probe.hip -o tools/exp/_build_valu/mall_probe
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

template <bool NTL, bool NTS> __global__ void k_copy(const v4f *__restrict__ x, v4f *__restrict__ y, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        v4f v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = NTL ? __builtin_nontemporal_load(x + i + u * stride) : x[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) { if (NTS) __builtin_nontemporal_store(v[u], y + i + u * stride); else y[i + u * stride] = v[u]; }
    }
    for (; i < n; i += stride) y[i] = x[i];
}

// persistent: workgroup w handles tiles w, w + G, ...; a tile is `slot` float4; VIA: through the workgroup's slot of A
template <bool VIA, bool NTA> __global__ void __launch_bounds__(256) k_tile(const v4f *__restrict__ x, v4f *__restrict__ a, v4f *__restrict__ y,
                                                                           long ntiles, long slot)
{
    v4f *mine = a + (long)blockIdx.x * slot;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const v4f *src = x + t * slot;
        v4f *dst = y + t * slot;
        if (VIA) {
            for (long i = threadIdx.x; i < slot; i += 1024) {
                v4f v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + i + 256 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) { if (NTA) __builtin_nontemporal_store(v[u], mine + i + 256 * u); else mine[i + 256 * u] = v[u]; }
            }
            __syncthreads();
            for (long i = threadIdx.x; i < slot; i += 1024) {
                v4f v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = NTA ? __builtin_nontemporal_load(mine + i + 256 * ((u + 1) & 3)) : mine[i + 256 * ((u + 1) & 3)];
#pragma unroll
                for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], dst + i + 256 * ((u + 1) & 3));
            }
            __syncthreads();
        } else {
            for (long i = threadIdx.x; i < slot; i += 1024) {
                v4f v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + i + 256 * u);
#pragma unroll
                for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], dst + i + 256 * u);
            }
        }
    }
}

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(_e), __LINE__); return 1; } } while (0)

int main()
{
    const size_t G2 = (size_t)2 << 30;
    v4f *X, *OUT, *A;
    CK(hipMalloc(&X, G2)); CK(hipMalloc(&OUT, G2)); CK(hipMalloc(&A, G2));
    fill<<<4096, 256>>>((unsigned *)X, G2 / 4); fill<<<4096, 256>>>((unsigned *)A, G2 / 4); fill<<<4096, 256>>>((unsigned *)OUT, G2 / 4);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // clock ramp
    for (int r = 0; r < 40; ++r) k_copy<false, false><<<8192, 256>>>(X, OUT, (long)(G2 / 16));
    CK(hipDeviceSynchronize());
    printf("# (1) launches per chunk: X_k -> A, A -> OUT_k; TB/s of moved bytes (4 S per chunk), 2 GiB of X per repetition, median of 5\n");
    for (int policy = 0; policy < 3; ++policy) { // 0 plain, 1 nt on X loads / OUT stores only, 2 nt everywhere
        for (int fresh = 0; fresh < 2; ++fresh) {
            for (size_t S = (size_t)16 << 20; S <= ((size_t)1 << 30); S <<= 1) {
                const long n4 = (long)(S / 16);
                const int grid = (int)((n4 / 256 / 4) < 8192 ? (n4 / 256 / 4) : 8192);
                float best[5];
                for (int rep = 0; rep < 5; ++rep) {
                    CK(hipEventRecord(e0));
                    for (size_t k = 0; k < G2 / S; ++k) {
                        v4f *a = fresh ? A + k * (S / 16) : A;
                        if (policy == 0) { k_copy<false, false><<<grid, 256>>>(X + k * (S / 16), a, n4); k_copy<false, false><<<grid, 256>>>(a, OUT + k * (S / 16), n4); }
                        else if (policy == 1) { k_copy<true, false><<<grid, 256>>>(X + k * (S / 16), a, n4); k_copy<false, true><<<grid, 256>>>(a, OUT + k * (S / 16), n4); }
                        else { k_copy<true, true><<<grid, 256>>>(X + k * (S / 16), a, n4); k_copy<true, true><<<grid, 256>>>(a, OUT + k * (S / 16), n4); }
                    }
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&best[rep], e0, e1));
                }
                for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (best[j] < best[i]) { float t = best[i]; best[i] = best[j]; best[j] = t; }
                printf("policy %d %-8s S %5zu MiB : %.3f ms  %.2f TB/s moved\n", policy, fresh ? "fresh" : "pingpong", S >> 20, best[2], 4.0 * G2 / best[2] * 1e-9);
            }
        }
    }
    printf("# (2) one persistent launch, 2048 workgroups of 256: tile X -> own slot of A -> OUT; TB/s of X + OUT bytes only (4 GiB), median of 5\n");
    for (int mode = 0; mode < 3; ++mode) { // 0 direct, 1 via plain, 2 via nt
        for (long slotB = 32 << 10; slotB <= (1 << 20); slotB <<= 1) {
            const long slot = slotB / 16, ntiles = (long)(G2 / slotB);
            float best[5];
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipEventRecord(e0));
                if (mode == 0) k_tile<false, false><<<2048, 256>>>(X, A, OUT, ntiles, slot);
                else if (mode == 1) k_tile<true, false><<<2048, 256>>>(X, A, OUT, ntiles, slot);
                else k_tile<true, true><<<2048, 256>>>(X, A, OUT, ntiles, slot);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&best[rep], e0, e1));
            }
            for (int i = 0; i < 5; ++i) for (int j = i + 1; j < 5; ++j) if (best[j] < best[i]) { float t = best[i]; best[i] = best[j]; best[j] = t; }
            printf("%-9s slot %5ld KiB (A in use %5ld MiB) : %.3f ms  %.2f TB/s of X + OUT\n", mode == 0 ? "direct" : mode == 1 ? "via" : "via-nt", slotB >> 10,
                   (2048 * slotB) >> 20, best[2], 2.0 * G2 / best[2] * 1e-9);
        }
    }
    return 0;
}
