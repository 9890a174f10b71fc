// chain_v5.hip -- the north-star chain kernel, generation 5: chain_v4 (wave per block run, halo
// carried in LDS, FIR results kept in registers for the native transform) with DYNAMIC work
// distribution.  Measured on v4 (tools/placement_probe.py): all waves start within 0.5 us and are spread
// evenly (8 per CU), yet wave lifetimes range 300-560 us for identical work, because the two waves
// that share a SIMD are arbitrated "oldest first"; the launch lasts as long as its slowest wave while
// the SIMDs of finished waves idle.  Here every wave draws chunks of CB consecutive blocks from one
// device-scope counter (one returning atomic per chunk, requested a sub-tile ahead), so fast waves
// simply process more chunks and all waves end together.
// Arithmetic, layouts and results are identical to v3/v4 (bit-exact with the oracle).
#include "fir_core.h"
#include "fft_wave.h"
#include "redio_internal.h"
#include <type_traits>

namespace redio {

typedef float v4f5 __attribute__((ext_vector_type(4)));

template <int N, typename F>
__device__ __forceinline__ void static_for5(F &&f)
{
    if constexpr (N > 0) {
        static_for5<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int K, int D, bool FUSED, int WPS, int CH, int CB>
__global__ __launch_bounds__(64, WPS) void chain_v5_kernel(const float2 *__restrict__ x, const float *__restrict__ taps,
                                                           const float2 *__restrict__ tw, float2 *__restrict__ out,
                                                           long nblocks, unsigned *__restrict__ queue)
{
    constexpr int R = 4;
    using G = FirGeomV<K, D, R>;
    constexpr int SUB_OUT = 64 * R;
    constexpr int SUB_NEW = SUB_OUT * D;
    constexpr int HALO = G::tile_in(SUB_OUT) - SUB_NEW;
    static_assert(HALO % 2 == 0 && SUB_NEW % 128 == 0 && 4 * SUB_OUT == 1024, "geometry");
    constexpr int HALO_V = HALO / 2;
    constexpr int NLD = SUB_NEW / 2 / 64;
    static_assert(HALO_V <= 64, "the halo moves with one instruction per lane");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v4f5 *xs4 = reinterpret_cast<v4f5 *>(smem);
    float2 *ex = reinterpret_cast<float2 *>(smem);
    const int lane = threadIdx.x;
    const v4f5 *x4 = reinterpret_cast<const v4f5 *>(x);

    auto grab = [&]() -> long { // next chunk of CB blocks; wave-uniform
        unsigned c = 0;
        if (lane == 0) c = atomicAdd(queue, 1u);
        return (long)__builtin_amdgcn_readfirstlane(c) * CB;
    };

    long b0 = grab();
    if (b0 >= nblocks) return;
    long nsub = 4 * ((b0 + CB < nblocks ? b0 + CB : nblocks) - b0);
    const v4f5 *src0 = x4 + b0 * (1024 * (long)D / 2) + lane; // float4 index of the chunk's first sample

    v4f5 pre[NLD];
    v4f5 head = v4f5{0.f, 0.f, 0.f, 0.f};
    auto fetch = [&](const v4f5 *base, long j) { // NEW samples of sub-tile j of the chunk at `base`
        const v4f5 *src = base + HALO_V + j * (SUB_NEW / 2);
        static_for5<NLD>([&](auto I) { pre[I.value] = src[64 * I.value]; });
    };
    auto park = [&]() {
        static_for5<NLD>([&](auto I) { xs4[G::lds_index(HALO + 2 * (lane + 64 * I.value)) / 2] = pre[I.value]; });
    };

    if (lane < HALO_V) xs4[G::lds_index(2 * lane) / 2] = src0[0];
    fetch(src0, 0);
    park();
    wave_lds_fence();

    float2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = make_float2(0.f, 0.f);
    long nb0 = -1; // next chunk, requested one sub-tile before it is needed
    long j = 0;
#pragma unroll 1
    for (;;) {
        const bool last = (j + 1 == nsub);
        if (j + 2 == nsub || (nsub == 1 && j == 0)) nb0 = grab(); // nsub is a multiple of 4, so j+2==nsub always occurs
        const bool next_chunk_ok = nb0 >= 0 && nb0 < nblocks;
        const bool more = !last || next_chunk_ok;
        const v4f5 *nsrc0 = x4 + nb0 * (1024 * (long)D / 2) + lane;
        if (more) {
            if (!last) {
                fetch(src0, j + 1);
            } else { // first sub-tile of the next chunk: its head comes from memory, not from this image
                if (lane < HALO_V) head = nsrc0[0];
                fetch(nsrc0, 0);
            }
        }
        float2 acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = make_float2(0.f, 0.f);
        int lf = lane;
        asm volatile("" : "+v"(lf));
        fir_lane_v<K, D, R, FUSED, CH>(xs4, lf, taps, acc);
#pragma unroll
        for (int i = 0; i < 12; ++i) a[i] = a[i + 4];
#pragma unroll
        for (int r = 0; r < R; ++r) a[12 + r] = acc[r];
        wave_lds_fence();
        v4f5 halo = head;
        if (!last && lf < HALO_V) halo = xs4[G::lds_index(SUB_NEW + 2 * lf) / 2];
        if ((j & 3) == 3) {
            wave_lds_fence();
            int ln = lane;
            asm volatile("" : "+v"(ln));
            fft1kn_wave<false>(a, ex, tw, out + (b0 + (j >> 2)) * 1024, ln);
            wave_lds_fence();
        }
        if (!more) break;
        if (lf < HALO_V) xs4[G::lds_index(2 * lf) / 2] = halo;
        park();
        wave_lds_fence();
        if (last) {
            b0 = nb0;
            nsub = 4 * ((b0 + CB < nblocks ? b0 + CB : nblocks) - b0);
            src0 = nsrc0;
            nb0 = -1;
            j = 0;
        } else {
            ++j;
        }
    }
}

static int num_cus_v5()
{
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

template <int K, int D, int WPS, int CH, int CB>
static hipError_t launch_v5_t(const float2 *x, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused,
                              unsigned *queue, hipStream_t s)
{
    using G = FirGeomV<K, D, 4>;
    constexpr int ELEMS = G::lds_elems(256) > FFT1KN_LDS ? G::lds_elems(256) : FFT1KN_LDS;
    constexpr size_t LDS_NEED = (size_t)ELEMS * sizeof(float2);
    static_assert(4 * WPS * LDS_NEED <= 160 * 1024, "4*WPS waves per CU");
    // one wave per residency slot; the LDS request caps a CU at 4*WPS waves so the slots fill evenly
    constexpr size_t LDS = (160 * 1024 / (4 * WPS)) - 480 > LDS_NEED ? (160 * 1024 / (4 * WPS)) - 480 : LDS_NEED;
    static_assert((4 * WPS + 1) * LDS > 160 * 1024, "one more wave must not fit");
    long waves = 4L * WPS * num_cus_v5();
    const long nchunks = (nblocks + CB - 1) / CB;
    if (waves > nchunks) waves = nchunks;
    hipError_t e = hipMemsetAsync(queue, 0, sizeof(unsigned), s); // the work queue starts at chunk 0 every launch
    if (e != hipSuccess) return e;
    if (fused) hipLaunchKernelGGL((chain_v5_kernel<K, D, true, WPS, CH, CB>), dim3((unsigned)waves), dim3(64), LDS, s, x, taps, tw, out, nblocks, queue);
    else hipLaunchKernelGGL((chain_v5_kernel<K, D, false, WPS, CH, CB>), dim3((unsigned)waves), dim3(64), LDS, s, x, taps, tw, out, nblocks, queue);
    return hipGetLastError();
}

hipError_t launch_chain_v5(const float2 *x, const float *taps, const float2 *tw, float2 *out, long nblocks, bool fused, int tuning,
                           unsigned *queue, hipStream_t s)
{
    switch (tuning) {
    case 1: return launch_v5_t<127, 5, 2, 8, 1>(x, taps, tw, out, nblocks, fused, queue, s);
    case 2: return launch_v5_t<127, 5, 3, 6, 2>(x, taps, tw, out, nblocks, fused, queue, s);
    case 3: return launch_v5_t<127, 5, 2, 8, 4>(x, taps, tw, out, nblocks, fused, queue, s);
    default: return launch_v5_t<127, 5, 2, 8, 2>(x, taps, tw, out, nblocks, fused, queue, s);
    }
}

} // namespace redio
