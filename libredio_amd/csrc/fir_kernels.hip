// fir_kernels.hip -- gfx950 kernels for dsputils::convolve (src/dsputils/src/dsputils.rs:30-32)
// and its decimating / complex-input extensions.
//
// Bound: HBM for D>=1 at the north-star sizes, with a VALU floor close behind (127 taps / 5 on cf32
// is 50.8 FMA per input sample).  Design: one 256-thread workgroup stages a contiguous input tile
// in LDS with coalesced 16-byte loads; each lane then produces R consecutive outputs from registers
// (fir_core.h).  Taps are wave-uniform: read through the scalar cache into SGPRs, never LDS/VGPR.
#include "fir_core.h"
#include "fir_tile.h"
#include "redio_internal.h"

namespace redio {

// ---- specialised tiled kernel ------------------------------------------------------------------
template <typename T, int K, int D, int R, bool FUSED, int NT>
__global__ __launch_bounds__(NT) void fir_tiled_kernel(const T *__restrict__ x, long n_in,
                                                       const float *__restrict__ taps,
                                                       T *__restrict__ y, long n_out, int vec_ok)
{
    using G = FirGeom<K, D, R>;
    constexpr int TILE_OUT = NT * R;
    constexpr int TILE_IN = G::tile_in(TILE_OUT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *xs = reinterpret_cast<T *>(smem);

    const long tile = blockIdx.x;
    load_tile<T, G, NT, TILE_IN>(x, n_in, tile * (long)TILE_OUT * D, xs, vec_ok != 0);
    __syncthreads();

    T acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = T{};
    fir_lane<T, K, D, R, FUSED>(xs, (int)threadIdx.x, taps, acc);

    const long o0 = tile * (long)TILE_OUT + (long)threadIdx.x * R;
    if (o0 + R <= n_out) {
        constexpr int BYTES = R * sizeof(T);
        if constexpr (BYTES % 16 == 0) {
            if (vec_ok) { // y base 16-B aligned and o0*sizeof(T) a multiple of 16
                float4 *y4 = reinterpret_cast<float4 *>(y + o0);
                const float *a = reinterpret_cast<const float *>(acc);
#pragma unroll
                for (int q = 0; q < BYTES / 16; ++q) y4[q] = make_float4(a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
                return;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) y[o0 + r] = acc[r];
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (o0 + r < n_out) y[o0 + r] = acc[r];
    }
}

typedef float fir_v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 fir_nt_ld(const float4 *p)
{
    const fir_v4f v = __builtin_nontemporal_load(reinterpret_cast<const fir_v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
// ---- chunked tiled kernel: ANY tap count, decimation from a small compile-time set ---------------
// The register blocking of fir_core.h needs compile-time tap indices; here the taps are walked in chunks of
// CH = 16 with a run-time chunk count instead: per chunk a lane reads the CH + (R-1)*D samples its R outputs
// need for these taps (statically indexed registers), the 16 taps arrive as one scalar load, and the
// multiply-adds are fully unrolled inside the chunk.  Every accumulator still receives its products in
// ascending tap order (chunks ascending, taps ascending inside a chunk), so results are bit-identical to the
// reference fold.  The K % 16 taps behind the last whole chunk run as chunks of 8, 4, 2 and 1 taps (the binary
// digits of the remainder): the same straight-line code at smaller sizes, no per-tap guard (a guarded 16-tap
// chunk compiled to a select per product and cost as much as three whole chunks).
// ALIGNED: j0 is a multiple of the lane stride R*D (whole chunks whose size is one), so the pad element count of window sample i
// is j0 / LSTR + i / LSTR with a compile-time second term: one scalar base per chunk and immediate offsets.  Round 3: before, every
// one of the window reads paid a wave-uniform division by LSTR (SQ counters of 255 taps / 10: 9.0e7 scalar against 6.7e7 vector
// instructions per launch).
template <typename T, int D, int R, bool FUSED, int CH, bool ALIGNED = false>
__device__ __forceinline__ void fir_chunk(const T *xs, int base, int j0, const float *__restrict__ taps, T (&acc)[R])
{
    constexpr int WIN = CH + (R - 1) * D, LSTR = R * D;
    constexpr bool PAD = (LSTR % 2) == 0;
    T xv[WIN];
    const int cb = base + (PAD ? j0 + j0 / LSTR : j0); // ALIGNED: the window's first sample in the image
    float h[CH];
    if constexpr (ALIGNED && sizeof(T) == 8) {
        // One ds_read_b64 per sample, spelled out: from base + immediate the compiler pairs them into ds_read2_b64, which the LDS
        // serves at half the bytes per clock (MI355X_MICROARCH.md, LDS table: 8 cycles against 2 x 2; SQ_LDS_IDX_ACTIVE of this
        // kernel rose by half when the offsets became immediates).  The reads land in order behind the taps' scalar load.
        typedef float v2f __attribute__((ext_vector_type(2)));
        v2f q[WIN];
        const unsigned a = (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void *)(xs + cb);
#pragma unroll
        for (int i = 0; i < WIN; ++i) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(q[i]) : "v"(a), "n"((PAD ? i + i / LSTR : i) * 8));
#pragma unroll
        for (int j = 0; j < CH; ++j) h[j] = taps[j0 + j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < WIN; ++i) {
            asm volatile("" : "+v"(q[i])); // uses stay behind the wait
            xv[i] = T{q[i].x, q[i].y};
        }
    } else {
#pragma unroll
        for (int i = 0; i < WIN; ++i) {
            const int m = j0 + i; // wave-uniform
            xv[i] = ALIGNED ? xs[cb + (PAD ? i + i / LSTR : i)] : xs[base + (PAD ? m + m / LSTR : m)];
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) h[j] = taps[j0 + j];
    }
#pragma unroll
    for (int i = 0; i < WIN; ++i)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = i - r * D;
            if (j >= 0 && j < CH) acc[r] = mac<FUSED>(xv[i], h[j], acc[r]);
        }
}

// A two-stage form of the whole chunks (window reads and taps of chunk c + 1 in flight into a second register set during the
// multiply-adds of chunk c, chunks of one lane stride) was built and measured slower on all ten shapes tried (255 taps / 10: 0.192 ms
// against 0.178; 101 / 3: 0.188 against 0.154): 100-190 VGPRs instead of 50-110, and the occupancy they cost hid more latency than
// the second register set did.  profiles/r03_fir_chunked.txt.
// all K taps of the R outputs of a lane: whole chunks, then the remainder's binary digits
template <typename T, int D, int R, bool FUSED, int CH = 16>
__device__ __forceinline__ void fir_chunks(const T *xs, int base, int K, const float *__restrict__ taps, T (&acc)[R])
{
    constexpr bool ALIGNED = CH % (R * D) == 0;
    const int nfull = K / CH;
    for (int c = 0; c < nfull; ++c) fir_chunk<T, D, R, FUSED, CH, ALIGNED>(xs, base, c * CH, taps, acc);
    int j0 = nfull * CH;
    const int rest = K - j0; // < CH: its binary digits as chunks of 32, 16, 8, 4, 2, 1 taps
    if (CH > 32 && (rest & 32)) { fir_chunk<T, D, R, FUSED, 32>(xs, base, j0, taps, acc); j0 += 32; }
    if (CH > 16 && (rest & 16)) { fir_chunk<T, D, R, FUSED, 16>(xs, base, j0, taps, acc); j0 += 16; }
    if (rest & 8) { fir_chunk<T, D, R, FUSED, 8>(xs, base, j0, taps, acc); j0 += 8; }
    if (rest & 4) { fir_chunk<T, D, R, FUSED, 4>(xs, base, j0, taps, acc); j0 += 4; }
    if (rest & 2) { fir_chunk<T, D, R, FUSED, 2>(xs, base, j0, taps, acc); j0 += 2; }
    if (rest & 1) fir_chunk<T, D, R, FUSED, 1>(xs, base, j0, taps, acc);
}

// Memory side (round 2; the skeleton alone -- tile in, tile out, no arithmetic -- ran at 2.9 TB/s with 8-byte loads and each lane
// storing its R consecutive outputs; 5.5 TB/s now): the tile arrives as 16-byte loads, and the outputs leave through an LDS
// transpose (lane writes its R results at a stride of R + 1 elements, conflict-free; the workgroup reads them back in output
// order) so that every store instruction writes 1 KiB of consecutive bytes.  A persistent form (a workgroup walks tiles with the
// next tile's loads in flight during the arithmetic) was built and measured slower -- at 64 taps / 1 the tile's LDS traffic
// (237 KB: window reads, tile, transpose) and its multiply-adds each fill most of the time on their own, and fewer resident waves
// hide less of it: profiles/r02_fir_generic_experiments.txt.
// Round 3, again for the shapes HBM binds (decimation >= 3): a resident grid, each workgroup walking tiles with the first rounds of the
// NEXT tile's 16-byte loads issued right after the current tile's image is complete (in flight during its arithmetic and stores):
// slower on 7 of 8 shapes (101 taps / 3: 0.197 ms against 0.154; 255 / 10: 0.185 against 0.178).  profiles/r03_fir_chunked.txt.
template <typename T, int D, int R, bool FUSED, int CHW>
__global__ __launch_bounds__(256) void fir_chunked_kernel(const T *__restrict__ x, long n_in, const float *__restrict__ taps, int K,
                                                          T *__restrict__ y, long n_out, int vec_in, int vec_out)
{
    constexpr int NT = 256, CH = 16, TILE_OUT = NT * R, LSTR = R * D, VEC = 16 / (int)sizeof(T);
    constexpr bool PAD = (LSTR % 2) == 0;
    constexpr bool BATCH = !PAD || LSTR % VEC == 0; // all VEC elements of a 16-byte load share one pad count
    // non-temporal tile loads and output stores where the samples go by once (round 3, per shape, both together: 31 taps / 2 -8 %, 255 / 10
    // -5 %, 129 / 8 -4 %, 101 / 3 and 200 / 4 unchanged); without decimation the loads alone cost 8 % at 64 taps, so D = 1 keeps the default policy
    constexpr bool STREAM = D >= 2;
    static_assert((R & (R - 1)) == 0, "R is a power of two");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T *xs = reinterpret_cast<T *>(smem);
    const int tid = threadIdx.x;
    const int kc = (K + CH - 1) / CH * CH;
    const int tile_in = (TILE_OUT - 1) * D + kc;
    const int nv = (tile_in + VEC - 1) / VEC;
    auto put = [&](int n, T v) { xs[PAD ? n + n / LSTR : n] = v; };
    auto put16 = [&](int v, const float4 &q) { // load v of the tile: VEC elements from n = v * VEC
        const int n = v * VEC;
        T *d = xs + (PAD ? n + n / LSTR : n);
        if constexpr (sizeof(T) == 8) {
            d[0] = T{q.x, q.y}; d[1] = T{q.z, q.w};
        } else {
            d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w;
        }
    };
    const long t = blockIdx.x; // the tile
    {
        const long in0 = t * TILE_OUT * D;
        if (vec_in) { // x is 16-byte aligned (in0 * sizeof(T) always is)
            const float4 *x4 = reinterpret_cast<const float4 *>(x + in0);
            int vfirst = tid;
            if (BATCH && in0 + (long)nv * VEC <= n_in) {
                // the whole tile exists (workgroup-uniform): LB loads in flight per lane, no bound checks.  Round 3: the guarded loop
                // below keeps four loads in flight and waits three times for a 43 KB tile (255 taps / 10); here every wait covers eight.
                constexpr int LB = 8;
                for (; vfirst < nv; vfirst += LB * NT) {
                    float4 q[LB];
#pragma unroll
                    for (int b = 0; b < LB; ++b) {
                        const int v = vfirst + b * NT;
                        q[b] = STREAM ? fir_nt_ld(x4 + (v < nv ? v : nv - 1)) : x4[v < nv ? v : nv - 1];
                    }
#pragma unroll
                    for (int b = 0; b < LB; ++b)
                        if (vfirst + b * NT < nv) put16(vfirst + b * NT, q[b]);
                }
            }
#pragma unroll 4
            for (int v = vfirst; v < nv; v += NT) {
                const int n = v * VEC;
                if (in0 + n + VEC <= n_in) {
                    const float4 q = x4[v];
                    if constexpr (sizeof(T) == 8) {
                        put(n, T{q.x, q.y}); put(n + 1, T{q.z, q.w});
                    } else {
                        put(n, q.x); put(n + 1, q.y); put(n + 2, q.z); put(n + 3, q.w);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        T v1{};
                        if (in0 + n + e < n_in) v1 = x[in0 + n + e];
                        put(n + e, v1); // at most VEC - 1 elements past tile_in: inside the allocation (launch_chunked)
                    }
                }
            }
        } else {
#pragma unroll 4
            for (int n = tid; n < tile_in; n += NT) {
                T v{};
                if (in0 + n < n_in) v = x[in0 + n];
                put(n, v);
            }
        }
        __syncthreads();
        T acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = T{};
        const int base = tid * (LSTR + (PAD ? 1 : 0));
        fir_chunks<T, D, R, FUSED, CHW>(xs, base, K, taps, acc);
        const long o0 = t * TILE_OUT;
        // outputs through LDS: element e of the tile sits at ys[(e / R) * (R + 1) + e % R]
        __syncthreads(); // every wave is done with the input tile
        T *ys = xs;
#pragma unroll
        for (int r = 0; r < R; ++r) ys[tid * (R + 1) + r] = acc[r];
        __syncthreads();
        const long left = n_out - o0; // outputs of this tile that exist
#pragma unroll
        for (int v = tid; v < TILE_OUT / VEC; v += NT) {
            const int e0 = v * VEC;
            T o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = ys[((e0 + e) / R) * (R + 1) + ((e0 + e) & (R - 1))];
            if (vec_out && e0 + VEC <= left) {
                const float *f = reinterpret_cast<const float *>(o);
                if constexpr (STREAM) __builtin_nontemporal_store(fir_v4f{f[0], f[1], f[2], f[3]}, reinterpret_cast<fir_v4f *>(y + o0) + v);
                else reinterpret_cast<float4 *>(y + o0)[v] = make_float4(f[0], f[1], f[2], f[3]);
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e)
                    if (e0 + e < left) y[o0 + e0 + e] = o[e];
            }
        }
    }
}

template <typename T, int D, int R, bool FUSED, int CHW = 16>
static hipError_t launch_chunked(const T *x, long n_in, const float *taps, int K, T *y, long n_out, hipStream_t s)
{
    constexpr int TILE_OUT = 256 * R, LSTR = R * D;
    const int kc = (K + 15) / 16 * 16;
    const long tile_in = (long)(TILE_OUT - 1) * D + kc + 4; // + the tail of the last 16-byte load
    long elems = tile_in + ((LSTR % 2) == 0 ? tile_in / LSTR : 0) + 2;
    if (elems < 256L * (R + 1)) elems = 256L * (R + 1); // the output transpose reuses the tile
    const size_t lds = (size_t)elems * sizeof(T);
    if (lds > 150 * 1024) return hipErrorNotSupported;
    auto kern = fir_chunked_kernel<T, D, R, FUSED, CHW>;
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    const int vec_in = (reinterpret_cast<uintptr_t>(x) & 15) == 0, vec_out = (reinterpret_cast<uintptr_t>(y) & 15) == 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)((n_out + TILE_OUT - 1) / TILE_OUT)), dim3(256), lds, s, x, n_in, taps, K, y, n_out, vec_in, vec_out);
    return hipGetLastError();
}

// ---- direct kernel: any K and D, no tile (L1/L2 serve the overlap); one output per thread -------
template <typename T, bool FUSED>
__global__ __launch_bounds__(256) void fir_direct_kernel(const T *__restrict__ x, const float *__restrict__ taps,
                                                         int K, long D, T *__restrict__ y, long n_out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_out) return;
    const T *w = x + i * D;
    T acc{};
    for (int j = 0; j < K; ++j) acc = mac<FUSED>(w[j], taps[j], acc);
    y[i] = acc;
}

template <typename T, int K, int D, int R, bool FUSED>
static hipError_t launch_tiled(const T *x, long n_in, const float *taps, T *y, long n_out, hipStream_t s)
{
    constexpr int NT = 256;
    using G = FirGeom<K, D, R>;
    constexpr int TILE_OUT = NT * R;
    constexpr size_t LDS = (size_t)G::lds_elems(TILE_OUT) * sizeof(T);
    static_assert(LDS <= 160 * 1024, "tile does not fit LDS");
    auto kern = fir_tiled_kernel<T, K, D, R, FUSED, NT>;
    if (LDS > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS);
        if (e != hipSuccess) return e;
    }
    const long ntiles = (n_out + TILE_OUT - 1) / TILE_OUT;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    hipLaunchKernelGGL(kern, dim3((unsigned)ntiles), dim3(NT), LDS, s, x, n_in, taps, y, n_out, vec_ok);
    return hipGetLastError();
}

// ---- real samples, no decimation: the pair-image tile (fir_core.h FirGeomPairs / fir_lane_pairs, round 6) ------------------------
// The tile of tile_in = TILE_OUT - 1 + K samples is staged twice, as even-start and as odd-start sample pairs, each lane then folds R
// outputs as R / 2 packed accumulators with nothing but aligned 8-byte window reads and packed multiply-adds.
template <int K, int R, bool FUSED, int NT>
__global__ __launch_bounds__(NT) void fir_pairs_kernel(const float *__restrict__ x, long n_in, const float *__restrict__ taps,
                                                       float *__restrict__ y, long n_out, int vec_ok)
{
    using G = FirGeomPairs<K, R>;
    constexpr int TILE_OUT = NT * R, TILE_IN = G::tile_in(TILE_OUT), COPY = G::copy_elems(TILE_OUT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2 *e2 = reinterpret_cast<float2 *>(smem), *o2 = e2 + COPY;
    float *ef = reinterpret_cast<float *>(e2), *of = reinterpret_cast<float *>(o2);
    const long in0 = (long)blockIdx.x * TILE_OUT;
    const int tid = threadIdx.x;
    // sample n of the tile -> E2[n / 2] half n % 2, and O2[(n - 1) / 2] half (n - 1) % 2 for n >= 1
    auto put = [&](int n, float v) {
        ef[2 * G::lds_index(n >> 1) + (n & 1)] = v;
        if (n >= 1) of[2 * G::lds_index((n - 1) >> 1) + ((n - 1) & 1)] = v;
    };
    constexpr int NV = (TILE_IN + 3) / 4;
    if (vec_ok) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x + in0);
#pragma unroll 2
        for (int v = tid; v < NV; v += NT) {
            const int n = 4 * v;
            if (in0 + n + 4 <= n_in) {
                const float4 q = fir_nt_ld(x4 + v);
                e2[G::lds_index(2 * v)] = make_float2(q.x, q.y);      // n is a multiple of 4: two whole E2 elements ...
                e2[G::lds_index(2 * v + 1)] = make_float2(q.z, q.w);
                o2[G::lds_index(2 * v)] = make_float2(q.y, q.z);      // ... one whole O2 element and two halves
                if (v > 0) of[2 * G::lds_index(2 * v - 1) + 1] = q.x;
                of[2 * G::lds_index(2 * v + 1)] = q.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) put(n + e, in0 + n + e < n_in ? x[in0 + n + e] : 0.f);
            }
        }
    } else {
        for (int n = tid; n < 4 * NV; n += NT) put(n, in0 + n < n_in ? x[in0 + n] : 0.f);
    }
    __syncthreads();
    float2 acc[R / 2];
#pragma unroll
    for (int p = 0; p < R / 2; ++p) acc[p] = make_float2(0.f, 0.f);
    fir_lane_pairs<K, R, FUSED>(e2, o2, tid, taps, acc);
    const long o0 = in0 + (long)tid * R;
    if (o0 + R <= n_out && vec_ok) {
        fir_v4f *y4 = reinterpret_cast<fir_v4f *>(y + o0);
#pragma unroll
        for (int q = 0; q < R / 4; ++q) y4[q] = fir_v4f{acc[2 * q].x, acc[2 * q].y, acc[2 * q + 1].x, acc[2 * q + 1].y};
    } else {
#pragma unroll
        for (int p = 0; p < R / 2; ++p) {
            if (o0 + 2 * p < n_out) y[o0 + 2 * p] = acc[p].x;
            if (o0 + 2 * p + 1 < n_out) y[o0 + 2 * p + 1] = acc[p].y;
        }
    }
}

template <int K, int R, int NT, bool FUSED>
static hipError_t launch_pairs(const float *x, long n_in, const float *taps, float *y, long n_out, hipStream_t s)
{
    using G = FirGeomPairs<K, R>;
    constexpr int TILE_OUT = NT * R;
    constexpr size_t LDS = 2 * (size_t)G::copy_elems(TILE_OUT) * sizeof(float2);
    static_assert(LDS <= 64 * 1024, "tile does not fit LDS");
    const long ntiles = (n_out + TILE_OUT - 1) / TILE_OUT;
    const int vec_ok = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    hipLaunchKernelGGL((fir_pairs_kernel<K, R, FUSED, NT>), dim3((unsigned)ntiles), dim3(NT), LDS, s, x, n_in, taps, y, n_out, vec_ok);
    return hipGetLastError();
}

template <typename T, bool FUSED>
static hipError_t launch_fir_t(const T *x, long n_in, const float *taps, int K, long D, T *y, long n_out, hipStream_t s)
{
    if (n_out <= 0) return hipSuccess;
#ifdef REDIO_MEASURE
    // real samples, no decimation: the pair-image tile (fir_pairs_kernel) -- bit-identical, measured SLOWER than the scalar lane program
    // below in every tile shape (profiles/r06_fir_real_forms.txt): measurement builds only, REDIO_FIR_PAIRS = R x 1000 + threads
    if constexpr (sizeof(T) == 4) {
        const char *e = measure_env("REDIO_FIR_PAIRS");
        const int sel = e ? atoi(e) : 0;
        if (K == 63 && D == 1 && sel) {
            if (sel == 16128) return launch_pairs<63, 16, 128, FUSED>(x, n_in, taps, y, n_out, s);
            if (sel == 8256) return launch_pairs<63, 8, 256, FUSED>(x, n_in, taps, y, n_out, s);
            if (sel == 16256) return launch_pairs<63, 16, 256, FUSED>(x, n_in, taps, y, n_out, s);
            if (sel == 8128) return launch_pairs<63, 8, 128, FUSED>(x, n_in, taps, y, n_out, s);
            if (sel == 16064) return launch_pairs<63, 16, 64, FUSED>(x, n_in, taps, y, n_out, s);
        }
    }
#endif
    // specialisations for the configurations BASELINE.json names (63 / 127 taps, decimate 1 / 5)
    if (K == 127 && D == 5) return launch_tiled<T, 127, 5, 4, FUSED>(x, n_in, taps, y, n_out, s);
    if (K == 127 && D == 1) return launch_tiled<T, 127, 1, 8, FUSED>(x, n_in, taps, y, n_out, s);
    if (K == 63 && D == 1) return launch_tiled<T, 63, 1, 8, FUSED>(x, n_in, taps, y, n_out, s);
    if (K == 63 && D == 5) return launch_tiled<T, 63, 5, 4, FUSED>(x, n_in, taps, y, n_out, s);
    {   // any tap count with a common decimation: the chunked tiled kernel (falls through if the tile cannot fit LDS)
        hipError_t e = hipErrorNotSupported;
        switch (D) {
        // whole-chunk sizes: a multiple of the lane stride R*D wherever the image is padded (immediate window offsets, fir_chunk)
        case 1: e = launch_chunked<T, 1, 8, FUSED>(x, n_in, taps, K, y, n_out, s); break;          // lane stride 8, chunks of 16
        case 2: e = launch_chunked<T, 2, 4, FUSED>(x, n_in, taps, K, y, n_out, s); break;          // 8, 16
        case 3: e = launch_chunked<T, 3, 4, FUSED, 24>(x, n_in, taps, K, y, n_out, s); break;      // 12, 24
        case 4: e = launch_chunked<T, 4, 4, FUSED>(x, n_in, taps, K, y, n_out, s); break;          // 16, 16
        case 5: e = launch_chunked<T, 5, 4, FUSED, 40>(x, n_in, taps, K, y, n_out, s); break;      // 20, 40
        case 8: e = launch_chunked<T, 8, 2, FUSED>(x, n_in, taps, K, y, n_out, s); break;          // 16, 16
        case 10: e = launch_chunked<T, 10, 2, FUSED, 40>(x, n_in, taps, K, y, n_out, s); break;    // 20, 40 (1 or 4 outputs per lane: 0.267 / 0.304 ms against 0.197)
        default: break;
        }
        if (e != hipErrorNotSupported) return e;
    }
    const long nb = (n_out + 255) / 256;
    hipLaunchKernelGGL((fir_direct_kernel<T, FUSED>), dim3((unsigned)nb), dim3(256), 0, s, x, taps, K, D, y, n_out);
    return hipGetLastError();
}

hipError_t launch_fir_v4(int K, int D, const float2 *x, const float *taps, float2 *y, long nblocks, bool fused, hipStream_t s); // chain_v4.hip
hipError_t launch_fir_run_real(int K, int D, const float *x, long n_in, const float *taps, float *y, long n_out, bool fused, hipStream_t s, long *done); // fir_run.hip

hipError_t launch_fir(const void *x, long n_in, const float *taps, int K, long D, void *y, long n_out,
                      bool cplx, bool fused, hipStream_t s)
{
    // the shapes the chain kernel is built for (127 / 63 taps / 5, 63 taps / 1) run on its data path (wave-private images,
    // halo carried in LDS, register prefetch): whole 1024-output blocks there, the remainder on the tiled kernel
    if (cplx && n_out >= 1024 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && !measure_env("REDIO_FIR_NO_V4")) {
        const long nblocks = n_out / 1024;
        hipError_t e = launch_fir_v4(K, (int)(D <= 5 ? D : 0), (const float2 *)x, taps, (float2 *)y, nblocks, fused, s);
        if (e == hipSuccess) {
            const long done = nblocks * 1024;
            if (done == n_out) return hipSuccess;
            x = (const float2 *)x + done * D;
            y = (float2 *)y + done;
            n_in -= done * D;
            n_out -= done;
        } else if (e != hipErrorNotSupported) {
            return e;
        }
    }
#ifdef REDIO_MEASURE
    // real samples (dsputils::convolve's own type): the shapes with a run-form instantiation (fir_run.hip) take whole sub-tiles there
    if (!cplx && D <= 16 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && measure_env("REDIO_FIR_RUN")) { // measured slower than the tiled kernels (profiles/r06_fir_real_forms.txt): measurement builds only
        long done = 0;
        hipError_t e = launch_fir_run_real(K, (int)D, (const float *)x, n_in, taps, (float *)y, n_out, fused, s, &done);
        if (e == hipSuccess) {
            if (done == n_out) return hipSuccess;
            x = (const float *)x + done * D;
            y = (float *)y + done;
            n_in -= done * D;
            n_out -= done;
        } else if (e != hipErrorNotSupported) {
            return e;
        }
    }
#endif
    if (cplx) {
        if (fused) return launch_fir_t<float2, true>((const float2 *)x, n_in, taps, K, D, (float2 *)y, n_out, s);
        return launch_fir_t<float2, false>((const float2 *)x, n_in, taps, K, D, (float2 *)y, n_out, s);
    }
    if (fused) return launch_fir_t<float, true>((const float *)x, n_in, taps, K, D, (float *)y, n_out, s);
    return launch_fir_t<float, false>((const float *)x, n_in, taps, K, D, (float *)y, n_out, s);
}

} // namespace redio
