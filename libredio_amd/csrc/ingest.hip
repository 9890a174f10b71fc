// ingest.hip -- the steps immediately upstream of the hot path in the one graph LibRedio ships
// (src/ratpak.rs:60-76) -- SURVEY.md 8f rank 1, the "bit-exact slicing/indexing" clause:
//   rtlsdr::data_to_samples  src/rtlsdr/src/rtlsdr.rs:159-162   u8 IQ pairs -> cf32, i as f32/127.0 - 1.0
//   |x| map                  src/ratpak.rs:64-68                 Complex::norm = hypotf(re, im)
//   bitfount::trigger        src/bitfount/src/bitfount.rs:36-85  per-block sum (sequential f32), adaptive
//                                                                threshold, collect blocks while triggered
//   bitfount::discretize     src/bitfount/src/bitfount.rs:87-96  max = fold(0.0, f32::max); (x > max/2)
// Every result is bit-identical to the oracle (oracle_bits.c): IEEE divide/subtract, hypotf evaluated
// as (float)sqrt((double)re*re + (double)im*im) (= glibc's hypotf on the whole u8 domain and on random
// data), the block sum accumulated in sample order by one lane per block, max by an order-free
// wave-shuffle + atomic reduction.  All kernels are HBM-bound elementwise / reduction work.
#include "../../include/redio.h"
#include "redio_internal.h"
#include <new>
#include <string.h>
#include <vector>

namespace redio {

__device__ __forceinline__ float norm_f32(float re, float im)
{
    return (float)sqrt((double)re * (double)re + (double)im * (double)im);
}

// two samples per lane step: one 4-byte load, one 16-byte store (1 KiB contiguous per wave instruction)
__global__ __launch_bounds__(256) void data_to_samples_kernel(const uint8_t *__restrict__ d, float2 *__restrict__ out, long nsamp)
{
    const long stride = (long)gridDim.x * blockDim.x, tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool aligned = ((uintptr_t)d & 3) == 0 && ((uintptr_t)out & 15) == 0;
    const long n2 = aligned ? nsamp / 2 : 0;
    for (long q = tid; q < n2; q += stride) {
        const unsigned w = reinterpret_cast<const unsigned *>(d)[q];
        reinterpret_cast<float4 *>(out)[q] = make_float4(i2f(w & 255u), i2f((w >> 8) & 255u), i2f((w >> 16) & 255u), i2f(w >> 24));
    }
    for (long i = 2 * n2 + tid; i < nsamp; i += stride) out[i] = make_float2(i2f(d[2 * i]), i2f(d[2 * i + 1]));
}

__global__ __launch_bounds__(256) void norm_kernel(const float2 *__restrict__ x, float *__restrict__ out, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = norm_f32(x[i].x, x[i].y);
}

// fused: u8 IQ -> magnitude, 8 bytes (4 samples) in and 16 bytes out per lane per step
__global__ __launch_bounds__(256) void ingest_mag_kernel(const uint8_t *__restrict__ d, float *__restrict__ mag, long nsamp)
{
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = nsamp / 4;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const uint2 w = reinterpret_cast<const uint2 *>(d)[q];
        float4 m;
        m.x = norm_f32(i2f(w.x & 255u), i2f((w.x >> 8) & 255u));
        m.y = norm_f32(i2f((w.x >> 16) & 255u), i2f(w.x >> 24));
        m.z = norm_f32(i2f(w.y & 255u), i2f((w.y >> 8) & 255u));
        m.w = norm_f32(i2f((w.y >> 16) & 255u), i2f(w.y >> 24));
        reinterpret_cast<float4 *>(mag)[q] = m;
    }
    if (blockIdx.x == 0 && threadIdx.x < (nsamp & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        mag[i] = norm_f32(i2f(d[2 * i]), i2f(d[2 * i + 1]));
    }
}

// The same through a table: |x| of a byte pair depends only on the unordered pair (the double-precision sum of
// the two squares commutes), so 32896 floats cover the whole input domain.  A persistent 1024-thread workgroup
// per CU builds the table in LDS with the expression above (exact by construction), then every sample is one
// LDS lookup; the f64 square root leaves the streaming loop.
constexpr int MAG_LUT = 256 * 257 / 2;
__global__ __launch_bounds__(1024) void ingest_mag_lut_kernel(const uint8_t *__restrict__ d, float *__restrict__ mag, long nsamp)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *lut = reinterpret_cast<float *>(smem);
    for (int p = threadIdx.x; p < 65536; p += 1024) {
        const int a = p >> 8, b = p & 255;
        if (b <= a) lut[a * (a + 1) / 2 + b] = norm_f32(i2f((unsigned)a), i2f((unsigned)b));
    }
    __syncthreads();
    auto look = [&](unsigned re, unsigned im) {
        const unsigned hi = re > im ? re : im, lo = re > im ? im : re;
        return lut[hi * (hi + 1) / 2 + lo];
    };
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = nsamp / 4;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += stride) {
        const uint2 w = reinterpret_cast<const uint2 *>(d)[q];
        float4 m;
        m.x = look(w.x & 255u, (w.x >> 8) & 255u);
        m.y = look((w.x >> 16) & 255u, w.x >> 24);
        m.z = look(w.y & 255u, (w.y >> 8) & 255u);
        m.w = look((w.y >> 16) & 255u, w.y >> 24);
        typedef float mag_v4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(mag_v4{m.x, m.y, m.z, m.w}, reinterpret_cast<mag_v4 *>(mag) + q);
    }
    if (blockIdx.x == 0 && threadIdx.x < (nsamp & 3)) {
        const long i = n4 * 4 + threadIdx.x;
        mag[i] = look(d[2 * i], d[2 * i + 1]);
    }
}

// per-block sums in sample order (bitfount.rs:48): one lane per block; a wave stages 64 blocks'
// chunks through LDS so that global reads stay coalesced (row padded by one float: lane stride 65).
// Any block length and alignment.
__global__ __launch_bounds__(64) void block_sum_generic_kernel(const float *__restrict__ x, float *__restrict__ sums, long nblocks, int block)
{
    __shared__ float tile[64 * 65];
    const int lane = threadIdx.x;
    const long b0 = (long)blockIdx.x * 64;
    float s = 0.0f;
    for (int c0 = 0; c0 < block; c0 += 64) {
        // rows = blocks b0..b0+63, columns = samples c0..c0+63 of each block
        for (int r = 0; r < 64; ++r) {
            const long b = b0 + r;
            const int c = c0 + lane;
            tile[r * 65 + lane] = (b < nblocks && c < block) ? x[b * block + c] : 0.0f;
        }
        __syncthreads();
        const int lim = (block - c0 < 64) ? block - c0 : 64;
        for (int c = 0; c < lim; ++c) s = add_rn(s, tile[lane * 65 + c]);
        __syncthreads();
    }
    if (b0 + lane < nblocks) sums[b0 + lane] = s;
}

// The same with 16-byte accesses end to end (block % 64 == 0, x 16-byte aligned -- the reference's 512):
// per 64-sample chunk a wave issues sixteen 1 KiB loads (4 rows x 256 B each) before it stores any of
// them to LDS, then every lane walks its own row with ds_read_b128 (row stride 68 floats: the 16-lane
// groups of a b128 access cover all 64 banks) and adds the samples in order.
__global__ __launch_bounds__(64) void block_sum_kernel(const float *__restrict__ x, float *__restrict__ sums, long nblocks, int block)
{
    constexpr int LD = 68;
    __shared__ __attribute__((aligned(16))) float tile[64 * LD];
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x;
    const long b0 = (long)blockIdx.x * 64;
    const int rsub = lane >> 4, c4 = 4 * (lane & 15);
    float s = 0.0f;
    for (int c0 = 0; c0 < block; c0 += 64) {
        v4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            long b = b0 + 4 * k + rsub;
            if (b >= nblocks) b = nblocks - 1; // read a valid row; its lane never stores a result
            v[k] = __builtin_nontemporal_load(reinterpret_cast<const v4 *>(x + b * block + c0 + c4));
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) *reinterpret_cast<v4 *>(&tile[(4 * k + rsub) * LD + c4]) = v[k];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const v4 t = *reinterpret_cast<const v4 *>(&tile[lane * LD + 4 * q]);
            s = add_rn(s, t.x); s = add_rn(s, t.y); s = add_rn(s, t.z); s = add_rn(s, t.w);
        }
        __syncthreads();
    }
    if (b0 + lane < nblocks) sums[b0 + lane] = s;
}

// discretize pass 1: max = fold(0.0, f32::max): NaN operands are ignored, negatives never win, so the
// result is a non-negative float whose bit pattern orders like an unsigned integer
__global__ __launch_bounds__(256) void max_kernel(const float *__restrict__ x, long n, unsigned *max_bits)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    float m = 0.0f;
    const long stride = (long)gridDim.x * blockDim.x, tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long head = (n < 4 || ((uintptr_t)x & 15) == 0) ? 0 : (16 - ((uintptr_t)x & 15)) / 4; // scalars before the first aligned 16 bytes
    const long n4 = n >= head ? (n - head) / 4 : 0;
    const v4 *x4 = reinterpret_cast<const v4 *>(x + head);
    for (long i = tid; i < n4; i += stride) {
        const v4 v = __builtin_nontemporal_load(x4 + i); // the slicer reads the stream again, but only after all of it has gone by
        if (v.x > m) m = v.x; // false for NaN
        if (v.y > m) m = v.y;
        if (v.z > m) m = v.z;
        if (v.w > m) m = v.w;
    }
    for (long i = tid; i < n - 4 * n4; i += stride) { // head and tail scalars
        const float v = x[i < head ? i : 4 * n4 + i];
        if (v > m) m = v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off);
        if (o > m) m = o;
    }
    // one atomic per workgroup (same-address atomics serialise in L2)
    __shared__ float wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (wmax[w] > m) m = wmax[w];
        atomicMax(max_bits, __float_as_uint(m));
    }
}

// discretize pass 2: four samples per lane step -- one 16-byte load (1 KiB contiguous per wave instruction), one 4-byte store of 0/1 bytes;
// four steps per thread, workgroups in dispatch order (round 3: the form with four 16-byte loads at a 64-byte lane stride and one 16-byte
// store per step ran at 68.7 % of 8 TB/s for max + slice together)
__global__ __launch_bounds__(256) void slice_kernel(const float *__restrict__ x, long n, const unsigned *max_bits, uint8_t *__restrict__ out)
{
    typedef float v4 __attribute__((ext_vector_type(4)));
    const float thr = __uint_as_float(*max_bits) / 2.0f;
    const long stride = (long)gridDim.x * blockDim.x, tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool aligned = ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 3) == 0;
    const long n4 = aligned ? n / 4 : 0;
    const v4 *x4 = reinterpret_cast<const v4 *>(x);
    unsigned *o1 = reinterpret_cast<unsigned *>(out);
    for (long i0 = (long)blockIdx.x * 1024 + threadIdx.x; i0 < n4; i0 += stride * 4) {
        v4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long i = i0 + 256 * q;
            v[q] = x4[i < n4 ? i : n4 - 1];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long i = i0 + 256 * q;
            if (i < n4) o1[i] = (v[q].x > thr ? 1u : 0u) | (v[q].y > thr ? 0x100u : 0u) | (v[q].z > thr ? 0x10000u : 0u) | (v[q].w > thr ? 0x1000000u : 0u);
        }
    }
    for (long i = 4 * n4 + tid; i < n; i += stride) out[i] = x[i] > thr ? 1 : 0;
}

// gather whole blocks (trigger's push_all of triggered blocks): seg = (src_block, dst_offset)
__global__ __launch_bounds__(256) void gather_blocks_kernel(const float *__restrict__ x, const long *__restrict__ src_block,
                                                            const long *__restrict__ dst_off, int block, float *__restrict__ out)
{
    const long b = blockIdx.x;
    const float *s = x + src_block[b] * block;
    float *d = out + dst_off[b];
    for (int i = threadIdx.x; i < block; i += blockDim.x) d[i] = s[i];
}

// `per` items per workgroup, at most `cap` workgroups (the kernels walk the rest with a grid stride)
// (a launch of 2^32 threads or more is rejected: callers that want dispatch-order grids pass cap = 2^24 - 1 workgroups of 256 and
// rely on their kernel's grid-stride loop beyond it)
static unsigned grid_for(long n, int per = 256, long cap = 16384)
{
    long g = (n + per - 1) / per;
    return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g));
}

} // namespace redio
using namespace redio;

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? REDIO_OK : REDIO_ERR_HIP_BASE - (int)e; }
#define IN_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return hip_rc(_e); } while (0)

extern "C" int redio_data_to_samples(const void *d_bytes, size_t nbytes, void *d_out, void *stream)
{
    if (nbytes & 1) return REDIO_ERR_ASSERT; // chunks(2) then i[1]: index panic (rtlsdr.rs:161)
    if (nbytes == 0) return REDIO_OK;
    if (!d_bytes || !d_out) return REDIO_ERR_ARG;
    const long ns = (long)(nbytes / 2);
    // four 4-byte loads (16-byte stores) per thread, workgroups in dispatch order: swept over 1 ... 128 per thread at 2^28 samples (round 3), 0.466 ms
    // against 0.557 with the grid capped at 16384 workgroups and 0.607 with one load per thread
    hipLaunchKernelGGL(data_to_samples_kernel, dim3(grid_for(ns / 2, 1024, 0xffffffL)), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_bytes, (float2 *)d_out, ns);
    return hip_rc(hipGetLastError());
}

extern "C" int redio_norm_c32(const void *d_in, size_t n, void *d_out, void *stream)
{
    if (n == 0) return REDIO_OK;
    if (!d_in || !d_out) return REDIO_ERR_ARG;
    hipLaunchKernelGGL(norm_kernel, dim3(grid_for((long)n)), dim3(256), 0, (hipStream_t)stream, (const float2 *)d_in, (float *)d_out, (long)n);
    return hip_rc(hipGetLastError());
}

extern "C" int redio_ingest_u8_mag(const void *d_bytes, size_t nbytes, void *d_mag, void *stream)
{
    if (nbytes & 1) return REDIO_ERR_ASSERT;
    if (nbytes == 0) return REDIO_OK;
    if (!d_bytes || !d_mag) return REDIO_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(d_bytes) & 7) || (reinterpret_cast<uintptr_t>(d_mag) & 15)) return REDIO_ERR_ARG;
    const long ns = (long)(nbytes / 2);
    if (ns >= (1L << 22)) { // long streams: the table pays for itself (one persistent workgroup per CU)
        static int cus = 0;
        if (!cus) {
            int dev = 0;
            hipDeviceProp_t prop;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                      ? prop.multiProcessorCount : 256;
        }
        const size_t lds = MAG_LUT * sizeof(float);
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(ingest_mag_lut_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return hip_rc(e);
        hipLaunchKernelGGL(ingest_mag_lut_kernel, dim3((unsigned)cus), dim3(1024), lds, (hipStream_t)stream, (const uint8_t *)d_bytes, (float *)d_mag, ns);
        return hip_rc(hipGetLastError());
    }
    hipLaunchKernelGGL(ingest_mag_kernel, dim3(grid_for(ns / 4 + 1)), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)d_bytes, (float *)d_mag, ns);
    return hip_rc(hipGetLastError());
}

extern "C" int redio_block_sums(const void *d_in, size_t nblocks, size_t block, void *d_sums, void *stream)
{
    if (nblocks == 0) return REDIO_OK;
    if (!d_in || !d_sums || block == 0 || block > (1u << 30)) return REDIO_ERR_ARG;
    const bool vec = block % 64 == 0 && ((uintptr_t)d_in & 15) == 0;
    hipLaunchKernelGGL(vec ? block_sum_kernel : block_sum_generic_kernel, dim3((unsigned)((nblocks + 63) / 64)), dim3(64), 0,
                       (hipStream_t)stream, (const float *)d_in, (float *)d_sums, (long)nblocks, (int)block);
    return hip_rc(hipGetLastError());
}

extern "C" int redio_discretize(const void *d_in, size_t n, void *d_out_u8, void *d_scratch_u32, void *stream)
{
    if (n == 0) return REDIO_OK;
    if (!d_in || !d_out_u8 || !d_scratch_u32) return REDIO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    IN_TRY(hipMemsetAsync(d_scratch_u32, 0, sizeof(unsigned), st)); // fold starts at 0.0
    hipLaunchKernelGGL(max_kernel, dim3(grid_for((long)n, 1024) > 2048 ? 2048 : grid_for((long)n, 1024)), dim3(256), 0, st, (const float *)d_in, (long)n, (unsigned *)d_scratch_u32);
    hipLaunchKernelGGL(slice_kernel, dim3(grid_for((long)n / 4, 1024, 0xffffffL)), dim3(256), 0, st, (const float *)d_in, (long)n, (const unsigned *)d_scratch_u32,
                       (uint8_t *)d_out_u8);
    return hip_rc(hipGetLastError());
}

// ---------------------------------------------------------------- trigger (bitfount.rs:36-85)
struct redio_trigger {
    int device;
    long trigger;       // :42
    float threshold;    // :44
    // sample_buffer (:43) lives on the device; it starts as [0.0]
    float *d_buf;
    size_t len, cap;
    float *d_sums; size_t sums_cap;
    long *d_src, *d_dst; size_t seg_cap;
};

extern "C" int redio_trigger_create(redio_trigger **h)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return REDIO_ERR_NO_DEVICE;
    redio_trigger *t = new (std::nothrow) redio_trigger();
    if (!t) return REDIO_ERR_NOMEM;
    memset(t, 0, sizeof(*t));
    t->device = dev;
    t->cap = 1 + 64 * 512;
    hipError_t e = hipMalloc((void **)&t->d_buf, t->cap * sizeof(float));
    if (e == hipSuccess) e = hipMemset(t->d_buf, 0, sizeof(float)); // vec!(0.0)
    if (e != hipSuccess) { delete t; return hip_rc(e); }
    t->len = 1;
    *h = t;
    return REDIO_OK;
}

extern "C" int redio_trigger_destroy(redio_trigger *t)
{
    if (!t) return REDIO_OK;
    hipFree(t->d_buf); hipFree(t->d_sums); hipFree(t->d_src); hipFree(t->d_dst);
    delete t;
    return REDIO_OK;
}

static int trig_reserve(redio_trigger *t, size_t need, hipStream_t st)
{
    if (need <= t->cap) return REDIO_OK;
    size_t nc = t->cap;
    while (nc < need) nc *= 2;
    float *nb = nullptr;
    IN_TRY(hipMalloc((void **)&nb, nc * sizeof(float)));
    IN_TRY(hipMemcpyAsync(nb, t->d_buf, t->len * sizeof(float), hipMemcpyDeviceToDevice, st));
    IN_TRY(hipStreamSynchronize(st));
    hipFree(t->d_buf);
    t->d_buf = nb;
    t->cap = nc;
    return REDIO_OK;
}

// Feeds nblocks blocks of `block` magnitudes (device).  Emitted buffers are appended to d_out (device,
// capacity out_cap floats) and their lengths to lens (host).  Synchronous: the adaptive threshold is a
// scalar recurrence over the block sums and runs on the host between two launches.
extern "C" int redio_trigger_feed(redio_trigger *t, const void *d_blocks, size_t nblocks, size_t block, void *d_out, size_t out_cap,
                                  size_t *lens, size_t lens_cap, size_t *nemit_out, size_t *total_out, void *stream)
{
    if (nemit_out) *nemit_out = 0;
    if (total_out) *total_out = 0;
    if (!t) return REDIO_ERR_ARG;
    if (nblocks == 0) return REDIO_OK;
    if (!d_blocks || block == 0) return REDIO_ERR_ARG;
    IN_TRY(hipSetDevice(t->device));
    hipStream_t st = (hipStream_t)stream;
    if (nblocks > t->sums_cap) {
        hipFree(t->d_sums); t->d_sums = nullptr; t->sums_cap = 0;
        IN_TRY(hipMalloc((void **)&t->d_sums, nblocks * sizeof(float)));
        t->sums_cap = nblocks;
    }
    int rc = redio_block_sums(d_blocks, nblocks, block, t->d_sums, st);
    if (rc) return rc;
    std::vector<float> s(nblocks);
    IN_TRY(hipMemcpyAsync(s.data(), t->d_sums, nblocks * sizeof(float), hipMemcpyDeviceToHost, st));
    IN_TRY(hipStreamSynchronize(st));

    const long trigger_duration = 50;  // :41
    const size_t block_size = 512;     // :38 (only in the OOM bound)
    size_t nemit = 0, total = 0;
    { // dry run of the scalar recurrence on a copy of the state: what this call would emit.  A result that does not
      // fit is refused BEFORE anything is consumed (the sums are recomputed by the retry), as redio_rle_feed does.
        long trig = t->trigger; float thr = t->threshold; size_t len = t->len, ne = 0, tot = 0;
        for (size_t b = 0; b < nblocks; ++b) {
            trig -= 1;
            if (len > 1000 * (size_t)trigger_duration * block_size) len = 1;
            if (thr == 0.0f) thr = s[b];
            if (trig < 0) { thr += s[b] / 1000.0f; thr -= thr * 0.002f; }
            if (s[b] > thr * 4.0f) trig = trigger_duration;
            if (trig > 1) len += block;
            if (trig == 0) { tot += len; ++ne; len = 0; }
        }
        if (ne > lens_cap || tot > out_cap || (ne && (!d_out || !lens))) {
            if (nemit_out) *nemit_out = ne;   // the capacities a retry needs
            if (total_out) *total_out = tot;
            return REDIO_ERR_ARG;
        }
    }
    std::vector<long> src, dst; // pending pushes into the device sample_buffer since the last flush
    auto flush = [&]() -> int {
        if (src.empty()) return REDIO_OK;
        if (src.size() > t->seg_cap) {
            hipFree(t->d_src); hipFree(t->d_dst); t->d_src = t->d_dst = nullptr; t->seg_cap = 0;
            IN_TRY(hipMalloc((void **)&t->d_src, src.size() * 2 * sizeof(long)));
            IN_TRY(hipMalloc((void **)&t->d_dst, src.size() * 2 * sizeof(long)));
            t->seg_cap = src.size() * 2;
        }
        IN_TRY(hipMemcpyAsync(t->d_src, src.data(), src.size() * sizeof(long), hipMemcpyHostToDevice, st));
        IN_TRY(hipMemcpyAsync(t->d_dst, dst.data(), dst.size() * sizeof(long), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gather_blocks_kernel, dim3((unsigned)src.size()), dim3(256), 0, st, (const float *)d_blocks, t->d_src, t->d_dst,
                           (int)block, t->d_buf);
        IN_TRY(hipStreamSynchronize(st)); // src/dst are reused
        src.clear(); dst.clear();
        return REDIO_OK;
    };
    for (size_t b = 0; b < nblocks; ++b) {
        t->trigger -= 1;                                              // :46
        if (t->len > 1000 * (size_t)trigger_duration * block_size) {  // :52-54
            rc = flush(); if (rc) return rc;
            IN_TRY(hipMemsetAsync(t->d_buf, 0, sizeof(float), st));
            t->len = 1;
        }
        if (t->threshold == 0.0f) t->threshold = s[b];                // :57-59
        if (t->trigger < 0) {                                         // :62-65
            t->threshold += s[b] / 1000.0f;
            t->threshold -= t->threshold * 0.002f;
        }
        if (s[b] > t->threshold * 4.0f) t->trigger = trigger_duration; // :68-70
        if (t->trigger > 1) {                                          // :73-75
            if (t->len + block > t->cap) { rc = flush(); if (rc) return rc; rc = trig_reserve(t, t->len + block, st); if (rc) return rc; }
            src.push_back((long)b); dst.push_back((long)t->len);
            t->len += block;
        }
        if (t->trigger == 0) {                                         // :78-81
            rc = flush(); if (rc) return rc;
            if (t->len) IN_TRY(hipMemcpyAsync((float *)d_out + total, t->d_buf, t->len * sizeof(float), hipMemcpyDeviceToDevice, st));
            lens[nemit] = t->len; // fits: checked by the dry run above
            total += t->len;
            ++nemit;
            t->len = 0;
        }
    }
    rc = flush();
    if (rc) return rc;
    IN_TRY(hipStreamSynchronize(st));
    if (nemit_out) *nemit_out = nemit;
    if (total_out) *total_out = total;
    return REDIO_OK;
}
