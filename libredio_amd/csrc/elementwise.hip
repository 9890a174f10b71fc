// elementwise.hip -- the vector maps of the kpn plumbing on device:
//   kpn::mul_vecs  src/kpn/src/kpn.rs:198-203   out[i] = x[i] * c[i]   (zip: the shorter length wins)
//   kpn::sum_vecs  src/kpn/src/kpn.rs:227-231   out[i] = x[i] + c[i]
// for f32 and for Complex<f32> (num 0.1.22: (ar*br - ai*bi, ar*bi + ai*br), every operation rounded on
// its own -- Rust never contracts).  HBM-bound: 3 words moved per word produced.
#include "../../include/redio.h"
#include "redio_internal.h"
#include <algorithm>

namespace redio {

// NT: non-temporal loads and stores -- for operands well beyond the caches (zip_common), where they stream 7 % faster (round 3: 0.502 against
// 0.541 ms per 2^28 floats); a message that fits the L2 keeps the default policy, its consumer usually runs next
typedef float zip_v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 zip_ld(const float4 *p)
{
    if constexpr (NT) {
        const zip_v4f v = __builtin_nontemporal_load(reinterpret_cast<const zip_v4f *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else return *p;
}
template <bool NT>
__device__ __forceinline__ void zip_st(float4 *p, float4 r)
{
    if constexpr (NT) __builtin_nontemporal_store(zip_v4f{r.x, r.y, r.z, r.w}, reinterpret_cast<zip_v4f *>(p));
    else *p = r;
}

template <int OP, bool NT>
__global__ __launch_bounds__(256) void zip_f32_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ o,
                                                      long n4, const float *__restrict__ ta, const float *__restrict__ tb,
                                                      float *__restrict__ to, long tail)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = zip_ld<NT>(a + i), y = zip_ld<NT>(b + i);
        float4 r;
        if (OP == 0) { r.x = mul_rn(x.x, y.x); r.y = mul_rn(x.y, y.y); r.z = mul_rn(x.z, y.z); r.w = mul_rn(x.w, y.w); }
        else { r.x = add_rn(x.x, y.x); r.y = add_rn(x.y, y.y); r.z = add_rn(x.z, y.z); r.w = add_rn(x.w, y.w); }
        zip_st<NT>(o + i, r);
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tail; i += stride) { // what the 16-byte groups do not cover
        const float x = ta[i], y = tb[i];
        to[i] = OP == 0 ? mul_rn(x, y) : add_rn(x, y);
    }
}

template <int OP, bool NT>
__global__ __launch_bounds__(256) void zip_c32_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ o,
                                                      long n2, const float2 *__restrict__ ta, const float2 *__restrict__ tb,
                                                      float2 *__restrict__ to, long tail)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const float4 x = zip_ld<NT>(a + i), y = zip_ld<NT>(b + i);
        float2 p, q;
        if (OP == 0) { p = cmul_rn(make_float2(x.x, x.y), make_float2(y.x, y.y)); q = cmul_rn(make_float2(x.z, x.w), make_float2(y.z, y.w)); }
        else { p = cadd_rn(make_float2(x.x, x.y), make_float2(y.x, y.y)); q = cadd_rn(make_float2(x.z, x.w), make_float2(y.z, y.w)); }
        zip_st<NT>(o + i, make_float4(p.x, p.y, q.x, q.y));
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tail; i += stride) {
        const float2 x = ta[i], y = tb[i];
        to[i] = OP == 0 ? cmul_rn(x, y) : cadd_rn(x, y);
    }
}

} // namespace redio
using namespace redio;

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? REDIO_OK : REDIO_ERR_HIP_BASE - (int)e; }

// 16-byte groups go through the vector loop when all three pointers are 16-byte aligned; the rest
// (a short tail, or everything when a pointer is not aligned) through the scalar loop of the same launch
template <typename Launch>
static int zip_common(const void *a, const void *b, void *o, size_t n, size_t elem_bytes, Launch launch)
{
    if (n == 0) return REDIO_OK;
    if (!a || !b || !o) return REDIO_ERR_ARG;
    const size_t per16 = 16 / elem_bytes;
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)o) & 15) == 0;
    const size_t nvec = aligned ? n / per16 : 0;
    launch((long)nvec, nvec * per16, (long)(n - nvec * per16));
    return hip_rc(hipGetLastError());
}

// One 16-byte group per thread, workgroups in dispatch order: 0.534 ms per 2^28 floats against 0.679 with the grid capped at 8192
// workgroups and a grid-stride loop (round 3).  HIP rejects a launch of 2^32 threads or more, so the grid stops at 2^24 - 1
// workgroups of 256 and the kernels' grid-stride loop covers what is beyond (operands above 64 GiB).
constexpr size_t ZIP_NT_BYTES = 64u << 20; // per operand: twice the eight L2s
#define ZIP_ENTRY(name, T, kernel, OP)                                                                                         \
    extern "C" int name(const void *d_a, const void *d_b, void *d_out, size_t n, void *stream)                                 \
    {                                                                                                                          \
        hipStream_t st = (hipStream_t)stream;                                                                                  \
        return zip_common(d_a, d_b, d_out, n, sizeof(T), [&](long nvec, size_t tail_at, long tail) {                            \
            long work = nvec > tail ? nvec : tail;                                                                             \
            unsigned grid = (unsigned)((work + 255) / 256 > 0xffffffL ? 0xffffffL : (work + 255) / 256);                       \
            if (grid < 1) grid = 1;                                                                                            \
            auto kern = n * sizeof(T) >= ZIP_NT_BYTES ? kernel<OP, true> : kernel<OP, false>;                                  \
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, (const float4 *)d_a, (const float4 *)d_b,                   \
                               (float4 *)d_out, nvec, (const T *)d_a + tail_at, (const T *)d_b + tail_at, (T *)d_out + tail_at, \
                               tail);                                                                                          \
        });                                                                                                                    \
    }

ZIP_ENTRY(redio_mul_f32, float, zip_f32_kernel, 0)
ZIP_ENTRY(redio_add_f32, float, zip_f32_kernel, 1)
ZIP_ENTRY(redio_mul_c32, float2, zip_c32_kernel, 0)
ZIP_ENTRY(redio_add_c32, float2, zip_c32_kernel, 1)

// ---- the checking sink of a device-resident graph: an order-free 64-bit sum of 32-bit words ----
// Integer addition commutes, so lanes, waves and workgroups may add in any order and the result is exact: 16-byte loads, a per-lane u64
// partial, a wave reduction by __shfl_xor (the order-free case of SURVEY.md 8f rank 1), one atomic per workgroup.  HBM-bound: 4 B per word.
namespace redio {
__global__ __launch_bounds__(256) void checksum_u32_kernel(const uint4 *__restrict__ v, long n4, const uint32_t *__restrict__ t, long tail,
                                                           unsigned long long *__restrict__ sum)
{
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned long long acc = 0;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) { // four independent 16-byte loads in flight per lane
        const uint4 a = v[i], b = v[i + stride], c = v[i + 2 * stride], d = v[i + 3 * stride];
        acc += ((unsigned long long)a.x + a.y + a.z + a.w) + ((unsigned long long)b.x + b.y + b.z + b.w) +
               ((unsigned long long)c.x + c.y + c.z + c.w) + ((unsigned long long)d.x + d.y + d.z + d.w);
    }
    for (; i < n4; i += stride) {
        const uint4 w = v[i];
        acc += (unsigned long long)w.x + w.y + w.z + w.w;
    }
    for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < tail; k += stride) acc += t[k];
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(sum, part[0] + part[1] + part[2] + part[3]);
}
} // namespace redio

extern "C" int redio_checksum_u32(const void *d_words, size_t n, void *d_sum_u64, void *stream)
{
    if (n == 0) return REDIO_OK;
    if (!d_words || !d_sum_u64 || ((uintptr_t)d_words & 3) || ((uintptr_t)d_sum_u64 & 7)) return REDIO_ERR_ARG;
    // words before the first 16-byte boundary and after the last whole group go through the scalar loop of the same launch
    const size_t head = std::min(n, (size_t)((16 - ((uintptr_t)d_words & 15)) & 15) / 4);
    const uint32_t *w = (const uint32_t *)d_words;
    const size_t n4 = (n - head) / 4, rest = n - head - 4 * n4;
    // two workgroups per CU at most: the sums meet in ONE atomic per workgroup, and 4096 of them on one address cost more than the reads
    // of a 27 MB message (round 6: 53 us -> the read time)
    long grid = (long)((n4 + 1023) / 1024);
    grid = grid < 1 ? 1 : (grid > 512 ? 512 : grid);
    if (head) // rare: an unaligned view; its few words take a launch of their own so that the main loop keeps one tail pointer
        hipLaunchKernelGGL(checksum_u32_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const uint4 *)nullptr, 0L, w, (long)head,
                           (unsigned long long *)d_sum_u64);
    hipLaunchKernelGGL(checksum_u32_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, (const uint4 *)(w + head), (long)n4,
                       w + head + 4 * n4, (long)rest, (unsigned long long *)d_sum_u64);
    return hip_rc(hipGetLastError());
}
