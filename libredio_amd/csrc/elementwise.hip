// elementwise.hip -- the vector maps of the kpn plumbing on device:
//   kpn::mul_vecs  src/kpn/src/kpn.rs:198-203   out[i] = x[i] * c[i]   (zip: the shorter length wins)
//   kpn::sum_vecs  src/kpn/src/kpn.rs:227-231   out[i] = x[i] + c[i]
// for f32 and for Complex<f32> (num 0.1.22: (ar*br - ai*bi, ar*bi + ai*br), every operation rounded on
// its own -- Rust never contracts).  HBM-bound: 3 words moved per word produced.
#include "../../include/redio.h"
#include "redio_internal.h"

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
