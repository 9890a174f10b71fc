// elementwise.hip -- the vector maps of the kpn plumbing on device:
//   kpn::mul_vecs  src/kpn/src/kpn.rs:198-203   out[i] = x[i] * c[i]   (zip: the shorter length wins)
//   kpn::sum_vecs  src/kpn/src/kpn.rs:227-231   out[i] = x[i] + c[i]
// for f32 and for Complex<f32> (num 0.1.22: (ar*br - ai*bi, ar*bi + ai*br), every operation rounded on
// its own -- Rust never contracts).  HBM-bound: 3 words moved per word produced.
#include "../../include/redio.h"
#include "redio_internal.h"

namespace redio {

template <int OP>
__global__ __launch_bounds__(256) void zip_f32_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ o,
                                                      long n4, const float *__restrict__ ta, const float *__restrict__ tb,
                                                      float *__restrict__ to, long tail)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = a[i], y = b[i];
        float4 r;
        if (OP == 0) { r.x = mul_rn(x.x, y.x); r.y = mul_rn(x.y, y.y); r.z = mul_rn(x.z, y.z); r.w = mul_rn(x.w, y.w); }
        else { r.x = add_rn(x.x, y.x); r.y = add_rn(x.y, y.y); r.z = add_rn(x.z, y.z); r.w = add_rn(x.w, y.w); }
        o[i] = r;
    }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < tail; i += stride) { // what the 16-byte groups do not cover
        const float x = ta[i], y = tb[i];
        to[i] = OP == 0 ? mul_rn(x, y) : add_rn(x, y);
    }
}

template <int OP>
__global__ __launch_bounds__(256) void zip_c32_kernel(const float4 *__restrict__ a, const float4 *__restrict__ b, float4 *__restrict__ o,
                                                      long n2, const float2 *__restrict__ ta, const float2 *__restrict__ tb,
                                                      float2 *__restrict__ to, long tail)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) {
        const float4 x = a[i], y = b[i];
        float2 p, q;
        if (OP == 0) { p = cmul_rn(make_float2(x.x, x.y), make_float2(y.x, y.y)); q = cmul_rn(make_float2(x.z, x.w), make_float2(y.z, y.w)); }
        else { p = cadd_rn(make_float2(x.x, x.y), make_float2(y.x, y.y)); q = cadd_rn(make_float2(x.z, x.w), make_float2(y.z, y.w)); }
        o[i] = make_float4(p.x, p.y, q.x, q.y);
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

#define ZIP_ENTRY(name, T, kernel, OP)                                                                                         \
    extern "C" int name(const void *d_a, const void *d_b, void *d_out, size_t n, void *stream)                                 \
    {                                                                                                                          \
        hipStream_t st = (hipStream_t)stream;                                                                                  \
        return zip_common(d_a, d_b, d_out, n, sizeof(T), [&](long nvec, size_t tail_at, long tail) {                            \
            long work = nvec > tail ? nvec : tail;                                                                             \
            unsigned grid = (unsigned)((work + 255) / 256);                                                                    \
            if (grid < 1) grid = 1;                                                                                            \
            if (grid > 256u * 32u) grid = 256u * 32u;                                                                          \
            hipLaunchKernelGGL((kernel<OP>), dim3(grid), dim3(256), 0, st, (const float4 *)d_a, (const float4 *)d_b,           \
                               (float4 *)d_out, nvec, (const T *)d_a + tail_at, (const T *)d_b + tail_at, (T *)d_out + tail_at, \
                               tail);                                                                                          \
        });                                                                                                                    \
    }

ZIP_ENTRY(redio_mul_f32, float, zip_f32_kernel, 0)
ZIP_ENTRY(redio_add_f32, float, zip_f32_kernel, 1)
ZIP_ENTRY(redio_mul_c32, float2, zip_c32_kernel, 0)
ZIP_ENTRY(redio_add_c32, float2, zip_c32_kernel, 1)
