// redio_device.h -- shared helpers for the gfx950 kernels.
//
// Every arithmetic helper exists in two flavours so that the lane/LDS index logic of the kernels
// can be exercised on the CPU (tests/emu) with the very same source: under hipcc the helpers map to
// the non-contracting intrinsics (__fmul_rn/__fadd_rn) or to fmaf; under g++ (built with
// -ffp-contract=off) they are the plain operators.
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RD_HD __host__ __device__ __forceinline__
#define RD_D __device__ __forceinline__
#else
#include <math.h>
#define RD_HD inline
#define RD_D inline
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 r = {x, y}; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = {x, y, z, w}; return r; }
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define RD_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// pins a float2 as ONE 64-bit register pair (two scalar operands would break v_pk_* formation)
#define RD_PIN_F2(a)                                        \
    do {                                                    \
        redio_v2f _t = {(a).x, (a).y};                      \
        asm volatile("" : "+v"(_t) : : "memory");           \
        (a).x = _t.x;                                       \
        (a).y = _t.y;                                       \
    } while (0)
typedef float redio_v2f __attribute__((ext_vector_type(2)));
// names a wave-uniform value and a vector value as inputs of an ordered, memory-clobbering statement
#define RD_PIN_SV(s, v) asm volatile("" : : "s"(s), "v"(v) : "memory")
#else
#define RD_SCHED_BARRIER() ((void)0)
#define RD_PIN_F2(a) ((void)0)
#define RD_PIN_SV(s, v) ((void)0)
#endif

namespace redio {

// one rounding per operation, never contracted (reference semantics: Rust / plain-cc kissfft)
RD_HD float mul_rn(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fmul_rn(a, b);
#else
    return a * b; // host builds use -ffp-contract=off
#endif
}
RD_HD float add_rn(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fadd_rn(a, b);
#else
    return a + b;
#endif
}
RD_HD float sub_rn(float a, float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fsub_rn(a, b);
#else
    return a - b;
#endif
}
RD_HD float fma_rn(float a, float b, float c)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __fmaf_rn(a, b, c);
#else
    return fmaf(a, b, c);
#endif
}

// acc <- acc + x*h with the rounding behaviour selected at compile time
template <bool FUSED>
RD_HD float mac(float x, float h, float acc)
{
    if (FUSED) return fma_rn(x, h, acc);
    return add_rn(acc, mul_rn(x, h));
}
template <bool FUSED>
RD_HD float2 mac(float2 x, float h, float2 acc)
{
    return make_float2(mac<FUSED>(x.x, h, acc.x), mac<FUSED>(x.y, h, acc.y));
}

// complex helpers in the published kissfft macro order: C_MUL = (ar*br - ai*bi, ar*bi + ai*br)
RD_HD float2 cmul_rn(float2 a, float2 b)
{
    return make_float2(sub_rn(mul_rn(a.x, b.x), mul_rn(a.y, b.y)), add_rn(mul_rn(a.x, b.y), mul_rn(a.y, b.x)));
}
RD_HD float2 cadd_rn(float2 a, float2 b) { return make_float2(add_rn(a.x, b.x), add_rn(a.y, b.y)); }
RD_HD float2 csub_rn(float2 a, float2 b) { return make_float2(sub_rn(a.x, b.x), sub_rn(a.y, b.y)); }

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// rtlsdr.rs:159: i as f32 / 127.0 - 1.0.  The IEEE quotient without the division sequence: 1/127 as a two-float constant r_hi + r_lo
// (r_hi = fl(1/127), r_lo = fl(1/127 - r_hi)), q = fma(b, r_hi, fl(b * r_lo)): b * r_lo is far below half an ulp of the result, and the sum is
// rounded once -- the correctly rounded b/127 for every byte b (the domain has 256 points: tests/test_gpu_ingest.py checks all of them, and
// all 65536 byte pairs, against the oracle's plain division).  One multiply + one FMA per component; round 5's form (b * r, the exact
// residual b - 127 q0 in an FMA, q0 + e * r) took three.
__device__ __forceinline__ float i2f(unsigned b)
{
    const float fb = (float)b;
    constexpr float r_hi = 0x1.020408p-7f, r_lo = 0x1.020408p-35f;
    return sub_rn(fma_rn(fb, r_hi, mul_rn(fb, r_lo)), 1.0f);
}
#endif

// murmur3 fmix32-based synthetic input; identical on host and device by construction
RD_HD uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
RD_HD uint32_t hash32(uint32_t seed, uint64_t index)
{
    uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
    uint32_t h = fmix32(lo ^ seed);
    return fmix32(h ^ (hi * 0x9E3779B9u + 0x7F4A7C15u));
}
RD_HD float unit_from_hash(uint32_t h) { return (float)(h >> 8) * 1.1920928955078125e-07f - 1.0f; }

} // namespace redio
