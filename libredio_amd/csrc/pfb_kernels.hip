// pfb_kernels.hip -- 64-channel polyphase channelizer (BASELINE.json configs[3]) on gfx950.
// Lane programs, layouts and the arithmetic contract: pfb_core.h.
//
// Algorithmic bytes: 8 B read + 8 B written per input sample (16 B/sample); 4 P + 30 flop per sample.
// Bound: HBM.  Each wavefront owns a contiguous range of rows (time instants), carries the P-1 rows
// of filter history in registers from tile to tile so every input row is loaded exactly once
// (512 contiguous bytes per wave instruction), and prefetches the next tile's rows while it computes.
#include "fft_wave.h"
#include "pfb_core.h"
#include <type_traits>
#include "redio_internal.h"

namespace redio {

// Cache policy of the two streams (round 6), NT bit 0: non-temporal row loads, bit 1: non-temporal row stores.  Every input row is read
// once and every output row written once, and the plain 1 : 1 copy of this pool runs 6 % faster with non-temporal accesses,
// but what a kernel gains depends on its store pattern and on what else bounds it -- measured per kernel (profiles/r06_channelizer_nt.txt):
//   the one-kernel shapes (pfb_p2_kernel, 16-byte stores of whole rows): both non-temporal, + 2-4 %;
//   pfb64_kernel from cf32: the DEFAULT policy (non-temporal stores of its 32-byte-per-lane rows cost 11 %, loads alone 2 %);
//   pfb64_kernel from u8 bytes: row-major both non-temporal (+ 5 %), the grouped layouts stores only (+ 4 %).
typedef float pfb_v2f __attribute__((ext_vector_type(2)));
typedef float pfb_v4f __attribute__((ext_vector_type(4)));
template <int NT>
__device__ __forceinline__ float2 pfb_ld_row(const float2 *p)
{
    if constexpr (NT & 1) {
        const pfb_v2f v = __builtin_nontemporal_load(reinterpret_cast<const pfb_v2f *>(p));
        return make_float2(v.x, v.y);
    } else return *p;
}
template <int NT>
__device__ __forceinline__ unsigned short pfb_ld_row(const unsigned short *p)
{
    if constexpr (NT & 1) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int NT>
__device__ __forceinline__ void pfb_st(float4 *p, float4 v)
{
    if constexpr (NT & 2) __builtin_nontemporal_store(pfb_v4f{v.x, v.y, v.z, v.w}, reinterpret_cast<pfb_v4f *>(p));
    else *p = v;
}
template <int NT>
__device__ __forceinline__ void pfb_st(float2 *p, float2 v)
{
    if constexpr (NT & 2) __builtin_nontemporal_store(pfb_v2f{v.x, v.y}, reinterpret_cast<pfb_v2f *>(p));
    else *p = v;
}


// ROWMAJOR: the plain [row][64] output (ngroups == 1) with 32-byte stores and no index division.
// IN_U8: `x` is the receiver's interleaved u8 I/Q bytes (rtlsdr::data_to_samples, rtlsdr.rs:159-162); a lane's 8-byte sample load
// becomes a 2-byte load, converted when the sample enters the register window: 2 + 8 bytes per sample through HBM instead of 8 + 8
// (+ 2 + 8 for a conversion kernel in front).
// Prefetch of the next tile's rows: cf32 rows are requested BEFORE the current tile's arithmetic into a second set of registers; u8 rows
// AFTER the branch filters, into the registers those have just emptied (164 instead of 180+ registers: 0.765 -> 0.726 ms from bytes; the
// cf32 kernel measured 0.810 -> 0.825 ms that way and keeps the early request; profiles/r04_channelizer_64_experiments.txt)
// IN_U8 == 2 (round 5; 4-byte aligned streams, the grouped output layouts): a wave instruction loads TWO rows of bytes, one dword (two samples)
// per lane -- lanes 0-31 the even row of the pair, lanes 32-63 the odd one -- through a buffer descriptor over the wave's own byte range
// (requests past the stream's end return zeros: no clamp arithmetic), and lane l takes its sample out of the dword of lane l / 2 (+ 32) with a
// ds_bpermute and a shift when the row enters the window.  A 16-row set is 8 registers instead of 16, so the next tile's rows are requested
// EARLY into a second set at the register count of the late form.  Measured on one box (profiles/r05_channelizer_u8_two_row_loads.txt): the
// grouped x8 layout (two wavefronts per SIMD either way) 0.820 -> 0.757 ms; the row-major layout (three wavefronts per SIMD in the 2-byte form)
// 0.675 -> 0.703 ms, also with the stores through a range-checked descriptor and the loop's waits counted past them -- it keeps IN_U8 == 1.
#ifndef REDIO_PFB_LATE_PREFETCH
#define REDIO_PFB_LATE_PREFETCH (IN_U8 == 1)
#endif
template <int P, bool FUSED, bool ROWMAJOR, int IN_U8 = 0, int NT = 0>
__global__ __launch_bounds__(256) void pfb64_kernel(const float2 *__restrict__ x, const float *__restrict__ h,
                                                    const float2 *__restrict__ tw, float2 *__restrict__ out, long rows,
                                                    long rows_per_wave, int ngroups)
{
    static_assert(PFB_TILE % P == 0, "the register window rotates in place only if the tile is a multiple of P");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    float2 *lds = reinterpret_cast<float2 *>(smem) + wave * PFB_LDS;
    const long t0 = ((long)blockIdx.x * 4 + wave) * rows_per_wave;
    if (t0 >= rows) return; // wave-uniform; no workgroup barrier in this kernel
    const long t1 = (t0 + rows_per_wave < rows) ? t0 + rows_per_wave : rows;
    const long last_in_row = rows + P - 2; // T - 1

    float g[P];
#pragma unroll
    for (int p = 0; p < P; ++p) g[p] = h[PFB_M * p + lane];
    // row r of the input = (x + 64 r)[lane]: wave-uniform row pointer + one 32-bit lane offset per load; rows past the
    // end of the stream (only the last tiles of the last wave ever ask for them) are clamped on a separate path
    constexpr bool PAIRS = IN_U8 == 2;
    using raw_t = typename std::conditional<IN_U8 != 0, unsigned short, float2>::type; // one sample as it lies in memory
    using set_t = typename std::conditional<PAIRS, unsigned, raw_t>::type;              // one register of a row set
    constexpr int NSET = PAIRS ? PFB_TILE / 2 : PFB_TILE;
    const raw_t *xraw = reinterpret_cast<const raw_t *>(x);
    auto sample = [](raw_t w) -> float2 {
        if constexpr (IN_U8 != 0) return make_float2(i2f(w & 255u), i2f((unsigned)w >> 8));
        else return w;
    };
    // PAIRS: the wave's byte range [row t0, the last row any of its tiles can ask for) and the lane constants of the extraction
    const char *wave_base = nullptr;
    long wave_bytes = 0;
    unsigned bp_addr = 0, bp_shift = 0;
    if constexpr (PAIRS) {
        long nrows_here = rows + P - 1 - t0, cap = rows_per_wave + 2 * PFB_TILE + P;
        nrows_here = nrows_here < cap ? nrows_here : cap;
        nrows_here = nrows_here < (1l << 23) ? nrows_here : (1l << 23);
        wave_base = reinterpret_cast<const char *>(xraw + PFB_M * t0);
        wave_bytes = nrows_here * (2 * PFB_M);
        bp_addr = 4u * ((unsigned)lane >> 1);
        bp_shift = 16u * ((unsigned)lane & 1u);
    }
    // row `ti` of a set, as this lane's sample
    auto row_of = [&](const set_t(&set)[NSET], int ti) -> float2 {
        if constexpr (PAIRS) {
            const unsigned d = (unsigned)__builtin_amdgcn_ds_bpermute((int)(bp_addr + ((ti & 1) ? 128u : 0u)), (int)set[ti >> 1]);
            const unsigned w = d >> bp_shift;
            return make_float2(i2f(w & 255u), i2f((w >> 8) & 255u));
        } else return sample(set[ti]);
    };
    // ONE straight-line path: a row past the end of the stream (only the last tiles of the last wave ever ask for one) is clamped with
    // scalar arithmetic, and a tile's request is never skipped (the last tile of a wave requests rows it will not use).  With a fast and a
    // clamped path, or a request under a condition, the compiler cannot count the loads in flight at a use and waits for ALL of them
    // (vmcnt(0)) in the middle of the branch filters -- i.e. for the request it issued a few instructions earlier.
    auto load_rows = [&](set_t(&dst)[NSET], long first) {
        if constexpr (PAIRS) {
            // A descriptor per request, built with scalar arithmetic: its base is the request's first row, its num_records what is left of
            // the wave's byte range from there, so every load is `4 * lane` in the vector offset + an immediate -- the two operands the
            // range check (what returns zeros for a row pair past the end of the stream) is documented to cover.  Round 5 passed the row
            // offset as the SCALAR offset, which LLVM documents as excluded from the check (the probe in tests/test_gpu_channelizer.py
            // measures that gfx950 does check it, so nothing was read past the buffer -- but the documented rule is the one to build on;
            // advisor, round 5); a running vector offset instead costs the grouped layout 1.7 % (profiles/r06_c4gen_ab.txt).
            const long boff = (first - t0) * (2 * PFB_M);
            const long left = wave_bytes - boff;
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wave_base + boff), 0, (int)(left > 0 ? left : 0), 0x00020000);
#pragma unroll
            for (int k = 0; k < NSET; ++k) dst[k] = __builtin_amdgcn_raw_buffer_load_b32(rq, 4u * (unsigned)lane + 4u * PFB_M * k, 0, (NT & 1) ? 2 : 0);
        } else {
#pragma unroll
            for (int ti = 0; ti < PFB_TILE; ++ti) {
                long r = first + ti;
                r = r < last_in_row ? r : last_in_row;
                dst[ti] = pfb_ld_row<NT>(xraw + PFB_M * r + (unsigned)lane);
            }
        }
    };
    // the transform's twiddles, once per wavefront (pfb_core.h: why not inside the tile loop): 20 scalar and 24 vector registers
    float2 twone = tw[0], twa1[4], twa2[4], twa3[4], twc1[4], twc2[4], twc3[4];
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) {
        twa1[k2] = tw[4 * k2]; twa2[k2] = tw[8 * k2]; twa3[k2] = tw[12 * k2];
        const int k = k2 + 4 * (lane & 3);
        twc1[k2] = tw[k]; twc2[k2] = tw[2 * k]; twc3[k2] = tw[3 * k];
    }
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) { // kept where they are: no reload, no rematerialisation inside the loop
        asm volatile("" : "+v"(twc1[k2].x), "+v"(twc1[k2].y), "+v"(twc2[k2].x), "+v"(twc2[k2].y), "+v"(twc3[k2].x), "+v"(twc3[k2].y));
        asm volatile("" : "+s"(twa1[k2].x), "+s"(twa1[k2].y), "+s"(twa2[k2].x), "+s"(twa2[k2].y), "+s"(twa3[k2].x), "+s"(twa3[k2].y));
    }
    asm volatile("" : "+s"(twone.x), "+s"(twone.y));
    float2 win[P];
#pragma unroll
    for (int p = 0; p < P - 1; ++p) win[p] = sample(pfb_ld_row<NT>(xraw + PFB_M * (t0 + p) + (unsigned)lane));
    // One tile: rows tb .. tb + 15 from `cur`, the next tile's rows requested into `nx`.  The tile loop below is unrolled by two with the
    // roles of the two register sets swapped instead of copying nx -> cur: the copies of a loop-carried array land on the loop's back
    // edge, BEHIND the tile's eight stores, where their wait (vmcnt(0): stores count too on gfx950) made every tile pay the write
    // acknowledgement of the tile before it (round 5; and the transform's twiddles come from registers: pfb_core.h).
    auto tile = [&](long tb, set_t(&cur)[NSET], set_t(&nx)[NSET]) {
        if (!REDIO_PFB_LATE_PREFETCH) load_rows(nx, tb + PFB_TILE + P - 1);
        // branch FIRs: lane = branch, strict fold over p (dsputils.rs:31)
#pragma unroll
        for (int ti = 0; ti < PFB_TILE; ++ti) {
            win[(ti + P - 1) % P] = row_of(cur, ti);
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int p = 0; p < P; ++p) acc = mac<FUSED>(win[(ti + p) % P], g[p], acc);
            lds[pfb_x1_store(ti, lane)] = acc;
        }
        if (REDIO_PFB_LATE_PREFETCH) load_rows(cur, tb + PFB_TILE + P - 1); // into the registers the branch filters have just emptied
        wave_lds_fence();
        float2 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = lds[pfb_x1_load(lane, e)];
        wave_lds_fence();
        pfb_fft64_passAB_pre<false>(v, twone, twa1, twa2, twa3);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2)
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) lds[pfb_x2_store(lane, k1, k2)] = v[k1 + 4 * k2];
        wave_lds_fence();
#pragma unroll
        for (int f = 0; f < 16; ++f) v[f] = lds[pfb_x2_load(lane, f)];
        wave_lds_fence();
        pfb_fft64_passC_pre<false>(v, twc1, twc2, twc3);
        const long row = tb + (lane >> 2);
        if (row < t1) {
            if (ROWMAJOR) { // channels 16 k0 + 4 (lane & 3) + k2 of row tb + (lane >> 2): 32 bytes per lane, 128 per row and k0
                float4 *orow = reinterpret_cast<float4 *>(out + PFB_M * tb) + (unsigned)(32 * (lane >> 2) + 2 * (lane & 3));
#pragma unroll
                for (int k0 = 0; k0 < 4; ++k0) {
                    pfb_st<NT>(orow + 8 * k0, make_float4(v[k0].x, v[k0].y, v[k0 + 4].x, v[k0 + 4].y));
                    pfb_st<NT>(orow + 8 * k0 + 1, make_float4(v[k0 + 8].x, v[k0 + 8].y, v[k0 + 12].x, v[k0 + 12].y));
                }
            } else {
#pragma unroll
                for (int k0 = 0; k0 < 4; ++k0) {
                    if (ngroups <= 16) { // the four k2 of one k0 are consecutive channels of one group: a 32-byte run
                        float2 *dst = out + pfb_out_index(row, pfb_out_channel(lane, k0, 0), rows, ngroups);
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) pfb_st<NT>(dst + k2, v[k0 + 4 * k2]);
                    } else {
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) pfb_st<NT>(out + pfb_out_index(row, pfb_out_channel(lane, k0, k2), rows, ngroups), v[k0 + 4 * k2]);
                    }
                }
            }
        }
    };
    set_t ra[NSET], rb[NSET];
    load_rows(ra, t0 + P - 1);
    for (long tb = t0; tb < t1; tb += 2 * PFB_TILE) {
        if constexpr (REDIO_PFB_LATE_PREFETCH) { // one register set: the request goes into the registers the branch filters have emptied
            tile(tb, ra, ra);
            if (tb + PFB_TILE < t1) tile(tb + PFB_TILE, ra, ra);
        } else {
            tile(tb, ra, rb);
            if (tb + PFB_TILE < t1) tile(tb + PFB_TILE, rb, ra);
        }
    }
}

bool pfb_supported(int nchan, int taps_per_branch)
{
    return nchan == PFB_M && (taps_per_branch == 4 || taps_per_branch == 8 || taps_per_branch == 16);
}

template <int P, int IN_U8 = 0>
static hipError_t launch_pfb_t(const float2 *x, const float *h, const float2 *tw, float2 *out, long rows, int ngroups,
                               bool fused, hipStream_t s)
{
    // contiguous row ranges per wave, a multiple of the 16-row tile; aim for >= 8 waves per CU
    long waves = 8L * 256; // 1024 ... 65536 wavefronts measured within 2 % of each other from 2048 up (profiles/r04_channelizer_64_experiments.txt)
    if (const char *e = measure_env("REDIO_PFB_WAVES")) { const long v = atol(e); if (v >= 1) waves = v; } // measurement only
    long rpw = (rows + waves - 1) / waves;
    rpw = ((rpw + PFB_TILE - 1) / PFB_TILE) * PFB_TILE;
    if (rpw < 4 * PFB_TILE) rpw = 4 * PFB_TILE; // amortise the P-1 row prologue
    const long nwaves = (rows + rpw - 1) / rpw;
    const unsigned grid = (unsigned)((nwaves + 3) / 4);
    const size_t lds = 4 * PFB_LDS * sizeof(float2);
    // cache policy per variant (the table at the top of this file); measurement builds: REDIO_PFB_NT = 0 .. 3 overrides it
    constexpr int NT_NAT = IN_U8 == 0 ? 0 : 3, NT_GRP = IN_U8 == 0 ? 0 : 2;
    auto go = [&](auto nt_nat, auto nt_grp) {
        constexpr int A = decltype(nt_nat)::value, B = decltype(nt_grp)::value;
        if (ngroups == 1) {
            if (fused) hipLaunchKernelGGL((pfb64_kernel<P, true, true, IN_U8, A>), dim3(grid), dim3(256), lds, s, x, h, tw, out, rows, rpw, ngroups);
            else hipLaunchKernelGGL((pfb64_kernel<P, false, true, IN_U8, A>), dim3(grid), dim3(256), lds, s, x, h, tw, out, rows, rpw, ngroups);
        } else {
            if (fused) hipLaunchKernelGGL((pfb64_kernel<P, true, false, IN_U8, B>), dim3(grid), dim3(256), lds, s, x, h, tw, out, rows, rpw, ngroups);
            else hipLaunchKernelGGL((pfb64_kernel<P, false, false, IN_U8, B>), dim3(grid), dim3(256), lds, s, x, h, tw, out, rows, rpw, ngroups);
        }
    };
#ifdef REDIO_MEASURE
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    if (const char *e = measure_env("REDIO_PFB_NT")) {
        if (P == 16) { // the BASELINE shape only: every policy of every variant is one more instantiation of a large kernel
            switch (atoi(e)) {
            case 0: go(I0{}, I0{}); return hipGetLastError();
            case 1: go(I1{}, I1{}); return hipGetLastError();
            case 2: go(I2{}, I2{}); return hipGetLastError();
            case 3: go(I3{}, I3{}); return hipGetLastError();
            default: break;
            }
        }
    }
#endif
    go(std::integral_constant<int, NT_NAT>{}, std::integral_constant<int, NT_GRP>{});
    return hipGetLastError();
}

hipError_t launch_pfb(const float2 *x, const float *h, const float2 *tw64, float2 *out, long rows, int taps_per_branch,
                      int ngroups, bool fused, hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    switch (taps_per_branch) {
    case 16: return launch_pfb_t<16>(x, h, tw64, out, rows, ngroups, fused, s);
    case 8: return launch_pfb_t<8>(x, h, tw64, out, rows, ngroups, fused, s);
    case 4: return launch_pfb_t<4>(x, h, tw64, out, rows, ngroups, fused, s);
    default: return hipErrorNotSupported;
    }
}

// the same kernel reading u8 I/Q bytes (2-byte aligned; 16 taps per branch, the BASELINE shape, only)
hipError_t launch_pfb_u8(const void *bytes, const float *h, const float2 *tw64, float2 *out, long rows, int taps_per_branch, int ngroups, bool fused,
                         hipStream_t s)
{
    if (rows <= 0) return hipSuccess;
    if (taps_per_branch != 16 || (reinterpret_cast<uintptr_t>(bytes) & 1)) return hipErrorNotSupported;
    // round 5: two rows per wave instruction (a dword per lane) for the grouped layouts when the bytes are 4-byte aligned
    if (ngroups != 1 && !(reinterpret_cast<uintptr_t>(bytes) & 3) && !measure_env("REDIO_PFB_U8_SHORTS"))
        return launch_pfb_t<16, 2>((const float2 *)bytes, h, tw64, out, rows, ngroups, fused, s);
    return launch_pfb_t<16, 1>((const float2 *)bytes, h, tw64, out, rows, ngroups, fused, s);
}

} // namespace redio
