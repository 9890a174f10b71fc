// runs.hip -- the run-length / bit-field stage of the shipped graph (src/ratpak.rs:77-119) on device --
// SURVEY.md 8f rank 4.  Integer work, bit-exact by construction, scan-shaped and HBM-bound.
//   kpn::rle      src/kpn/src/kpn.rs:17-29   value stream -> (value, run length); a run is emitted when the
//                                            value CHANGES, so the last run is never flushed (state carries it)
//   kpn::dle      :32-38   run length -> seconds: ct as f32 / s_rate as f32
//   kpn::rld      :50-56   (value, count) -> repeated values
//   kpn::dld      :41-47   (value, seconds) -> repeated values, n = (dur * s_rate) as usize
//   kpn::binconv  :295-299 = eat(:116-124) of b2d(:111-113) per message: MSB-first bit fields -> integers
#include "../../include/redio.h"
#include "redio_internal.h"
#include <new>
#include <string.h>

namespace redio {

constexpr int RUN_ROUND = 2048; // elements per workgroup round in the change-flag scan (8 consecutive per thread)
constexpr int RUN_ROUNDS = 8;   // rounds per workgroup: 16384-element tiles, so that the one-workgroup scan of the tile counts stays short
constexpr int RUN_TILE = RUN_ROUND * RUN_ROUNDS;

__device__ __forceinline__ int is_change(const uint8_t *__restrict__ x, long i, int have_prev, uint8_t prev)
{
    if (i == 0) return have_prev ? (x[0] != prev) : 0;
    return x[i] != x[i - 1];
}

// change flags of the 8 consecutive elements i0 .. i0+7 as a bit mask (bit k: element i0+k differs from
// its predecessor).  One 8-byte load when the group is whole and x + i0 is 8-byte aligned.
// before8 (optional): byte k = the value in front of element i0 + k, i.e. the value of the run that a change at i0 + k ends.
__device__ __forceinline__ unsigned change_mask8(const uint8_t *__restrict__ x, long i0, long n, int have_prev, uint8_t prev, bool aligned,
                                                 unsigned long long *before8 = nullptr)
{
    if (before8) *before8 = 0;
    if (i0 >= n) return 0;
    if (aligned && i0 + 8 <= n) {
        const unsigned long long w = *reinterpret_cast<const unsigned long long *>(x + i0);
        const unsigned long long before = i0 == 0 ? (have_prev ? prev : (w & 0xff)) : x[i0 - 1];
        if (before8) *before8 = (w << 8) | before;
        const unsigned long long d = w ^ ((w << 8) | before);
        // byte k of d is non-zero  <=>  bit 7 of byte k of nz is set
        const unsigned long long nz = (((d & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | d) & 0x8080808080808080ull;
        // gather the eight bit-7s into one byte: (nz >> 7) has bit 8k set; multiply packs them into the top byte
        return (unsigned)(((nz >> 7) * 0x0102040810204080ull) >> 56);
    }
    unsigned m = 0;
    for (int k = 0; k < 8; ++k)
        if (i0 + k < n && is_change(x, i0 + k, have_prev, prev)) {
            m |= 1u << k;
            if (before8) *before8 |= (unsigned long long)(i0 + k == 0 ? prev : x[i0 + k - 1]) << (8 * k);
        }
    return m;
}

// pass 1: number of value changes per tile (thread t owns elements 8 t .. 8 t + 7 of the tile, as pass 3 does)
__global__ __launch_bounds__(256) void rle_count_kernel(const uint8_t *__restrict__ x, long n, int have_prev, uint8_t prev,
                                                        unsigned *__restrict__ tile_counts)
{
    const bool aligned = ((uintptr_t)x & 7) == 0;
    int c = 0;
#pragma unroll
    for (int r = 0; r < RUN_ROUNDS; ++r) c += __popc(change_mask8(x, (long)blockIdx.x * RUN_TILE + r * RUN_ROUND + (long)threadIdx.x * 8, n, have_prev, prev, aligned));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    __shared__ int ws[4];
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) tile_counts[blockIdx.x] = (unsigned)(ws[0] + ws[1] + ws[2] + ws[3]);
}

// pass 2: exclusive scan of the tile counts (one workgroup; tiles <= a few hundred thousand).  A thread owns EIGHT consecutive counts per
// round (round 3: with one count per thread the 131072 tiles of a 2^28-byte message took 128 rounds of three barriers each, 184 us --
// as long as the pass over the bytes)
__global__ __launch_bounds__(1024) void scan_tiles_kernel(unsigned *__restrict__ counts, long ntiles, unsigned long long *__restrict__ total)
{
    constexpr int PER = 8;
    __shared__ unsigned long long carry;
    __shared__ unsigned long long ws[16];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long base = 0; base < ntiles; base += 1024 * PER) {
        const long i0 = base + (long)threadIdx.x * PER;
        unsigned v[PER];
        unsigned long long sum = 0;
#pragma unroll
        for (int k = 0; k < PER; ++k) { v[k] = i0 + k < ntiles ? counts[i0 + k] : 0u; sum += v[k]; }
        unsigned long long incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            unsigned long long o = __shfl_up(incl, off);
            if ((int)(threadIdx.x & 63) >= off) incl += o;
        }
        if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long wave_off = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += ws[w];
        unsigned long long excl = carry + wave_off + incl - sum;
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            if (i0 + k < ntiles) counts[i0 + k] = (unsigned)excl; // the emitted-run count of a call is bounded by 2^32-1
            excl += v[k];
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = excl;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// pass 3: position of every change, written at its rank (tile offset + rank inside the tile)
__global__ __launch_bounds__(256) void rle_positions_kernel(const uint8_t *__restrict__ x, long n, int have_prev, uint8_t prev,
                                                            const unsigned *__restrict__ tile_offsets, long *__restrict__ pos, uint8_t *__restrict__ vals)
{
    __shared__ int wsum[4];
    const long base = (long)blockIdx.x * RUN_TILE;
    const bool aligned = ((uintptr_t)x & 7) == 0;
    long tile_rank = (long)tile_offsets[blockIdx.x]; // rank of the first change of the current round
    unsigned masks[RUN_ROUNDS];
    unsigned long long befores[RUN_ROUNDS];
#pragma unroll
    for (int r = 0; r < RUN_ROUNDS; ++r) masks[r] = change_mask8(x, base + r * RUN_ROUND + (long)threadIdx.x * 8, n, have_prev, prev, aligned, &befores[r]); // all loads first
#pragma unroll
    for (int r = 0; r < RUN_ROUNDS; ++r) {
        // each thread owns 8 CONSECUTIVE elements of the round so that ranks follow stream order
        const long i0 = base + r * RUN_ROUND + (long)threadIdx.x * 8;
        const unsigned mask = masks[r];
        const int c = __popc(mask);
        int incl = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int o = __shfl_up(incl, off);
            if ((int)(threadIdx.x & 63) >= off) incl += o;
        }
        if (r) __syncthreads(); // the previous round's wave sums have been read
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) woff += wsum[w];
        long rk = tile_rank + woff + incl - c;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (mask & (1u << k)) { vals[rk] = (uint8_t)(befores[r] >> (8 * k)); pos[rk++] = i0 + k; } // the run's value here: pass 4 gathers nothing
        tile_rank += wsum[0] + wsum[1] + wsum[2] + wsum[3];
    }
}

// pass 4: length of run k = distance of change k to the previous change (its value was written by pass 3); the carried run length of
// earlier calls joins the first run
__global__ __launch_bounds__(256) void rle_emit_kernel(const long *__restrict__ pos, long nruns, unsigned long long carried,
                                                       unsigned long long *__restrict__ counts)
{
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nruns) return;
    const long p = pos[k];
    counts[k] = k == 0 ? (unsigned long long)p + carried : (unsigned long long)(p - pos[k - 1]);
}

__global__ __launch_bounds__(256) void dle_kernel(const unsigned long long *__restrict__ ct, long n, float s_rate, float *__restrict__ out)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)ct[i] / s_rate; // ct as f32 / s_rate as f32 (kpn.rs:35)
}

__global__ __launch_bounds__(256) void dld_counts_kernel(const float *__restrict__ dur, long n, float s_rate, unsigned long long *__restrict__ ct)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = dur[i] * s_rate; // (dur*s_rate) as usize (kpn.rs:44): saturating cast, NaN -> 0
    ct[i] = v > 0.0f ? (v >= 18446744073709551616.0f ? ~0ull : (unsigned long long)v) : 0ull;
}

// run k fills out[start[k] .. start[k]+count[k]) with vals[k]; one workgroup per run, grid-stride
__global__ __launch_bounds__(256) void rld_fill_kernel(const uint8_t *__restrict__ vals, const unsigned long long *__restrict__ counts,
                                                       const unsigned long long *__restrict__ starts, long nruns, uint8_t *__restrict__ out)
{
    for (long k = blockIdx.x; k < nruns; k += gridDim.x) {
        const uint8_t v = vals[k];
        uint8_t *d = out + starts[k];
        for (unsigned long long i = threadIdx.x; i < counts[k]; i += blockDim.x) d[i] = v;
    }
}

// exclusive scan of u64 counts in one workgroup (run lists are short compared with sample streams)
__global__ __launch_bounds__(1024) void scan_u64_kernel(const unsigned long long *__restrict__ in, long n, unsigned long long *__restrict__ out,
                                                        unsigned long long *__restrict__ total)
{
    __shared__ unsigned long long carry;
    __shared__ unsigned long long ws[16];
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (long base = 0; base < n; base += 1024) {
        const long i = base + threadIdx.x;
        unsigned long long v = i < n ? in[i] : 0, incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            unsigned long long o = __shfl_up(incl, off);
            if ((int)(threadIdx.x & 63) >= off) incl += o;
        }
        if ((threadIdx.x & 63) == 63) ws[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long wave_off = 0;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += ws[w];
        const unsigned long long excl = carry + wave_off + incl - v;
        if (i < n) out[i] = excl;
        __syncthreads();
        if (threadIdx.x == 1023) carry = excl + v;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

// binconv: one thread per (message, field); bits are one byte each (0/1), MSB first (kpn.rs:111-124)
__global__ __launch_bounds__(256) void binconv_kernel(const uint8_t *__restrict__ bits, long nmsg, int nbits, const int *__restrict__ starts,
                                                      const int *__restrict__ widths, int nfields, unsigned long long *__restrict__ out)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nmsg * nfields) return;
    const long m = t / nfields;
    const int f = (int)(t - m * nfields);
    const uint8_t *b = bits + m * nbits + starts[f];
    unsigned long long acc = 0;
    const int w = widths[f];
    for (int i = 0; i < w; ++i) acc += ((unsigned long long)1 << (w - i - 1)) * b[i];
    out[t] = acc;
}

} // namespace redio
using namespace redio;

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? REDIO_OK : REDIO_ERR_HIP_BASE - (int)e; }
#define RN_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return hip_rc(_e); } while (0)

struct redio_rle {
    int device;
    int have_prev;        // false until the first sample ever arrives (kpn.rs:18)
    uint8_t prev;         // x
    unsigned long long i; // current run length (kpn.rs:19)
    unsigned *d_tiles; size_t tiles_cap;
    long *d_pos; size_t pos_cap;
    unsigned long long *d_total;
};

extern "C" int redio_rle_create(redio_rle **h)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return REDIO_ERR_NO_DEVICE;
    redio_rle *r = new (std::nothrow) redio_rle();
    if (!r) return REDIO_ERR_NOMEM;
    memset(r, 0, sizeof(*r));
    r->device = dev;
    hipError_t e = hipMalloc((void **)&r->d_total, sizeof(unsigned long long));
    if (e != hipSuccess) { delete r; return hip_rc(e); }
    *h = r;
    return REDIO_OK;
}
extern "C" int redio_rle_destroy(redio_rle *r)
{
    if (!r) return REDIO_OK;
    hipFree(r->d_tiles); hipFree(r->d_pos); hipFree(r->d_total);
    delete r;
    return REDIO_OK;
}

// Feeds n one-byte values (device).  Emits up to cap completed runs into d_vals / d_counts (device);
// *nruns = runs completed by this call.  Synchronous (the run count sizes the outputs).
extern "C" int redio_rle_feed(redio_rle *r, const void *d_in, size_t n, void *d_vals, void *d_counts, size_t cap, size_t *nruns,
                              void *stream)
{
    if (nruns) *nruns = 0;
    if (!r) return REDIO_ERR_ARG;
    if (n == 0) return REDIO_OK;
    if (!d_in) return REDIO_ERR_ARG;
    RN_TRY(hipSetDevice(r->device));
    hipStream_t st = (hipStream_t)stream;
    const uint8_t *x = (const uint8_t *)d_in;
    const long ntiles = (long)((n + RUN_TILE - 1) / RUN_TILE);
    if ((size_t)ntiles > r->tiles_cap) {
        hipFree(r->d_tiles); r->d_tiles = nullptr; r->tiles_cap = 0;
        RN_TRY(hipMalloc((void **)&r->d_tiles, (size_t)ntiles * sizeof(unsigned)));
        r->tiles_cap = (size_t)ntiles;
    }
    hipLaunchKernelGGL(rle_count_kernel, dim3((unsigned)ntiles), dim3(256), 0, st, x, (long)n, r->have_prev, r->prev, r->d_tiles);
    hipLaunchKernelGGL(scan_tiles_kernel, dim3(1), dim3(1024), 0, st, r->d_tiles, ntiles, r->d_total);
    unsigned long long total = 0;
    RN_TRY(hipMemcpyAsync(&total, r->d_total, sizeof(total), hipMemcpyDeviceToHost, st));
    uint8_t last = 0;
    RN_TRY(hipMemcpyAsync(&last, x + n - 1, 1, hipMemcpyDeviceToHost, st));
    RN_TRY(hipStreamSynchronize(st));
    long last_change = -1; // position of the last change in this call, for the carried run length
    if (total > 0) {
        if (total > cap || !d_vals || !d_counts) return REDIO_ERR_ARG;
        if (total > r->pos_cap) {
            hipFree(r->d_pos); r->d_pos = nullptr; r->pos_cap = 0;
            RN_TRY(hipMalloc((void **)&r->d_pos, (size_t)total * sizeof(long)));
            r->pos_cap = (size_t)total;
        }
        hipLaunchKernelGGL(rle_positions_kernel, dim3((unsigned)ntiles), dim3(256), 0, st, x, (long)n, r->have_prev, r->prev, r->d_tiles, r->d_pos, (uint8_t *)d_vals);
        hipLaunchKernelGGL(rle_emit_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, r->d_pos, (long)total, r->i,
                           (unsigned long long *)d_counts);
        RN_TRY(hipMemcpyAsync(&last_change, r->d_pos + (total - 1), sizeof(long), hipMemcpyDeviceToHost, st));
        RN_TRY(hipStreamSynchronize(st));
    }
    // state for the next call: the open run
    if (!r->have_prev) { // the very first sample only seeds x and i = 1 (kpn.rs:18-19)
        r->have_prev = 1;
        r->i = (total > 0) ? (unsigned long long)(n - (size_t)last_change) : (unsigned long long)n;
    } else {
        r->i = (total > 0) ? (unsigned long long)(n - (size_t)last_change) : r->i + (unsigned long long)n;
    }
    r->prev = last;
    if (nruns) *nruns = (size_t)total;
    return hip_rc(hipGetLastError());
}

extern "C" int redio_dle(const void *d_counts, size_t n, size_t s_rate, void *d_seconds, void *stream)
{
    if (n == 0) return REDIO_OK;
    if (!d_counts || !d_seconds) return REDIO_ERR_ARG;
    hipLaunchKernelGGL(dle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned long long *)d_counts,
                       (long)n, (float)s_rate, (float *)d_seconds);
    return hip_rc(hipGetLastError());
}

// rld: expands nruns (value, count) pairs; d_scratch holds nruns+1 u64.  *nout = total length (must be <= cap).
extern "C" int redio_rld(const void *d_vals, const void *d_counts, size_t nruns, void *d_out, size_t cap, void *d_scratch, size_t *nout,
                         void *stream)
{
    if (nout) *nout = 0;
    if (nruns == 0) return REDIO_OK;
    if (!d_vals || !d_counts || !d_scratch) return REDIO_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    unsigned long long *starts = (unsigned long long *)d_scratch;
    hipLaunchKernelGGL(scan_u64_kernel, dim3(1), dim3(1024), 0, st, (const unsigned long long *)d_counts, (long)nruns, starts, starts + nruns);
    unsigned long long total = 0;
    RN_TRY(hipMemcpyAsync(&total, starts + nruns, sizeof(total), hipMemcpyDeviceToHost, st));
    RN_TRY(hipStreamSynchronize(st));
    if (nout) *nout = (size_t)total;
    if (total == 0) return REDIO_OK;
    if (total > cap || !d_out) return REDIO_ERR_ARG;
    const unsigned grid = (unsigned)(nruns < 16384 ? nruns : 16384);
    hipLaunchKernelGGL(rld_fill_kernel, dim3(grid), dim3(256), 0, st, (const uint8_t *)d_vals, (const unsigned long long *)d_counts, starts,
                       (long)nruns, (uint8_t *)d_out);
    return hip_rc(hipGetLastError());
}

// dld: durations -> counts ((dur*s_rate) as usize), then rld; d_scratch holds 2*nruns+1 u64
extern "C" int redio_dld(const void *d_vals, const void *d_seconds, size_t nruns, float s_rate, void *d_out, size_t cap, void *d_scratch,
                         size_t *nout, void *stream)
{
    if (nout) *nout = 0;
    if (nruns == 0) return REDIO_OK;
    if (!d_vals || !d_seconds || !d_scratch) return REDIO_ERR_ARG;
    unsigned long long *ct = (unsigned long long *)d_scratch + nruns + 1;
    hipLaunchKernelGGL(dld_counts_kernel, dim3((unsigned)((nruns + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float *)d_seconds,
                       (long)nruns, s_rate, ct);
    return redio_rld(d_vals, ct, nruns, d_out, cap, d_scratch, nout, stream);
}

// binconv: nmsg messages of nbits one-byte binary digits -> nfields integers each (widths on the host);
// a width list that overruns a message is the reference's slice panic -> REDIO_ERR_ASSERT
extern "C" int redio_binconv(const void *d_bits, size_t nmsg, size_t nbits, const size_t *widths, size_t nfields, void *d_out, void *stream)
{
    if (nmsg == 0 || nfields == 0) return REDIO_OK;
    if (!d_bits || !widths || !d_out) return REDIO_ERR_ARG;
    if (nfields > 64) return REDIO_ERR_UNSUPPORTED;
    int hs[64], hw[64];
    size_t off = 0;
    for (size_t f = 0; f < nfields; ++f) {
        if (widths[f] > 64) return REDIO_ERR_UNSUPPORTED;
        hs[f] = (int)off; hw[f] = (int)widths[f];
        off += widths[f];
        if (off > nbits) return REDIO_ERR_ASSERT;
    }
    int *d_tab = nullptr;
    RN_TRY(hipMalloc((void **)&d_tab, 128 * sizeof(int)));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemcpyAsync(d_tab, hs, nfields * sizeof(int), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(d_tab + 64, hw, nfields * sizeof(int), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
        const long total = (long)(nmsg * nfields);
        hipLaunchKernelGGL(binconv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const uint8_t *)d_bits, (long)nmsg, (int)nbits,
                           d_tab, d_tab + 64, (int)nfields, (unsigned long long *)d_out);
        e = hipStreamSynchronize(st); // the field table is freed below
    }
    hipFree(d_tab);
    return hip_rc(e);
}
