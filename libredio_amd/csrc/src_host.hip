// src_host.hip -- the converter state machine of samplerate::resample's native side
// (src/samplerate/src/samplerate.rs:61,76: src_new / src_process) for nchan independent mono streams
// that share ratio and block lengths.  It follows the published libsamplerate 0.1.8 control flow
// (src_process checks, sinc_mono_vari_process, prepare_data) with the ring buffer mirrored in device
// memory; every output sample is computed by src_sinc_exact_kernel.  Between two buffer refills the
// outputs depend on one buffer image, so each refill epoch costs one compute launch.
#include "../../include/redio.h"
#include "redio_internal.h"
#include "src_position.h"
#include "src_internal.h"
#include <math.h>
#include <new>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace redio {

// ---- coefficient tables: same sizes / increments as the library's three sinc converters; the
// contents are a stated design (Kaiser-windowed sinc) because the library's tables are not available
// (DESIGN.md section 2).  Evaluated in double, rounded once to float.
struct SrcTableSpec { int n; int increment; double fc; double beta; };
static const SrcTableSpec kSpecs[3] = {
    {340239, 2381, 0.9650, 16.0}, // SRC_SINC_BEST_QUALITY
    {22438, 491, 0.9425, 12.4},   // SRC_SINC_MEDIUM_QUALITY
    {2464, 128, 0.80, 9.0},       // SRC_SINC_FASTEST
};

static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0;
    const double q = x * x / 4.0;
    for (int k = 1; k < 500; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

bool src_make_table(int converter, std::vector<float> &c, int &half_len, int &increment)
{
    if (converter < 0 || converter > 2) return false;
    const SrcTableSpec &t = kSpecs[converter];
    const double pi = 3.14159265358979323846;
    c.assign((size_t)t.n, 0.0f);
    const int half = t.n - 2;
    const double i0b = bessel_i0(t.beta);
    for (int i = 0; i <= half; ++i) {
        const double x = (double)i / (double)t.increment;
        const double a = pi * t.fc * x;
        const double s = (i == 0) ? 1.0 : sin(a) / a;
        const double r = (double)i / (double)half;
        const double w = bessel_i0(t.beta * sqrt(1.0 - r * r)) / i0b;
        c[(size_t)i] = (float)(t.fc * s * w);
    }
    half_len = half;
    increment = t.increment;
    return true;
}

} // namespace redio

using namespace redio;

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? REDIO_OK : REDIO_ERR_HIP_BASE - (int)e; }
#define SRC_TRY(expr)                               \
    do {                                            \
        hipError_t _e = (expr);                     \
        if (_e != hipSuccess) return hip_rc(_e);    \
    } while (0)

enum { SRC_MAX_RATIO = 256, SRC_SHIFT = 12 };

struct redio_src {
    int device, converter, nchan;
    int coeff_half_len, index_inc;
    float *d_coeffs;
    // converter state (shared by every channel)
    double last_ratio, last_position;
    int b_current, b_end, b_real_end, b_len;
    // device mirror of the library's buffer: two images of [nchan][buf_stride], buf_stride = front + b_len + 1: `front` zero floats that
    // nothing ever writes sit in front of index 0 of every row (d_buf points at index 0 of row 0; d_buf_base is the allocation).
    // DEFINED where the published code is not: when the ratio DECREASES between two calls the filter widens while b_current still
    // sits where the narrower filter left it, and the left wing's data index b_current - coeff_count (and prepare_data's move source
    // b_current - half) can be negative -- libsamplerate 0.1.8 reads the words in front of its buffer there.  Oracle and device read
    // +0.0f (silence before the stream); such a call (front_short) runs the per-lane kernel only, whose reads are not clamped.
    float *d_buf[2], *d_buf_base[2];
    int front, front_short;
    int cur; // which image is live
    long buf_stride;
    // per-call scratch
    // per-output values of the host recurrence, in PINNED host memory so that their uploads are true asynchronous DMAs
    // (a pageable hipMemcpyAsync is staged synchronously: tens of microseconds per refill epoch)
    int *h_pos, *h_start, *h_inc;
    double *h_scale;
    int *d_pos, *d_start, *d_inc;
    double *d_scale;
    size_t d_cap;
    float *d_stage_in, *d_stage_out; // only used by the host-buffer entry point
    // uniform-phase fast path: interpolated coefficients of the current increment, far end first
    std::vector<float> h_coeffs;
    int fast_inc;
    double *d_cl, *d_cr, *d_tabs; // tables inside ONE guarded allocation (d_tabs)
    int ncl, ncr;
    float2 *d_T2; int nm; double fast_scale; // packed f32 tap pairs of the polyphase path
    float *d_Hp; int fastp_nc;               // the same taps per phase, [S][src_fastp_row(fastp_nc)]; fastp_nc = tap pairs per phase of the kernel instantiation
    // periodic-phase path (rational ratios): per-epoch tables [tap][phase] on the device, host copies kept until the call ends
    double *d_pL, *d_pR; size_t pL_cap, pR_cap;
    int *d_pint; // dpos | skipL | skipR, 256 each
    int period_hint;
    char *h_arena; size_t arena_cap, arena_used; // pinned: the periodic tables of one call, carved out epoch by epoch
    long periodic_launches, general_launches, tile_launches; // diagnostics (redio_src_path_counts); tile: general epochs that took the LDS-tile kernel
    // converters 3 / 4 (zero-order hold, linear): the value carried from the previous call, per channel, and the reset flag
    float *d_last; int zl_reset;
    float *d_rows_in, *d_rows_out; size_t rows_in_cap, rows_out_cap; // interleaved host form, nchan > 1: de-interleaved rows
    int mode;                                // REDIO_SRC_EXACT / REDIO_SRC_FAST
    int window_ok;                           // single-launch path enabled (off: one launch per buffer refill)
    int zl_channels;                         // converters 3 / 4: the channel count the library's end-of-input comparison multiplies by --
                                             // the state's channels for an interleaved message, 1 for the batched rows (independent mono streams)
    size_t stage_in_cap, stage_out_cap;
    hipStream_t host_stream; // the host-buffer entry point's own stream: states on different threads do not serialise
};

static inline double fmod_one(double x) { return src_fmod_one(x); }
static bool is_bad_src_ratio(double r) { return r < (1.0 / SRC_MAX_RATIO) || r > (1.0 * SRC_MAX_RATIO); }

extern "C" int redio_src_reset(redio_src *s)
{
    if (!s) return REDIO_SRC_ERR_BAD_STATE;
    SRC_TRY(hipSetDevice(s->device));
    s->last_ratio = 0.0;
    s->last_position = 0.0;
    if (s->converter >= 3) { // zoh_reset / linear_reset
        s->zl_reset = 1;
        SRC_TRY(hipMemset(s->d_last, 0, (size_t)s->nchan * sizeof(float)));
        SRC_TRY(hipStreamSynchronize(nullptr));
        return REDIO_OK;
    }
    s->b_current = s->b_end = 0;
    s->b_real_end = -1;
    s->cur = 0;
    SRC_TRY(hipMemset(s->d_buf_base[0], 0, ((size_t)s->front + (size_t)s->nchan * s->buf_stride) * sizeof(float)));
    SRC_TRY(hipMemset(s->d_buf_base[1], 0, ((size_t)s->front + (size_t)s->nchan * s->buf_stride) * sizeof(float)));
    SRC_TRY(hipStreamSynchronize(nullptr)); // later work runs on non-blocking streams, which do not wait for the default stream
    return REDIO_OK;
}

extern "C" int redio_src_create(redio_src **h, int converter, int nchan)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (nchan < 1) return REDIO_SRC_ERR_BAD_CHANNEL_COUNT;
    std::vector<float> coeffs;
    int half = 0, inc = 0;
    const bool zl = converter == 3 || converter == 4; // SRC_ZERO_ORDER_HOLD / SRC_LINEAR (samplerate.rs:29-30): no table
    if (!zl && !src_make_table(converter, coeffs, half, inc)) return REDIO_SRC_ERR_BAD_CONVERTER;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return REDIO_ERR_NO_DEVICE;
    redio_src *s = new (std::nothrow) redio_src();
    if (!s) return REDIO_ERR_NOMEM;
    s->device = dev; s->converter = converter; s->nchan = nchan;
    s->d_last = nullptr; s->zl_reset = 1; s->d_rows_in = s->d_rows_out = nullptr; s->rows_in_cap = s->rows_out_cap = 0;
    s->coeff_half_len = half; s->index_inc = inc;
    s->d_coeffs = nullptr; s->d_buf[0] = s->d_buf[1] = s->d_buf_base[0] = s->d_buf_base[1] = nullptr; s->front = 0; s->front_short = 0;
    s->d_pos = s->d_start = s->d_inc = nullptr; s->d_scale = nullptr; s->d_cap = 0;
    s->d_stage_in = s->d_stage_out = nullptr; s->stage_in_cap = s->stage_out_cap = 0; s->host_stream = nullptr;
    s->fast_inc = 0; s->d_cl = s->d_cr = s->d_tabs = nullptr; s->ncl = s->ncr = 0;
    s->d_pL = s->d_pR = nullptr; s->pL_cap = s->pR_cap = 0; s->d_pint = nullptr; s->period_hint = 0;
    s->h_pos = s->h_start = s->h_inc = nullptr; s->h_scale = nullptr; s->h_arena = nullptr; s->arena_cap = s->arena_used = 0;
    s->periodic_launches = s->general_launches = s->tile_launches = 0;
    s->d_T2 = nullptr; s->nm = 0; s->fast_scale = 0.0; s->d_Hp = nullptr; s->fastp_nc = 0; s->mode = REDIO_SRC_EXACT; s->window_ok = 1; s->zl_channels = 1;
    s->h_coeffs = coeffs;
    if (zl) {
        s->b_len = 0; s->buf_stride = 0;
        hipError_t ez = hipMalloc((void **)&s->d_last, (size_t)nchan * sizeof(float));
        if (ez != hipSuccess) { redio_src_destroy(s); return hip_rc(ez); }
        const int rcz = redio_src_reset(s);
        if (rcz) { redio_src_destroy(s); return rcz; }
        *h = s;
        return REDIO_OK;
    }
    long bl = lrint(2.5 * half / (inc * 1.0) * SRC_MAX_RATIO);
    if (bl < 4096) bl = 4096;
    s->b_len = (int)bl;
    s->front = (int)lrint((half + 2.0) / inc * SRC_MAX_RATIO) + 64; // the widest filter's reach (oracle/oracle_src.c: the same figure)
    s->front = (s->front + 3) & ~3;                                 // rows keep their 16-byte alignment
    s->buf_stride = (((long)s->b_len + 1 + 3) & ~3l) + s->front;
    hipError_t e = hipMalloc((void **)&s->d_coeffs, coeffs.size() * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(s->d_coeffs, coeffs.data(), coeffs.size() * sizeof(float), hipMemcpyHostToDevice);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipMalloc((void **)&s->d_buf_base[i], ((size_t)s->front + (size_t)nchan * s->buf_stride) * sizeof(float));
        if (e == hipSuccess) s->d_buf[i] = s->d_buf_base[i] + s->front;
    }
    if (e != hipSuccess) { redio_src_destroy(s); return hip_rc(e); }
    int rc = redio_src_reset(s);
    if (rc) { redio_src_destroy(s); return rc; }
    *h = s;
    return REDIO_OK;
}

extern "C" int redio_src_destroy(redio_src *s)
{
    if (!s) return REDIO_OK;
    hipFree(s->d_coeffs); hipFree(s->d_buf_base[0]); hipFree(s->d_buf_base[1]);
    hipFree(s->d_pos); hipFree(s->d_start); hipFree(s->d_inc); hipFree(s->d_scale);
    hipFree(s->d_stage_in); hipFree(s->d_stage_out);
    if (s->host_stream) hipStreamDestroy(s->host_stream);
    hipFree(s->d_tabs); hipFree(s->d_T2); hipFree(s->d_Hp);
    hipFree(s->d_pL); hipFree(s->d_pR); hipFree(s->d_pint);
    hipFree(s->d_last); hipFree(s->d_rows_in); hipFree(s->d_rows_out);
    if (s->h_pos) hipHostFree(s->h_pos);
    if (s->h_start) hipHostFree(s->h_start);
    if (s->h_inc) hipHostFree(s->h_inc);
    if (s->h_scale) hipHostFree(s->h_scale);
    if (s->h_arena) hipHostFree(s->h_arena);
    delete s;
    return REDIO_OK;
}

extern "C" int redio_src_set_mode(redio_src *s, int mode)
{
    if (!s) return REDIO_SRC_ERR_BAD_STATE;
    if (mode != REDIO_SRC_EXACT && mode != REDIO_SRC_FAST && mode != REDIO_SRC_EPOCHS) return REDIO_ERR_ARG;
    if (mode == REDIO_SRC_EPOCHS) { s->window_ok = 0; s->mode = REDIO_SRC_EXACT; }
    else { s->window_ok = 1; s->mode = mode; }
    return REDIO_OK;
}

extern "C" int redio_src_set_ratio(redio_src *s, double ratio)
{
    if (!s) return REDIO_SRC_ERR_BAD_STATE;
    if (is_bad_src_ratio(ratio)) return REDIO_SRC_ERR_BAD_SRC_RATIO;
    s->last_ratio = ratio;
    return REDIO_OK;
}

extern "C" int redio_src_table(int converter, float *coeffs_out, int *half_len, int *increment)
{
    std::vector<float> c;
    int h = 0, inc = 0;
    if (!src_make_table(converter, c, h, inc)) return REDIO_SRC_ERR_BAD_CONVERTER;
    if (half_len) *half_len = h;
    if (increment) *increment = inc;
    if (coeffs_out) memcpy(coeffs_out, c.data(), c.size() * sizeof(float));
    return REDIO_OK;
}

static int ensure_scratch(redio_src *s, size_t nout)
{
    if (nout <= s->d_cap) return REDIO_OK;
    hipFree(s->d_pos); hipFree(s->d_start); hipFree(s->d_inc); hipFree(s->d_scale);
    s->d_pos = s->d_start = s->d_inc = nullptr; s->d_scale = nullptr; s->d_cap = 0;
    size_t cap = nout + nout / 4 + 1024;
    SRC_TRY(hipMalloc((void **)&s->d_pos, cap * sizeof(int)));
    SRC_TRY(hipMalloc((void **)&s->d_start, cap * sizeof(int)));
    SRC_TRY(hipMalloc((void **)&s->d_inc, cap * sizeof(int)));
    SRC_TRY(hipMalloc((void **)&s->d_scale, cap * sizeof(double)));
    s->d_cap = cap;
    if (s->h_pos) hipHostFree(s->h_pos);
    if (s->h_start) hipHostFree(s->h_start);
    if (s->h_inc) hipHostFree(s->h_inc);
    if (s->h_scale) hipHostFree(s->h_scale);
    s->h_pos = s->h_start = s->h_inc = nullptr; s->h_scale = nullptr;
    SRC_TRY(hipHostMalloc((void **)&s->h_pos, cap * sizeof(int), hipHostMallocDefault));
    SRC_TRY(hipHostMalloc((void **)&s->h_start, cap * sizeof(int), hipHostMallocDefault));
    SRC_TRY(hipHostMalloc((void **)&s->h_inc, cap * sizeof(int), hipHostMallocDefault));
    SRC_TRY(hipHostMalloc((void **)&s->h_scale, cap * sizeof(double), hipHostMallocDefault));
    return REDIO_OK;
}

// where the new input of this call lives
struct SrcInput {
    const float *host; // host pointer (mono, nchan == 1) or NULL
    const float *dev;  // device [nchan][in_stride] or NULL
    long in_stride;
};

// prepare_data: refill the (device) buffer image, keeping `half` samples of history before b_current
static int prepare_data(redio_src *f, const SrcInput &in, long in_count, long &in_used, int end_of_input, int half, hipStream_t st)
{
    int len;
    if (f->b_real_end >= 0) return REDIO_OK;
    if (f->b_current == 0) {
        len = f->b_len - 2 * half;
        f->b_current = f->b_end = half;
    } else if (f->b_end + half + 1 < f->b_len) {
        len = f->b_len - f->b_current - half;
        if (len < 0) len = 0;
    } else {
        len = f->b_end - f->b_current;
        // memmove(buffer, buffer + b_current - half, half + len): through the other image
        const int other = f->cur ^ 1;
        SRC_TRY(launch_src_copy_rows(f->d_buf[f->cur], f->buf_stride, f->b_current - half, f->d_buf[other], f->buf_stride, 0,
                                     (long)half + len, f->nchan, st));
        f->cur = other;
        f->b_current = half;
        f->b_end = f->b_current + len;
        len = f->b_len - f->b_current - half;
        if (len < 0) len = 0;
    }
    const long avail = in_count - in_used;
    if (avail < len) len = (int)avail;
    if (len < 0 || f->b_end + len > f->b_len) return REDIO_SRC_ERR_SINC_PREPARE_DATA_BAD_LEN;
    if (len > 0) {
        if (in.host) {
            SRC_TRY(hipMemcpyAsync(f->d_buf[f->cur] + f->b_end, in.host + in_used, (size_t)len * sizeof(float), hipMemcpyHostToDevice, st));
        } else {
            SRC_TRY(launch_src_copy_rows(in.dev, in.in_stride, in_used, f->d_buf[f->cur], f->buf_stride, f->b_end, len, f->nchan, st));
        }
    }
    f->b_end += len;
    in_used += len;
    if (in_used == in_count && f->b_end - f->b_current < 2 * half && end_of_input) {
        if (f->b_len - f->b_end < half + 5) {
            len = f->b_end - f->b_current;
            if (half + len > f->b_len) return REDIO_SRC_ERR_SINC_PREPARE_DATA_BAD_LEN; // include/samplerate.h (iv): the library's move would overrun its buffer
            const int other = f->cur ^ 1;
            SRC_TRY(launch_src_copy_rows(f->d_buf[f->cur], f->buf_stride, f->b_current - half, f->d_buf[other], f->buf_stride, 0,
                                         (long)half + len, f->nchan, st));
            f->cur = other;
            f->b_current = half;
            f->b_end = f->b_current + len;
        }
        f->b_real_end = f->b_end;
        len = half + 5;
        if (len < 0 || f->b_end + len > f->b_len) len = f->b_len - f->b_end;
        SRC_TRY(launch_src_fill_rows(f->d_buf[f->cur], f->buf_stride, f->b_end, len, f->nchan, 0.0f, st));
        f->b_end += len;
    }
    return REDIO_OK;
}

// coefficients of the uniform-phase path for one increment: icoeff exactly as calc_output_single forms it
static int prepare_uniform(redio_src *f, int inc)
{
    if (f->fast_inc == inc && f->d_cl) return REDIO_OK;
    const int maxf = f->coeff_half_len << SRC_SHIFT;
    const int cl = maxf / inc, cr = (maxf - inc) / inc;
    std::vector<double> L((size_t)cl + 1), R((size_t)cr + 1);
    auto icoeff = [&](int filter_index) {
        const double fraction = (double)(filter_index & ((1 << SRC_SHIFT) - 1)) * (1.0 / (double)(1 << SRC_SHIFT));
        const int indx = filter_index >> SRC_SHIFT;
        const float c0 = f->h_coeffs[(size_t)indx];
        const float dc = f->h_coeffs[(size_t)indx + 1] - c0;
        return (double)c0 + fraction * (double)dc;
    };
    for (int t = 0; t <= cl; ++t) L[(size_t)t] = icoeff((cl - t) * inc);          // far end first
    for (int t = 0; t <= cr; ++t) R[(size_t)t] = icoeff((cr - t) * inc + inc);
    // ONE allocation [guard | L | guard | R | guard] of zeros around the tables: the register-blocked kernel's ramp steps
    // load coefficient runs that start up to 3*S + 16 entries before a table and end as far behind it (src_core.h)
    const size_t guard = 4 * 256 + 32; // the uniform-phase paths take S <= 256
    hipFree(f->d_tabs);
    f->d_tabs = nullptr; f->d_cl = f->d_cr = nullptr; f->fast_inc = 0;
    std::vector<double> all(3 * guard + L.size() + R.size(), 0.0);
    std::copy(L.begin(), L.end(), all.begin() + (long)guard);
    std::copy(R.begin(), R.end(), all.begin() + (long)(2 * guard + L.size()));
    SRC_TRY(hipMalloc((void **)&f->d_tabs, all.size() * sizeof(double)));
    SRC_TRY(hipMemcpy(f->d_tabs, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice));
    f->d_cl = f->d_tabs + guard;
    f->d_cr = f->d_tabs + 2 * guard + L.size();
    // at zero phase the right wing starts one increment in and R[t] == L[t] bit for bit over its whole length: both wings
    // then walk ONE table (half the scalar-cache footprint, and the wavefronts of a workgroup share every line they load)
    if (R.size() + 1 == L.size() && memcmp(R.data(), L.data(), R.size() * sizeof(double)) == 0 && !measure_env("REDIO_SRC_TWO_TABLES")) f->d_cr = f->d_cl;
    f->ncl = cl + 1; f->ncr = cr + 1; f->fast_inc = inc;
    hipFree(f->d_T2); f->d_T2 = nullptr; f->nm = 0; // rebuilt on demand for the new increment
    hipFree(f->d_Hp); f->d_Hp = nullptr; f->fastp_nc = 0;
    return REDIO_OK;
}

// packed f32 tap pairs of the polyphase path: H[j] = (float)(scale*icoeff_j) over both wings in data
// order, T2[m] = (H[m], H[m-S])
static int prepare_fast_taps(redio_src *f, int S, double scale)
{
    if (f->d_T2 && f->fast_scale == scale && f->nm >= 0 && f->nm == ((f->ncl + f->ncr + S + 127) & ~127)) return REDIO_OK;
    std::vector<double> L((size_t)f->ncl), R((size_t)f->ncr);
    SRC_TRY(hipMemcpy(L.data(), f->d_cl, L.size() * sizeof(double), hipMemcpyDeviceToHost));
    SRC_TRY(hipMemcpy(R.data(), f->d_cr, R.size() * sizeof(double), hipMemcpyDeviceToHost));
    const int KH = f->ncl + f->ncr;
    std::vector<float> H((size_t)KH);
    for (int j = 0; j < f->ncl; ++j) H[(size_t)j] = (float)(scale * L[(size_t)j]);                       // far end first
    for (int i = 0; i < f->ncr; ++i) H[(size_t)(f->ncl + i)] = (float)(scale * R[(size_t)(f->ncr - 1 - i)]); // near end first
    const int nm = (KH + S + 127) & ~127; // zero filled to a whole number of 16-tap steps for each of 8 waves
    std::vector<float2> T((size_t)nm);
    for (int m = 0; m < nm; ++m)
        T[(size_t)m] = make_float2(m < KH ? H[(size_t)m] : 0.0f, (m - S >= 0 && m - S < KH) ? H[(size_t)(m - S)] : 0.0f);
    hipFree(f->d_T2); f->d_T2 = nullptr;
    SRC_TRY(hipMalloc((void **)&f->d_T2, T.size() * sizeof(float2)));
    SRC_TRY(hipMemcpy(f->d_T2, T.data(), T.size() * sizeof(float2), hipMemcpyHostToDevice));
    f->nm = nm; f->fast_scale = scale;
    // the same taps by phase for the phase-split kernel: Hp[p][j] = H[S*j + p], zero filled to whole chunks of 32 taps
    hipFree(f->d_Hp); f->d_Hp = nullptr;
    f->fastp_nc = src_fastp_pairs(S, KH);
    if (f->fastp_nc > 0) {
        const int ntap = src_fastp_row(f->fastp_nc);
        std::vector<float> P((size_t)S * ntap, 0.0f);
        for (int m = 0; m < KH; ++m) P[(size_t)(m % S) * ntap + m / S] = H[(size_t)m];
        SRC_TRY(hipMalloc((void **)&f->d_Hp, P.size() * sizeof(float)));
        SRC_TRY(hipMemcpy(f->d_Hp, P.data(), P.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return REDIO_OK;
}

// ---- periodic-phase epochs: a constant rational ratio makes (start index, position step) repeat every P outputs ----
// Checks, on the per-output values the library's recurrence just produced, that outputs first .. first+count-1 are
// exactly periodic; builds the P sets of interpolated coefficients with the expression of calc_output_single (in
// double) as tables [tap][phase]; launches src_sinc_periodic_kernel.  Returns 1 when it handled the epoch, 0 when the
// epoch is not eligible (the general kernel then runs), an error code otherwise.
// ONE eligibility rule for the periodic-phase kernel, shared by the epoch path (which applies it to the epoch it is about to run)
// and by the single-launch window path's probe (which applies it to the epoch size it expects): P phases per Q input samples,
// NL + NR taps per phase, an epoch of `count` outputs.  The tables are rebuilt for every epoch (its first output may sit at any
// phase): only worth it while they are small next to the epoch's own work (0.0213 = 213 / 10000 is periodic too, but 213 phases
// x 4300 taps is not a table), and a tile of the kernel must fit the LDS.
static bool periodic_epoch_eligible(int P, int Q, int NL, int NR, int dpos_max, long count)
{
    if (count < 128 || P < 1) return false;
    int NT = 0; size_t lds = 0;
    if (!src_periodic_shape(P, Q, NL, NR, dpos_max, 1, &NT, &lds)) return false;
    return (long)P * (NL + NR) <= 65536 && (long)P * (NL + NR) <= 8 * count;
}

static int try_periodic_epoch(redio_src *f, long first, long count, float *d_out, long out_stride, hipStream_t st)
{
    if (count < 128) return 0;
    const int inc = f->h_inc[(size_t)first];
    const double scale = f->h_scale[(size_t)first];
    const int *pos = f->h_pos + first, *start = f->h_start + first;
    for (long k = 1; k < count; ++k)
        if (f->h_inc[(size_t)(first + k)] != inc || f->h_scale[(size_t)(first + k)] != scale || pos[k] < pos[k - 1]) return 0;
    auto is_period = [&](int P) {
        if (P < 1 || P > 256 || 2L * P > count) return false;
        const int Q = pos[P] - pos[0];
        for (long k = 0; k + P < count; ++k)
            if (start[k + P] != start[k] || pos[k + P] - pos[k] != Q) return false;
        return true;
    };
    int P = 0;
    if (is_period(f->period_hint)) P = f->period_hint;
    for (int c = 1; !P && c <= 256 && 2L * c <= count; ++c)
        if (start[c] == start[0] && is_period(c)) P = c;
    if (!P) return 0;
    f->period_hint = P;
    const int Q = pos[P] - pos[0];
    const int maxf = f->coeff_half_len << SRC_SHIFT;
    auto icoeff = [&](int filter_index) {
        const double fraction = (double)(filter_index & ((1 << SRC_SHIFT) - 1)) * (1.0 / (double)(1 << SRC_SHIFT));
        const int indx = filter_index >> SRC_SHIFT;
        const float c0 = f->h_coeffs[(size_t)indx];
        const float dc = f->h_coeffs[(size_t)indx + 1] - c0;
        return (double)c0 + fraction * (double)dc;
    };
    // the taps each wing loop visits, per phase (far end first), and where its nearest tap lands
    std::vector<std::vector<int>> Lf((size_t)P), Rf((size_t)P);
    int NL = 0, NR = 0;
    for (int p = 0; p < P; ++p) {
        int fi = start[p];
        int cc = (maxf - fi) / inc;
        fi += cc * inc;
        int di = -cc;
        do { Lf[(size_t)p].push_back(fi); fi -= inc; ++di; } while (fi >= 0);
        if (di - 1 != 0) return 0; // the nearest left tap must multiply x[pos] (always, unless start == inc)
        fi = inc - start[p];
        cc = (maxf - fi) / inc;
        fi += cc * inc;
        di = 1 + cc;
        do { Rf[(size_t)p].push_back(fi); fi -= inc; --di; } while (fi > 0);
        if (di + 1 != 1) return 0; // the nearest right tap must multiply x[pos + 1]
        NL = NL > (int)Lf[(size_t)p].size() ? NL : (int)Lf[(size_t)p].size();
        NR = NR > (int)Rf[(size_t)p].size() ? NR : (int)Rf[(size_t)p].size();
    }
    const int dpos_max = pos[P - 1] - pos[0];
    if (!periodic_epoch_eligible(P, Q, NL, NR, dpos_max, count)) return 0;
    // this epoch's tables in the pinned arena (kept until the call's final synchronisation: the uploads are asynchronous)
    const size_t nLd = (size_t)NL * P, nRd = (size_t)NR * P, need = (nLd + nRd) * sizeof(double) + 3 * 256 * sizeof(int);
    if (!f->h_arena) {
        SRC_TRY(hipHostMalloc((void **)&f->h_arena, (size_t)16 << 20, hipHostMallocDefault));
        f->arena_cap = (size_t)16 << 20; f->arena_used = 0;
    }
    if (f->arena_used + need > f->arena_cap) return 0; // a very long call: the remaining epochs take the general kernel
    double *L = reinterpret_cast<double *>(f->h_arena + f->arena_used), *R = L + nLd;
    int *I = reinterpret_cast<int *>(R + nRd);
    f->arena_used += (need + 63) & ~(size_t)63;
    memset(L, 0, (nLd + nRd) * sizeof(double));
    memset(I, 0, 3 * 256 * sizeof(int));
    int maxskipL = 0, maxskipR = 0;
    for (int p = 0; p < P; ++p) {
        const int sl = NL - (int)Lf[(size_t)p].size(), sr = NR - (int)Rf[(size_t)p].size();
        for (size_t t = 0; t < Lf[(size_t)p].size(); ++t) L[(size_t)(sl + (int)t) * P + p] = icoeff(Lf[(size_t)p][t]);
        for (size_t t = 0; t < Rf[(size_t)p].size(); ++t) R[(size_t)(sr + (int)t) * P + p] = icoeff(Rf[(size_t)p][t]);
        I[(size_t)p] = pos[p] - pos[0]; I[256 + (size_t)p] = sl; I[512 + (size_t)p] = sr;
        maxskipL = maxskipL > sl ? maxskipL : sl;
        maxskipR = maxskipR > sr ? maxskipR : sr;
    }
    // device side: one buffer [L | R | ints], same layout as the arena slice, one upload
    if (need > f->pL_cap) {
        hipFree(f->d_pL); f->d_pL = nullptr; f->pL_cap = 0;
        SRC_TRY(hipMalloc((void **)&f->d_pL, 2 * need));
        f->pL_cap = 2 * need;
    }
    SRC_TRY(hipMemcpyAsync(f->d_pL, L, need, hipMemcpyHostToDevice, st));
    const double *dL = f->d_pL, *dR = dL + nLd;
    const int *dI = reinterpret_cast<const int *>(dR + nRd);
    hipError_t e = launch_src_periodic(f->d_buf[f->cur], f->buf_stride, dL, dR, dI, dI + 256, dI + 512, P, Q, NL, NR,
                                       maxskipL, maxskipR, dpos_max, pos[0], scale, d_out + first, out_stride, count, f->nchan, st);
    if (e == hipErrorNotSupported) return 0;
    if (e != hipSuccess) return hip_rc(e);
    ++f->periodic_launches;
    return 1;
}

// how many epochs of this handle ran through the periodic-phase kernel / the general per-tap kernel (tests, tools)
extern "C" int redio_src_path_counts(const redio_src *s, long *periodic, long *general)
{
    if (!s) return REDIO_SRC_ERR_BAD_STATE;
    if (periodic) *periodic = s->periodic_launches;
    if (general) *general = s->general_launches;
    return REDIO_OK;
}

// flush the outputs decided since the last refill: they all read the current buffer image
static int flush_epoch(redio_src *f, long first, long count, float *d_out, long out_stride, hipStream_t st)
{
    if (count <= 0) return REDIO_OK;
    // uniform phase? (every start index 0, one increment, positions in arithmetic progression)
    if (!f->front_short) { // (front_short: a tap can sit in front of the image -- only the per-lane kernel below reads there unclamped)
        const int inc = f->h_inc[(size_t)first];
        const int S = count > 1 ? f->h_pos[(size_t)first + 1] - f->h_pos[(size_t)first] : 1;
        bool uniform = S >= 1 && S <= 256;
        for (long k = 0; uniform && k < count; ++k)
            uniform = f->h_start[(size_t)(first + k)] == 0 && f->h_inc[(size_t)(first + k)] == inc &&
                      f->h_scale[(size_t)(first + k)] == f->h_scale[(size_t)first] &&
                      f->h_pos[(size_t)(first + k)] == f->h_pos[(size_t)first] + (int)k * S;
        if (uniform) {
            const int maxf = f->coeff_half_len << SRC_SHIFT;
            const int cl = maxf / inc, cr = (maxf - inc) / inc;
            if (src_uniform_lds(64, S, cl, cr)) {
                int rc = prepare_uniform(f, inc);
                if (rc) return rc;
                SRC_TRY(launch_src_uniform(f->d_buf[f->cur], f->buf_stride, f->d_cl, f->ncl, f->d_cr, f->ncr, f->h_pos[(size_t)first], S,
                                           f->h_scale[(size_t)first], d_out + first, out_stride, count, f->nchan, st));
                return REDIO_OK;
            }
        }
    }
    if (f->window_ok && !f->front_short) { // rational ratios: P sets of coefficients instead of one interpolation per tap
        const int handled = try_periodic_epoch(f, first, count, d_out, out_stride, st);
        if (handled == 1) return REDIO_OK;
        if (handled != 0) return handled;
    }
    ++f->general_launches;
    SRC_TRY(hipMemcpyAsync(f->d_pos + first, f->h_pos + first, (size_t)count * sizeof(int), hipMemcpyHostToDevice, st));
    SRC_TRY(hipMemcpyAsync(f->d_start + first, f->h_start + first, (size_t)count * sizeof(int), hipMemcpyHostToDevice, st));
    if (f->window_ok && !f->front_short) { // constant increment and scale, positions in order: the LDS-tile form of the general kernel
        const int inc = f->h_inc[(size_t)first];
        const double scale = f->h_scale[(size_t)first];
        bool tile_ok = count >= 64;
        for (long k = 1; tile_ok && k < count; ++k)
            tile_ok = f->h_inc[(size_t)(first + k)] == inc && f->h_scale[(size_t)(first + k)] == scale &&
                      f->h_pos[(size_t)(first + k)] >= f->h_pos[(size_t)(first + k - 1)];
        const size_t lds = tile_ok ? src_tile_lds_bytes(f->h_pos + first, count, 128, f->coeff_half_len, inc) : 0;
        if (lds) {
            hipError_t e = launch_src_tile(f->d_buf[f->cur], f->buf_stride, nullptr, 0, f->buf_stride, f->buf_stride, f->d_coeffs, f->coeff_half_len,
                                           f->d_pos + first, f->d_start + first, inc, scale, d_out + first, out_stride, count, f->nchan, lds, st);
            if (e == hipSuccess) { ++f->tile_launches; return REDIO_OK; }
            if (e != hipErrorNotSupported) return hip_rc(e);
        }
    }
    SRC_TRY(hipMemcpyAsync(f->d_inc + first, f->h_inc + first, (size_t)count * sizeof(int), hipMemcpyHostToDevice, st));
    SRC_TRY(hipMemcpyAsync(f->d_scale + first, f->h_scale + first, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st));
    SRC_TRY(launch_src_exact(f->d_buf[f->cur], f->buf_stride, f->d_coeffs, f->coeff_half_len, f->d_pos + first, f->d_start + first,
                             f->d_inc + first, f->d_scale + first, d_out + first, out_stride, count, f->nchan, st));
    return REDIO_OK;
}

// ---- single-launch path for uniform phase (constant ratio, 1/ratio an integer, zero phase, no flush) ----
// Runs the library's control flow on counters only (same decisions, same in_used / out_gen / final
// buffer state), evaluates every output in one launch over [old image | new input], then rebuilds the
// part of the buffer image that later calls can read.  Returns 1 when it handled the call, 0 when the
// call is not eligible (the epoch path then runs), < 0 / error code on failure.
static int try_uniform_window(redio_src *f, const SrcInput &in, long in_count, float *d_out, long out_stride, long out_count,
                              double src_ratio_arg, int end_of_input, long *in_used_out, long *out_gen_out, hipStream_t st)
{
    if (end_of_input || f->b_real_end >= 0 || !in.dev) return 0;
    if (fabs(f->last_ratio - src_ratio_arg) > 1e-10) return 0;
    const double src_ratio = f->last_ratio;
    const double step = 1.0 / src_ratio;
    const int S = (int)step;
    if ((double)S != step || S < 1 || S > 256) return 0;
    if (fmod_one(f->last_position) != 0.0 || f->last_position != 0.0) return 0;
    double count = (f->coeff_half_len + 2.0) / f->index_inc;
    const double minr = f->last_ratio < src_ratio_arg ? f->last_ratio : src_ratio_arg;
    if (minr < 1.0) count /= minr;
    const int half = (int)lrint(count) + 1;
    const double float_increment = f->index_inc * (src_ratio < 1.0 ? src_ratio : 1.0);
    const int inc = (int)lrint(float_increment * (double)(1 << SRC_SHIFT));
    const double scale = float_increment / f->index_inc;
    const int maxf = f->coeff_half_len << SRC_SHIFT;
    const int cl = maxf / inc, cr = (maxf - inc) / inc;
    if (!src_uniform_lds(64, S, cl, cr)) return 0;

    // dry run of sinc_mono_vari_process / prepare_data
    int b_current = f->b_current, b_end = f->b_end;
    long A0 = 0, a_in0 = -1, in_used = 0, out_gen = 0, a_first = -1;
    while (out_gen < out_count) {
        int samples_in_hand = (b_end - b_current + f->b_len) % f->b_len;
        if (samples_in_hand <= half) {
            int len;
            if (b_current == 0) {
                len = f->b_len - 2 * half;
                b_current = b_end = half;
            } else if (b_end + half + 1 < f->b_len) {
                len = f->b_len - b_current - half;
                if (len < 0) len = 0;
            } else {
                len = b_end - b_current;
                A0 += b_current - half;
                b_current = half;
                b_end = b_current + len;
                len = f->b_len - b_current - half;
                if (len < 0) len = 0;
            }
            const long avail = in_count - in_used;
            if (avail < len) len = (int)avail;
            if (len < 0 || b_end + len > f->b_len) return 0; // let the epoch path report the library's error
            if (len > 0 && a_in0 < 0) a_in0 = A0 + b_end - in_used;
            b_end += len;
            in_used += len;
            samples_in_hand = (b_end - b_current + f->b_len) % f->b_len;
            if (samples_in_hand <= half) break;
        }
        if (a_first < 0) a_first = A0 + b_current;
        else if (A0 + b_current != a_first + (long)S * out_gen) return 0; // cannot happen; be safe
        // the library emits one output and advances S samples per turn of this loop until the samples in hand fall to `half`: that run
        // of turns in one step (samples_in_hand - j*S > half for j < run; no index wraps inside it: b_current + j*S < b_end), so the dry
        // run costs one turn per buffer refill instead of one per output (84 000 turns = 0.25 ms of host time per 2^22-frame call)
        long run = ((long)samples_in_hand - half + S - 1) / S;
        if (run > out_count - out_gen) run = out_count - out_gen;
        if (run < 1 || (long)b_current + (run - 1) * S >= f->b_len) run = 1; // (never: the buffer is linear between two refills)
        out_gen += run;
        b_current = (int)(((long)b_current + run * S) % f->b_len);
    }
    if (a_in0 < 0) a_in0 = A0 + b_end; // no input consumed: every index is served by the old image
    int rc = prepare_uniform(f, inc);
    if (rc) return rc;
    const bool fast = f->mode == REDIO_SRC_FAST;
    if (fast) { rc = prepare_fast_taps(f, S, scale); if (rc) return rc; }
    // the final image.  A call without a move (A0 == 0: the library only appended) appends the new samples to the LIVE image in place, like
    // prepare_data above -- the launch reads the image below a_in0 only and the append writes from a_in0 on, so the image keeps everything
    // since its last move and a small message costs its own samples, not the widest filter's reach (round 5; advisor, round 4).
    // A call with a move rebuilds the other buffer: [b_current - reach, b_end) with `reach` the WIDEST filter's (ratio 1 / 256), not this
    // call's -- a later call with a lower ratio reads that far back, and the library's buffer holds the true history there
    // (round 4: rebuilding only this call's half left stale cells a falling ratio then read; tests/fuzz_parity.py seed 60407)
    const bool in_place = A0 == 0;
    const int other = in_place ? f->cur : f->cur ^ 1;
    long j0 = in_place ? a_in0 : (long)b_current - f->front, j1 = b_end;
    if (j0 < 0) j0 = 0;
    hipError_t e = launch_src_window(f->d_buf[f->cur], f->buf_stride, in.dev, in.in_stride, a_in0, f->d_cl, f->ncl, f->d_cr, f->ncr,
                                     f->d_T2, f->nm, f->d_Hp, f->fastp_nc, fast, a_first < 0 ? 0 : a_first, S, scale, d_out, out_stride, out_gen, f->nchan,
                                     A0, j0, j1, f->d_buf[other], st);
    if (e == hipErrorNotSupported) return 0;
    if (e != hipSuccess) return hip_rc(e);
    f->cur = other;
    f->b_current = b_current;
    f->b_end = b_end;
    f->last_position = 0.0;
    f->last_ratio = src_ratio;
    if (in_used_out) *in_used_out = in_used;
    if (out_gen_out) *out_gen_out = out_gen;
    return 1;
}

// zoh_vari_process / linear_vari_process (src_zoh.c, src_linear.c) on device rows: the recurrence on the host in double,
// the samples on the device.  All channels share the per-output (index, fraction); frame units throughout.
static int zoh_linear_impl(redio_src *f, const float *d_in, long in_stride, long input_frames, float *d_out, long out_stride, long output_frames,
                           double src_ratio_arg, long *in_used_out, long *out_gen_out, hipStream_t st)
{
    const bool lin = f->converter == 4;
    if (input_frames <= 0) return REDIO_OK;
    if (f->zl_reset) { // just reset: the value "before" the stream is its first frame
        SRC_TRY(launch_src_copy_rows(d_in, in_stride, 0, f->d_last, 1, 0, 1, f->nchan, st));
        f->zl_reset = 0;
    }
    int rc = ensure_scratch(f, (size_t)output_frames);
    if (rc) return rc;
    const long in_count = input_frames, out_count = output_frames;
    long in_used = 0, out_gen = 0;
    double src_ratio = f->last_ratio, input_index = f->last_position, rem;
    // The library counts interleaved SAMPLES (in_used, in_count multiples of the channel count) and compares
    // in_used + channels * input_index with in_count in double; this loop counts frames, so the comparisons are written out in the
    // library's units -- for a channel count that is not a power of two the product rounds, and the frame form of the same
    // inequality can decide differently when the position sits on the boundary (found by the randomised run: 3 channels, 48000 / 44100).
    const int units = f->zl_channels > 0 ? f->zl_channels : 1;
    const double chd = (double)units;
    auto samples = [&](long frames) { return (double)(frames * (long)units); };
    while (input_index < 1.0 && out_gen < out_count) {
        if (lin ? (samples(in_used) + chd * (1.0 + input_index) >= samples(in_count)) : (samples(in_used) + chd * input_index >= samples(in_count))) break;
        if (out_count > 0 && fabs(f->last_ratio - src_ratio_arg) > 1e-20)
            src_ratio = f->last_ratio + samples(out_gen) * (src_ratio_arg - f->last_ratio) / samples(out_count); // the library's out_gen / out_count count samples
        f->h_pos[(size_t)out_gen] = -1;
        f->h_scale[(size_t)out_gen] = input_index;
        ++out_gen;
        input_index += 1.0 / src_ratio;
    }
    rem = fmod_one(input_index);
    in_used += lrint(input_index - rem);
    input_index = rem;
    while (out_gen < out_count && (lin ? (samples(in_used) + chd * input_index < samples(in_count)) : (samples(in_used) + chd * input_index <= samples(in_count)))) {
        if (out_count > 0 && fabs(f->last_ratio - src_ratio_arg) > 1e-20)
            src_ratio = f->last_ratio + samples(out_gen) * (src_ratio_arg - f->last_ratio) / samples(out_count); // the library's out_gen / out_count count samples
        f->h_pos[(size_t)out_gen] = (int)(in_used - 1);
        f->h_scale[(size_t)out_gen] = input_index;
        ++out_gen;
        input_index += 1.0 / src_ratio;
        rem = fmod_one(input_index);
        in_used += lrint(input_index - rem);
        input_index = rem;
    }
    if (in_used > in_count) {
        input_index += (double)(in_used - in_count);
        in_used = in_count;
    }
    if (out_gen > 0) {
        SRC_TRY(hipMemcpyAsync(f->d_pos, f->h_pos, (size_t)out_gen * sizeof(int), hipMemcpyHostToDevice, st));
        SRC_TRY(hipMemcpyAsync(f->d_scale, f->h_scale, (size_t)out_gen * sizeof(double), hipMemcpyHostToDevice, st));
        SRC_TRY(launch_src_zoh_linear(d_in, in_stride, f->d_last, f->d_pos, f->d_scale, d_out, out_stride, out_gen, f->nchan, lin, st));
    }
    f->last_position = input_index;
    if (in_used > 0) SRC_TRY(launch_src_copy_rows(d_in, in_stride, in_used - 1, f->d_last, 1, 0, 1, f->nchan, st));
    f->last_ratio = src_ratio;
    if (in_used_out) *in_used_out = in_used;
    if (out_gen_out) *out_gen_out = out_gen;
    return REDIO_OK;
}

// Would the refill epochs of this call run the periodic-phase kernel (small per-phase tables, no per-tap interpolation: the faster
// form where it applies)?  The period test of try_periodic_epoch on the first n outputs of the dry run, then the SAME eligibility
// rule (periodic_epoch_eligible) on the epoch the library's refill logic would produce: about `epoch_outputs` outputs per refill.
// (Round 2 tested the table size only: a ratio whose epochs the periodic rule then declined lost the single-launch window and ran
// one small launch per refill.)
static bool window_prefers_periodic(const redio_src *f, long n, int inc, long epoch_outputs)
{
    const int *pos = f->h_pos, *start = f->h_start;
    for (int P = 1; P <= 256 && 2L * P <= n; ++P) {
        if (start[P] != start[0]) continue;
        const int Q = pos[P] - pos[0];
        bool ok = true;
        for (long k = 0; ok && k + P < n; ++k) ok = start[k + P] == start[k] && pos[k + P] - pos[k] == Q;
        if (!ok) continue;
        // try_periodic_epoch applies the rule to the epoch's real wing lengths (at most wing + 1 taps) and its real output count
        // (epoch_outputs is an estimate): answer "periodic" only with a margin on both, so that a ratio near the table-size or
        // 8 x count boundary keeps the single-launch window instead of having every epoch declined afterwards
        const int wing = (int)((long)(f->coeff_half_len << SRC_SHIFT) / inc) + 3;
        return periodic_epoch_eligible(P, Q, wing, wing, pos[P - 1] - pos[0], epoch_outputs - epoch_outputs / 8);
    }
    return false;
}

// ---- single-launch path for ANY constant ratio ---------------------------------------------------------------------------
// The library's control flow (refills, positions, phases) is run on counters with its own double recurrence, exactly as the
// epoch loop of src_process_impl runs it, but nothing is launched per refill: every output's absolute position in the window
// [old buffer image | new input] and its start index are recorded and the LDS-tile kernel evaluates them 16384 at a time (a call
// at ratio 0.0213 has 46 refill epochs of 120 outputs each: far too little work per launch; a long call's next chunk is being
// decided by the host while the previous one runs), then the part of the buffer image later calls can read is rebuilt.  Ratios whose
// phases repeat with small tables are left to the epoch path's periodic kernel (window_prefers_periodic).  Same in_used / out_gen / final state as the epoch path (tests compare the two).  Returns 1 when it
// handled the call, 0 when not eligible, an error code otherwise.
static int try_general_window(redio_src *f, const SrcInput &in, long in_count, float *d_out, long out_stride, long out_count,
                              double src_ratio_arg, int end_of_input, long *in_used_out, long *out_gen_out, hipStream_t st)
{
    if (end_of_input || f->b_real_end >= 0 || !in.dev || out_count < 256) return 0;
    if (fabs(f->last_ratio - src_ratio_arg) > 1e-10) return 0; // the ratio varies inside the call: per-output increments
    const double src_ratio = f->last_ratio;
    double count = (f->coeff_half_len + 2.0) / f->index_inc;
    const double minr = f->last_ratio < src_ratio_arg ? f->last_ratio : src_ratio_arg;
    if (minr < 1.0) count /= minr;
    const int half = (int)lrint(count) + 1;
    const double fp_one = (double)(1 << SRC_SHIFT);
    const double float_increment = f->index_inc * (src_ratio < 1.0 ? src_ratio : 1.0);
    const int inc = (int)lrint(float_increment * fp_one);
    const double scale = float_increment / f->index_inc;
    const double step = 1.0 / src_ratio;
    constexpr int NT = 128;          // outputs per workgroup of the tile kernel
    constexpr long CHUNK = 16384;    // outputs per launch: the host's recurrence for the next chunk runs beside the kernel of this one
    constexpr long PROBE = 1024;     // outputs looked at before the first launch to choose between this path and periodic epochs
    // outputs between two refills of the library's buffer: a refill brings in at most b_len - 2*half frames, ratio outputs per frame
    const long epoch_outputs = (long)((double)(f->b_len - 2 * half) * src_ratio);
    if ((double)NT * step + 2.0 * (double)((long)(f->coeff_half_len << SRC_SHIFT) / inc) + 16.0 > 15000.0) return 0; // no LDS tile holds it
    int rc = ensure_scratch(f, (size_t)out_count);
    if (rc) return rc;
    auto wrap = [&](int x) { while (x >= f->b_len) x -= f->b_len; return x; };
    const long a_limit_max = 0x7fffffffl;
    // dry run of sinc_mono_vari_process / prepare_data on counters
    int b_current = f->b_current, b_end = f->b_end;
    long A0 = 0, in_used = 0, out_gen = 0, launched = 0;
    // absolute index of input[0]: the window is [old image | input], and the image's b_end is where the next new sample goes --
    // through every refill (a move shifts A0 and b_end by opposite amounts, an append moves b_end and in_used together)
    const long a_in0 = (b_current == 0 && f->b_end == 0) ? half : (long)b_end;
    const long a_limit = a_in0 + in_count > 0 ? a_in0 + in_count : 1; // absolute indices below this exist in the window
    bool probed = false;
    auto launch_chunk = [&](long first, long n) -> int {
        if (n <= 0) return 1;
        const size_t lds = src_tile_lds_bytes(f->h_pos + first, n, NT, f->coeff_half_len, inc);
        if (!lds) return 0;
        if (hipMemcpyAsync(f->d_pos + first, f->h_pos + first, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(f->d_start + first, f->h_start + first, (size_t)n * sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess)
            return hip_rc(hipGetLastError());
        hipError_t e = launch_src_tile(f->d_buf[f->cur], f->buf_stride, in.dev, in.in_stride, a_in0, a_limit, f->d_coeffs, f->coeff_half_len,
                                       f->d_pos + first, f->d_start + first, inc, scale, d_out + first, out_stride, n, f->nchan, lds, st);
        if (e == hipErrorNotSupported) return 0;
        if (e != hipSuccess) return hip_rc(e);
        ++f->general_launches; ++f->tile_launches;
        return 1;
    };
    // giving the call back to the epoch path after launches were issued: they read h_pos / h_start asynchronously, and that path
    // refills the same arrays -- wait for them first (the outputs they wrote are simply written again)
    auto give_back = [&](int code) { if (launched > 0) hipStreamSynchronize(st); return code; };
    double input_index = f->last_position;
    double rem = fmod_one(input_index);
    b_current = wrap(b_current + (int)lrint(input_index - rem));
    input_index = rem;
    while (out_gen < out_count) {
        int samples_in_hand = wrap(b_end - b_current + f->b_len);
        if (samples_in_hand <= half) {
            int len;
            if (b_current == 0) {
                len = f->b_len - 2 * half;
                b_current = b_end = half;
            } else if (b_end + half + 1 < f->b_len) {
                len = f->b_len - b_current - half;
                if (len < 0) len = 0;
            } else {
                len = b_end - b_current;
                A0 += b_current - half;
                b_current = half;
                b_end = b_current + len;
                len = f->b_len - b_current - half;
                if (len < 0) len = 0;
            }
            const long avail = in_count - in_used;
            if (avail < len) len = (int)avail;
            if (len < 0 || b_end + len > f->b_len) return give_back(0); // let the epoch path report the library's error
            if (len > 0 && A0 + b_end - in_used != a_in0) return give_back(0); // cannot happen (see a_in0)
            b_end += len;
            in_used += len;
            samples_in_hand = wrap(b_end - b_current + f->b_len);
            if (samples_in_hand <= half) break;
        }
        int start_fp = (int)lrint(input_index * float_increment * fp_one);
        long at = A0 + b_current;
        if (start_fp == inc) { start_fp = 0; at += 1; } // the next sample at phase zero: same taps, same samples, same order
        if (at >= a_limit_max) return give_back(0);
        f->h_start[(size_t)out_gen] = start_fp;
        f->h_pos[(size_t)out_gen] = (int)at;
        ++out_gen;
        if (!probed && out_gen == PROBE) {
            probed = true;
            if (window_prefers_periodic(f, out_gen, inc, epoch_outputs)) return 0;
        }
        if (out_gen - launched >= CHUNK) {
            const int r = launch_chunk(launched, out_gen - launched);
            if (r != 1) return give_back(r);
            launched = out_gen;
        }
        b_current = wrap(b_current + src_advance(input_index, step));
    }
    if (!probed && window_prefers_periodic(f, out_gen, inc, epoch_outputs)) return 0;
    {
        const int r = launch_chunk(launched, out_gen - launched);
        if (r != 1) return give_back(r);
        launched = out_gen;
    }
    // the final image: appended in place when the call moved nothing, else [b_current - widest filter's reach, b_end) rebuilt into the other
    // buffer (see try_uniform_window)
    const bool in_place = A0 == 0;
    const int other = in_place ? f->cur : f->cur ^ 1;
    long j0 = in_place ? a_in0 : (long)b_current - f->front, j1 = b_end;
    if (j0 < 0) j0 = 0;
    SRC_TRY(launch_src_window_image(f->d_buf[f->cur], f->buf_stride, in.dev, in.in_stride, a_in0, A0, j0, j1, f->d_buf[other], f->nchan, st));
    f->cur = other;
    f->b_current = b_current;
    f->b_end = b_end;
    f->last_position = input_index;
    f->last_ratio = src_ratio;
    if (in_used_out) *in_used_out = in_used;
    if (out_gen_out) *out_gen_out = out_gen;
    return 1;
}

// src_process + sinc_mono_vari_process; outputs land in d_out[nchan][out_stride]
static int src_process_impl(redio_src *f, const SrcInput &in, long input_frames, float *d_out, long out_stride, long output_frames,
                            double src_ratio_arg, int end_of_input, long *in_used_out, long *out_gen_out, hipStream_t st)
{
    if (is_bad_src_ratio(src_ratio_arg)) return REDIO_SRC_ERR_BAD_SRC_RATIO;
    if (input_frames < 0) input_frames = 0;
    if (output_frames < 0) output_frames = 0;
    if (f->last_ratio < (1.0 / SRC_MAX_RATIO)) f->last_ratio = src_ratio_arg;
    if (f->converter >= 3) {
        if (!in.dev && input_frames > 0) return REDIO_SRC_ERR_BAD_DATA_PTR;
        return zoh_linear_impl(f, in.dev, in.in_stride, input_frames, d_out, out_stride, output_frames, src_ratio_arg, in_used_out, out_gen_out, st);
    }

    const long in_count = input_frames, out_count = output_frames;
    long in_used = 0, out_gen = 0;
    double src_ratio = f->last_ratio;
    if (is_bad_src_ratio(src_ratio)) return REDIO_SRC_ERR_BAD_INTERNAL_STATE;
    { // can this call reach in front of the buffer image?  (the ratio fell since the last call: the filter is wider than the history kept)
        double cnt = (f->coeff_half_len + 2.0) / f->index_inc;
        const double mr = f->last_ratio < src_ratio_arg ? f->last_ratio : src_ratio_arg;
        if (mr < 1.0) cnt /= mr;
        const int half0 = (int)lrint(cnt) + 1;
        const double ii = f->last_position;
        const int bc = (f->b_current + (int)lrint(ii - fmod_one(ii))) % f->b_len;
        f->front_short = (bc != 0 || f->b_end != 0) && bc < half0;
    }
    if (f->window_ok && !f->front_short) {
        const int handled = try_uniform_window(f, in, in_count, d_out, out_stride, out_count, src_ratio_arg, end_of_input, in_used_out,
                                               out_gen_out, st);
        if (handled == 1) return REDIO_OK;
        if (handled != 0) return handled;
        const int general = try_general_window(f, in, in_count, d_out, out_stride, out_count, src_ratio_arg, end_of_input, in_used_out, out_gen_out, st);
        if (general == 1) return REDIO_OK;
        if (general != 0) return general;
    }
    int rc = ensure_scratch(f, (size_t)out_count);
    if (rc) return rc;
    f->arena_used = 0; // tables of the previous call: its uploads completed when that call synchronised
    // the scratch upload of an earlier call on another stream must not be overwritten while in
    // flight: calls on one handle are serialised by the caller (one block thread per handle)

    double count = (f->coeff_half_len + 2.0) / f->index_inc;
    const double minr = f->last_ratio < src_ratio_arg ? f->last_ratio : src_ratio_arg;
    if (minr < 1.0) count /= minr;
    const int half = (int)lrint(count) + 1;

    double input_index = f->last_position;
    double rem = fmod_one(input_index);
    f->b_current = (f->b_current + (int)lrint(input_index - rem)) % f->b_len;
    input_index = rem;
    const double terminate = 1.0 / src_ratio + 1e-20;
    const double fp_one = (double)(1 << SRC_SHIFT);

    // The per-output recurrence below is the library's, value for value; it runs once for ALL channels.  Two things are
    // rewritten without changing any result: x % b_len as a conditional subtraction (the operands are below 2 * b_len), and
    // the quantities that only depend on the ratio are formed once when the ratio does not vary inside the call (the
    // library recomputes the same expressions from the same operands for every output).
    const bool vary = out_count > 0 && fabs(f->last_ratio - src_ratio_arg) > 1e-10;
    auto wrap = [&](int x) { while (x >= f->b_len) x -= f->b_len; return x; };
    double float_increment = f->index_inc * (src_ratio < 1.0 ? src_ratio : 1.0);
    int inc_fp = (int)lrint(float_increment * fp_one);
    double scale = float_increment / f->index_inc;
    double step = 1.0 / src_ratio;
    long epoch_first = 0;
    while (out_gen < out_count) {
        int samples_in_hand = wrap(f->b_end - f->b_current + f->b_len);
        if (samples_in_hand <= half) {
            rc = flush_epoch(f, epoch_first, out_gen - epoch_first, d_out, out_stride, st); // before the image changes
            if (rc) return rc;
            epoch_first = out_gen;
            rc = prepare_data(f, in, in_count, in_used, end_of_input, half, st);
            if (rc) return rc;
            samples_in_hand = wrap(f->b_end - f->b_current + f->b_len);
            if (samples_in_hand <= half) break;
        }
        if (f->b_real_end >= 0) {
            if (f->b_current + input_index + terminate > f->b_real_end) break;
        }
        if (vary) {
            src_ratio = f->last_ratio + out_gen * (src_ratio_arg - f->last_ratio) / out_count;
            float_increment = f->index_inc * (src_ratio < 1.0 ? src_ratio : 1.0);
            inc_fp = (int)lrint(float_increment * fp_one);
            scale = float_increment / f->index_inc;
            step = 1.0 / src_ratio;
        }
        int start_fp = (int)lrint(input_index * float_increment * fp_one);
        int at = f->b_current;
        // a fractional position that rounds up to a whole increment is the next sample at phase zero: the two wing loops
        // then visit the same taps and the same samples in the same order (left: indices inc*(cc+1) .. 0 over x[at-cc .. at+1],
        // right: cc'*inc .. inc over x[at+1+cc' .. at+2]), so the output is the same bits; writing it that way keeps
        // rational ratios (1.5, 48000/44100 ...) exactly periodic for the periodic-phase kernel
        if (start_fp == inc_fp) { start_fp = 0; at += 1; }
        f->h_inc[(size_t)out_gen] = inc_fp;
        f->h_start[(size_t)out_gen] = start_fp;
        f->h_scale[(size_t)out_gen] = scale;
        f->h_pos[(size_t)out_gen] = at;
        ++out_gen;
        f->b_current = wrap(f->b_current + src_advance(input_index, step));
    }
    rc = flush_epoch(f, epoch_first, out_gen - epoch_first, d_out, out_stride, st);
    if (rc) return rc;
    f->last_position = input_index;
    f->last_ratio = src_ratio;
    if (in_used_out) *in_used_out = in_used;
    if (out_gen_out) *out_gen_out = out_gen;
    return REDIO_OK;
}

extern "C" int redio_src_process(redio_src *s, const void *d_in, long input_frames, long in_stride, void *d_out, long output_frames,
                                 long out_stride, double src_ratio, int end_of_input, long *input_frames_used,
                                 long *output_frames_gen, void *stream)
{
    if (input_frames_used) *input_frames_used = 0;
    if (output_frames_gen) *output_frames_gen = 0;
    if (!s) return REDIO_SRC_ERR_BAD_STATE;
    if ((!d_in && input_frames > 0) || (!d_out && output_frames > 0)) return REDIO_SRC_ERR_BAD_DATA_PTR; // an empty side may be NULL in the batched form
    SRC_TRY(hipSetDevice(s->device));
    SrcInput in = {nullptr, (const float *)d_in, in_stride};
    // pinned-free staging of the per-output parameters means the host arrays must stay untouched until
    // the uploads have run: synchronise at the end of the call (the uploads are tiny)
    s->zl_channels = 1; // rows are independent mono streams
    int rc = src_process_impl(s, in, input_frames, (float *)d_out, out_stride, output_frames, src_ratio, end_of_input,
                              input_frames_used, output_frames_gen, (hipStream_t)stream);
    hipError_t e = hipStreamSynchronize((hipStream_t)stream);
    if (rc) return rc;
    return hip_rc(e);
}

// host buffers, interleaved frames (nchan >= 1), synchronous: the body of the src_process drop-in (samplerate_shim.cpp).
// Channels are independent streams: the interleaved message is split into rows on the device, every row runs as
// the mono converter does, and the outputs are interleaved again.
extern "C" int redio_src_process_host(redio_src *s, const float *data_in, long input_frames, float *data_out, long output_frames,
                                      double src_ratio, int end_of_input, long *input_frames_used, long *output_frames_gen)
{
    if (input_frames_used) *input_frames_used = 0;
    if (output_frames_gen) *output_frames_gen = 0;
    if (!s) return REDIO_SRC_ERR_BAD_STATE;
    if (!data_in || !data_out) return REDIO_SRC_ERR_BAD_DATA_PTR;
    if (is_bad_src_ratio(src_ratio)) return REDIO_SRC_ERR_BAD_SRC_RATIO;
    if (input_frames < 0) input_frames = 0;
    if (output_frames < 0) output_frames = 0;
    const long nch = s->nchan;
    if (data_in < data_out) {
        if (data_in + input_frames * nch > data_out) return REDIO_SRC_ERR_DATA_OVERLAP;
    } else if (data_out + output_frames * nch > data_in) {
        return REDIO_SRC_ERR_DATA_OVERLAP;
    }
    SRC_TRY(hipSetDevice(s->device));
    if (!s->host_stream) SRC_TRY(hipStreamCreateWithFlags(&s->host_stream, hipStreamNonBlocking));
    hipStream_t st = s->host_stream;
    const size_t out_elems = (size_t)output_frames * nch, in_elems = (size_t)input_frames * nch;
    if (out_elems > s->stage_out_cap) {
        hipFree(s->d_stage_out);
        s->d_stage_out = nullptr; s->stage_out_cap = 0;
        SRC_TRY(hipMalloc((void **)&s->d_stage_out, (out_elems + 1024) * sizeof(float)));
        s->stage_out_cap = out_elems + 1024;
    }
    if (nch > 1 && out_elems > s->rows_out_cap) {
        hipFree(s->d_rows_out);
        s->d_rows_out = nullptr; s->rows_out_cap = 0;
        SRC_TRY(hipMalloc((void **)&s->d_rows_out, (out_elems + 1024) * sizeof(float)));
        s->rows_out_cap = out_elems + 1024;
    }
    SrcInput in = {nch == 1 ? data_in : nullptr, nullptr, 0};
    if (input_frames > 0) { // one upload of the whole message; the refills then copy on the device
        if (in_elems > s->stage_in_cap) {
            hipFree(s->d_stage_in);
            s->d_stage_in = nullptr; s->stage_in_cap = 0;
            SRC_TRY(hipMalloc((void **)&s->d_stage_in, (in_elems + 1024) * sizeof(float)));
            s->stage_in_cap = in_elems + 1024;
        }
        SRC_TRY(hipMemcpyAsync(s->d_stage_in, data_in, in_elems * sizeof(float), hipMemcpyHostToDevice, st));
        if (nch == 1) {
            in = {nullptr, s->d_stage_in, (long)s->stage_in_cap};
        } else {
            if (in_elems > s->rows_in_cap) {
                hipFree(s->d_rows_in);
                s->d_rows_in = nullptr; s->rows_in_cap = 0;
                SRC_TRY(hipMalloc((void **)&s->d_rows_in, (in_elems + 1024) * sizeof(float)));
                s->rows_in_cap = in_elems + 1024;
            }
            SRC_TRY(launch_src_interleave(s->d_stage_in, s->d_rows_in, input_frames, input_frames, (int)nch, true, st));
            in = {nullptr, s->d_rows_in, input_frames};
        }
    }
    long used = 0, gen = 0;
    float *d_rows = nch == 1 ? s->d_stage_out : s->d_rows_out;
    const long row_stride = nch == 1 ? (long)s->stage_out_cap : (output_frames > 0 ? output_frames : 1);
    s->zl_channels = (int)nch; // one interleaved multi-channel stream, the library's own counting
    int rc = src_process_impl(s, in, input_frames, d_rows, row_stride, output_frames, src_ratio, end_of_input, &used, &gen, st);
    hipError_t e = hipSuccess;
    if (rc == REDIO_OK && gen > 0) {
        if (nch > 1) e = launch_src_interleave(s->d_stage_out, s->d_rows_out, row_stride, gen, (int)nch, false, st);
        if (e == hipSuccess) e = hipMemcpyAsync(data_out, s->d_stage_out, (size_t)gen * nch * sizeof(float), hipMemcpyDeviceToHost, st);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (rc) return rc;
    if (e != hipSuccess) return hip_rc(e);
    if (input_frames_used) *input_frames_used = used;
    if (output_frames_gen) *output_frames_gen = gen;
    return REDIO_OK;
}
