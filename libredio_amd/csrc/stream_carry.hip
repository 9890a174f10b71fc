// stream_carry.hip -- carried history across calls for the windowed plans (include/redio.h, redio_*_stream_*).
//
// The reference's convolve is stateless per message (src/dsputils/src/dsputils.rs:30-32) and so loses K-1 outputs at
// every message seam; BASELINE.json configs[1] is defined on a STREAM ("history carried")
// and the results must not depend on how the stream was cut into messages.  Every windowed plan
// here has the same shape: unit u of the output needs the W input samples that start at u*H:
//     FIR          W = ntaps                       H = decim          unit = 1 output sample
//     chain        W = (nfft-1)*decim + ntaps      H = nfft*decim     unit = 1 spectrum (nfft samples)
//     channelizer  W = nchan*taps_per_branch       H = nchan          unit = 1 row (nchan samples)
//     overlap-save W = nfft                        H = hop            unit = hop output samples
// so one layer serves them all.  The handle keeps the stream's unconsumed tail (fewer than W samples) on the device.
// A call with n new samples
//   1. appends the first min(n, W-1) new samples to the tail in a plan-owned staging buffer (one small copy),
//   2. runs the plan's stateless kernel on the staging buffer for the units that START in the tail ("head"),
//   3. runs it on the caller's buffer, in place and without copying it, for every unit that starts in the new data,
//   4. keeps the new tail (one small copy).
// Every unit is computed by the same kernel from the same W samples whatever the segmentation, so a stream fed in any
// pieces yields the bits of the one-shot call.  All counters are host-side and advance at enqueue time: feed one
// stream from one thread on one HIP stream (or order the HIP streams yourself).  Nothing synchronises.
#include "../../include/redio.h"
#include "redio_internal.h"
#include "stream_split.h"
#include <atomic>
#include <new>

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? REDIO_OK : REDIO_ERR_HIP_BASE - (int)e; }
#define SC_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return hip_rc(_e); } while (0)

namespace {
// The seam copies (at most W - 1 samples: 42 KB for the chain) as a kernel of this library instead of a device-to-device hipMemcpyAsync.
// Measured (round 6, tools/carry_ab.sh: the chain as a stream block, messages of 2^13 ... 2^20 samples): 19.1 against 19.4 us per message,
// 23.3 against 23.7 -- 1-2 %: a call's cost at these sizes is the in-order execution of its four dependent operations on the stream (seam copy,
// the head's launch, the body's launch, tail copy: ~5 us each), not their enqueue.  Kept because it is never slower and takes the copy engines and
// their blit path out of the picture.  The widest access the three alignments allow, one element per thread.
template <typename V>
__global__ __launch_bounds__(256) void seam_copy_kernel(V *__restrict__ dst, const V *__restrict__ src, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
hipError_t seam_copy(void *dst, const void *src, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return hipSuccess;
    const uintptr_t a = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes;
#define SEAM_GO(V)                                                                                                               \
    {                                                                                                                            \
        const size_t n = bytes / sizeof(V);                                                                                      \
        hipLaunchKernelGGL(seam_copy_kernel<V>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (V *)dst, (const V *)src, n); \
        return hipGetLastError();                                                                                                \
    }
    if ((a & 15) == 0) SEAM_GO(uint4)
    if ((a & 7) == 0) SEAM_GO(uint2)
    if ((a & 3) == 0) SEAM_GO(uint32_t)
    if ((a & 1) == 0) SEAM_GO(uint16_t)
    SEAM_GO(uint8_t)
#undef SEAM_GO
}
enum Kind { K_FIR, K_CHAIN, K_PFB, K_OVSAVE, K_CHAIN_U8, K_PFB_U8 }; // _U8: the samples are interleaved u8 I/Q byte pairs
struct Carry {
    int device = 0;
    Kind kind = K_FIR;
    void *plan = nullptr;                 // not owned: the plan must outlive the stream handle
    size_t W = 0, H = 0;                  // window and hop, in input samples
    size_t in_elem = 0;                   // bytes per input sample
    size_t unit_out = 0, out_elem = 0;    // output samples per unit, bytes per output sample
    char *d_s[2] = {nullptr, nullptr};    // staging buffers, 2*W input samples each
    int cur = 0;
    size_t off = 0;                       // where the tail starts inside d_s[cur], in samples (off * in_elem is a multiple of 16)
    size_t hist = 0;                      // samples of the stream's tail held in d_s[cur] from `off` on
    size_t skip = 0;                      // H > W only: samples still to be dropped before the next unit starts (then hist == 0)
    unsigned long long total_in = 0, total_units = 0;
    std::atomic_flag busy = ATOMIC_FLAG_INIT; // one thread at a time inside enqueue / reset
};
struct Entered { // holds Carry::busy for the life of a call
    std::atomic_flag *f;
    explicit Entered(Carry *c) : f(c->busy.test_and_set(std::memory_order_acquire) ? nullptr : &c->busy) {}
    ~Entered() { if (f) f->clear(std::memory_order_release); }
    bool ok() const { return f != nullptr; }
};

// what a call with n new samples does, given c.hist carried ones (stream_split.h: pure host arithmetic, CPU-tested)
using redio::StreamSplit;
static StreamSplit split(const Carry &c, size_t n) { return redio::stream_split(c.hist, c.W, c.H, n); }

int run(const Carry &c, const void *d_in, size_t n_in, void *d_out, void *stream)
{
    switch (c.kind) {
    case K_FIR: return redio_fir_enqueue((redio_fir *)c.plan, d_in, n_in, d_out, stream);
    case K_CHAIN: return redio_chain_enqueue((redio_chain *)c.plan, d_in, n_in, d_out, stream);
    case K_PFB: return redio_pfb_enqueue((redio_pfb *)c.plan, d_in, n_in, d_out, 1, stream);
    case K_OVSAVE: return redio_ovsave_enqueue((redio_ovsave *)c.plan, d_in, n_in, d_out, stream);
    case K_CHAIN_U8: return redio_chain_enqueue_u8((redio_chain *)c.plan, d_in, 2 * n_in, d_out, stream);
    case K_PFB_U8: return redio_pfb_enqueue_u8((redio_pfb *)c.plan, d_in, 2 * n_in, d_out, 1, stream);
    }
    return REDIO_ERR_ARG;
}

int carry_create(Carry **out, Kind kind, void *plan, int device, size_t W, size_t H, size_t in_elem, size_t unit_out, size_t out_elem)
{
    *out = nullptr;
    if (!plan || W == 0 || H == 0) return REDIO_ERR_ARG;
    Carry *c = new (std::nothrow) Carry();
    if (!c) return REDIO_ERR_NOMEM;
    c->device = device; c->kind = kind; c->plan = plan; c->W = W; c->H = H; c->in_elem = in_elem; c->unit_out = unit_out; c->out_elem = out_elem;
    hipError_t e = hipSetDevice(device);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) e = hipMalloc((void **)&c->d_s[i], 2 * W * in_elem);
    if (e != hipSuccess) {
        for (int i = 0; i < 2; ++i) if (c->d_s[i]) hipFree(c->d_s[i]);
        delete c;
        return hip_rc(e);
    }
    *out = c;
    return REDIO_OK;
}
int carry_destroy(Carry *c)
{
    if (!c) return REDIO_OK;
    for (int i = 0; i < 2; ++i) if (c->d_s[i]) hipFree(c->d_s[i]);
    delete c;
    return REDIO_OK;
}
size_t carry_nout(const Carry *c, size_t n)
{
    if (!c) return 0;
    const size_t drop = c->skip < n ? c->skip : n;
    const StreamSplit s = split(*c, n - drop);
    return (s.nh + s.nb) * c->unit_out;
}
int carry_enqueue(Carry *c, const void *d_new, size_t n, void *d_out, size_t *nout, void *stream)
{
    if (nout) *nout = 0;
    if (!c) return REDIO_ERR_ARG;
    const Entered guard(c);
    if (!guard.ok()) return REDIO_ERR_ARG; // another thread is inside this stream: its counters are not ours to move
    if (n == 0) return REDIO_OK;
    if (!d_new) return REDIO_ERR_ARG;
    SC_TRY(hipSetDevice(c->device));
    hipStream_t st = (hipStream_t)stream;
    const char *src = (const char *)d_new;
    const size_t n_call = n;
    const size_t drop = c->skip < n ? c->skip : n; // a hop longer than the window: the samples between two windows are never read
    src += drop * c->in_elem; n -= drop;
    const StreamSplit s = n ? split(*c, n) : StreamSplit{0, 0, 0, 0, 0};
    if ((s.nh || s.nb) && !d_out) return REDIO_ERR_ARG;
    if (n == 0) { c->skip -= drop; c->total_in += n_call; return REDIO_OK; }
    // Everything up to the last launch leaves the handle's state alone (the seam copy below only writes staging memory behind
    // the carried tail, and writes the same bytes again if the call is repeated); the state is committed at the end.
    const size_t m = n < c->W - 1 ? n : c->W - 1;
    // The tail may sit at an offset inside its staging buffer (below: a small call that consumes whole units out of [tail | new] leaves the
    // new tail where it lies instead of copying it to the other buffer's front).  When this call's samples no longer fit behind it, the tail
    // moves to the other buffer's front first; that only re-seats the same samples, so it is committed at once.
    if (c->hist > 0 && c->off + c->hist + m > 2 * c->W) {
        SC_TRY(seam_copy(c->d_s[c->cur ^ 1], c->d_s[c->cur] + c->off * c->in_elem, c->hist * c->in_elem, st));
        c->cur ^= 1;
        c->off = 0;
    }
    char *S = c->d_s[c->cur] + c->off * c->in_elem, *T = c->d_s[c->cur ^ 1];
    if (c->hist > 0 && m > 0) SC_TRY(seam_copy(S + c->hist * c->in_elem, src, m * c->in_elem, st));
    char *out = (char *)d_out;
    if (s.nh) {
        const int rc = run(*c, S, s.head_in, out, stream);
        if (rc) return rc;
    }
    if (s.nb) {
        const int rc = run(*c, src + s.off * c->in_elem, s.body_in, out + s.nh * c->unit_out * c->out_elem, stream);
        if (rc) return rc;
    }
    // the new tail: everything from the start of the first unit not yet produced
    const size_t consumed = (s.nh + s.nb) * c->H; // staging coordinates ([tail | new])
    size_t new_hist = 0, new_skip = c->skip - drop, new_off = 0;
    bool flip = false;
    if (consumed >= c->hist + n) { // only when H > W: the next unit starts beyond what has arrived
        new_skip = consumed - (c->hist + n);
    } else {
        new_hist = c->hist + n - consumed;
        if (consumed >= c->hist) { // lies entirely in the caller's buffer
            if (new_hist) SC_TRY(seam_copy(T, src + (consumed - c->hist) * c->in_elem, new_hist * c->in_elem, st));
            flip = true;
        } else if (c->hist == 0) { // a first piece shorter than a window
            SC_TRY(seam_copy(T, src, n * c->in_elem, st));
            flip = true;
        } else if ((consumed * c->in_elem) % 16 == 0) {
            // starts inside the old tail: then n < W - 1 and the staging buffer holds all of [tail | new] -- the new tail stays where it is
            // (round 6: one dependent operation less per small message; 16-byte alignment of the next call's window kept)
            new_off = c->off + consumed;
        } else {
            SC_TRY(seam_copy(T, S + consumed * c->in_elem, new_hist * c->in_elem, st));
            flip = true;
        }
    }
    // commit
    if (flip) c->cur ^= 1;
    c->off = new_off;
    c->hist = new_hist;
    c->skip = new_skip;
    c->total_in += n_call;
    c->total_units += s.nh + s.nb;
    if (nout) *nout = (s.nh + s.nb) * c->unit_out;
    return REDIO_OK;
}
int carry_reset(Carry *c)
{
    const Entered guard(c);
    if (!guard.ok()) return REDIO_ERR_ARG;
    c->hist = 0; c->off = 0; c->skip = 0; c->total_in = 0; c->total_units = 0;
    return REDIO_OK;
}
} // namespace

struct redio_fir_stream { Carry *c; };
struct redio_chain_stream { Carry *c; };
struct redio_pfb_stream { Carry *c; };
struct redio_ovsave_stream { Carry *c; };

#define RD_STREAM_API(NAME)                                                                                                     \
    extern "C" int redio_##NAME##_stream_destroy(redio_##NAME##_stream *h)                                                      \
    {                                                                                                                           \
        if (!h) return REDIO_OK;                                                                                                \
        carry_destroy(h->c);                                                                                                    \
        delete h;                                                                                                               \
        return REDIO_OK;                                                                                                        \
    }                                                                                                                           \
    extern "C" int redio_##NAME##_stream_reset(redio_##NAME##_stream *h)                                                        \
    {                                                                                                                           \
        if (!h) return REDIO_ERR_ARG;                                                                                           \
        return carry_reset(h->c);                                                                                               \
    }                                                                                                                           \
    extern "C" size_t redio_##NAME##_stream_nout(const redio_##NAME##_stream *h, size_t n_new) { return h ? carry_nout(h->c, n_new) : 0; } \
    extern "C" size_t redio_##NAME##_stream_pending(const redio_##NAME##_stream *h) { return h ? h->c->hist : 0; }              \
    extern "C" int redio_##NAME##_stream_enqueue(redio_##NAME##_stream *h, const void *d_new, size_t n_new, void *d_out, size_t *nout, void *stream) \
    {                                                                                                                           \
        if (!h) return REDIO_ERR_ARG;                                                                                           \
        return carry_enqueue(h->c, d_new, n_new, d_out, nout, stream);                                                          \
    }
RD_STREAM_API(fir)
RD_STREAM_API(chain)
RD_STREAM_API(pfb)
RD_STREAM_API(ovsave)

template <typename Hd>
static int make(Hd **h, Kind kind, void *plan, int dev, size_t W, size_t H, size_t in_elem, size_t unit_out, size_t out_elem)
{
    Carry *c = nullptr;
    const int rc = carry_create(&c, kind, plan, dev, W, H, in_elem, unit_out, out_elem);
    if (rc) return rc;
    Hd *p = new (std::nothrow) Hd();
    if (!p) { carry_destroy(c); return REDIO_ERR_NOMEM; }
    p->c = c;
    *h = p;
    return REDIO_OK;
}

extern "C" int redio_fir_stream_create(redio_fir_stream **h, redio_fir *plan)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (!plan) return REDIO_ERR_ARG;
    size_t K, D; unsigned flags; int dev;
    redio_fir_shape(plan, &K, &D, &flags, &dev);
    const size_t el = (flags & REDIO_FIR_COMPLEX) ? 8 : 4;
    return make(h, K_FIR, plan, dev, K, D, el, 1, el);
}
extern "C" int redio_chain_stream_create(redio_chain_stream **h, redio_chain *plan)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (!plan) return REDIO_ERR_ARG;
    size_t K, D; int nfft, dev;
    redio_chain_shape(plan, &K, &D, &nfft, &dev);
    // the two-kernel path of the plan (unfused shapes, odd sample offsets) needs a plan-owned intermediate: sized here for the seam
    // windows (at most 2*W samples per head run); a body run longer than any earlier one on that path grows it at enqueue time
    // (redio.h: reserve with redio_chain_reserve(plan, largest message) to keep enqueue allocation-free)
    const size_t W = ((size_t)nfft - 1) * D + K;
    const int rr = redio_chain_reserve(plan, 2 * W);
    if (rr) return rr;
    return make(h, K_CHAIN, plan, dev, W, (size_t)nfft * D, 8, (size_t)nfft, 8);
}
// the same streams fed with the receiver's u8 I/Q bytes (2 bytes per sample; n_new still counts SAMPLES): the history is carried
// as bytes, and every window runs redio_chain_enqueue_u8 / redio_pfb_enqueue_u8
extern "C" int redio_chain_stream_create_u8(redio_chain_stream **h, redio_chain *plan)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (!plan) return REDIO_ERR_ARG;
    size_t K, D; int nfft, dev;
    redio_chain_shape(plan, &K, &D, &nfft, &dev);
    const size_t W = ((size_t)nfft - 1) * D + K;
    const int rr = redio_chain_reserve_u8(plan, 2 * (2 * W)); // seam windows; see redio_chain_stream_create
    if (rr) return rr;
    return make(h, K_CHAIN_U8, plan, dev, W, (size_t)nfft * D, 2, (size_t)nfft, 8);
}
extern "C" int redio_pfb_stream_create_u8(redio_pfb_stream **h, redio_pfb *plan)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (!plan) return REDIO_ERR_ARG;
    int M, P, dev;
    redio_pfb_shape(plan, &M, &P, &dev);
    return make(h, K_PFB_U8, plan, dev, (size_t)M * P, (size_t)M, 2, (size_t)M, 8);
}
extern "C" int redio_pfb_stream_create(redio_pfb_stream **h, redio_pfb *plan)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (!plan) return REDIO_ERR_ARG;
    int M, P, dev;
    redio_pfb_shape(plan, &M, &P, &dev);
    return make(h, K_PFB, plan, dev, (size_t)M * P, (size_t)M, 8, (size_t)M, 8);
}
extern "C" int redio_ovsave_stream_create(redio_ovsave_stream **h, redio_ovsave *plan)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (!plan) return REDIO_ERR_ARG;
    int nfft, dev; size_t hop;
    redio_ovsave_shape(plan, &nfft, &hop, &dev);
    return make(h, K_OVSAVE, plan, dev, (size_t)nfft, hop, 8, hop, 8);
}
