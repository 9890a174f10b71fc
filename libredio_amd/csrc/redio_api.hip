// redio_api.hip -- host runtime behind include/redio.h: handles, plans, error mapping.  No compute
// happens on the host: every arithmetic entry point launches a gfx950 kernel or fails.
#include "../../include/redio.h"
#include "redio_internal.h"
#include <atomic>
#include <math.h>
#include <mutex>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace redio;

// ---------------------------------------------------------------- errors
static inline int hip_rc(hipError_t e) { return e == hipSuccess ? REDIO_OK : REDIO_ERR_HIP_BASE - (int)e; }
#define RD_TRY(expr)                         \
    do {                                     \
        hipError_t _e = (expr);              \
        if (_e != hipSuccess) return hip_rc(_e); \
    } while (0)

extern "C" const char *redio_strerror(int code)
{
    switch (code) {
    case REDIO_OK: return "ok";
    case REDIO_ERR_ARG: return "invalid argument";
    case REDIO_ERR_NOMEM: return "out of memory";
    case REDIO_ERR_UNSUPPORTED: return "shape not supported by any kernel";
    case REDIO_ERR_NO_DEVICE: return "no HIP device available (libredio has no CPU fallback)";
    case REDIO_ERR_ASSERT: return "the reference would have panicked on this input";
    case REDIO_ERR_NOT_RESERVED: return "plan scratch not reserved (call the plan's *_reserve before capturing a graph)";
    case REDIO_ERR_COMM: return "RCCL communicator error (librccl.so missing or a collective failed)";
    default: break;
    }
    if (code <= REDIO_ERR_HIP_BASE) return hipGetErrorString((hipError_t)(REDIO_ERR_HIP_BASE - code));
    if (code > 0 && code <= 22) return "libsamplerate-style converter error (see src_strerror in include/samplerate.h)";
    return "unknown redio error";
}
extern "C" const char *redio_version(void) { return "libredio 0.1 (gfx950)"; }

// ---------------------------------------------------------------- device helpers
extern "C" int redio_device_count(int *count)
{
    if (!count) return REDIO_ERR_ARG;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return REDIO_ERR_NO_DEVICE; }
    *count = n;
    return REDIO_OK;
}
extern "C" int redio_set_device(int device) { return hip_rc(hipSetDevice(device)); }
extern "C" int redio_get_device(int *device) { return device ? hip_rc(hipGetDevice(device)) : REDIO_ERR_ARG; }
static std::atomic<unsigned long long> g_malloc_count{0};
extern "C" int redio_malloc(void **dptr, size_t bytes)
{
    if (!dptr) return REDIO_ERR_ARG;
    g_malloc_count.fetch_add(1, std::memory_order_relaxed);
    return hip_rc(hipMalloc(dptr, bytes ? bytes : 1));
}
extern "C" unsigned long long redio_malloc_count(void) { return g_malloc_count.load(std::memory_order_relaxed); }
extern "C" int redio_free(void *dptr) { return dptr ? hip_rc(hipFree(dptr)) : REDIO_OK; }
extern "C" int redio_upload(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!bytes) return REDIO_OK;
    if (!dst || !src) return REDIO_ERR_ARG;
    return hip_rc(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
}
extern "C" int redio_download(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!bytes) return REDIO_OK;
    if (!dst || !src) return REDIO_ERR_ARG;
    return hip_rc(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
}
extern "C" int redio_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    if (!bytes) return REDIO_OK;
    if (!dst || !src) return REDIO_ERR_ARG;
    return hip_rc(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
}
extern "C" int redio_host_alloc(void **host, void **dev, size_t bytes)
{
    if (!host || !dev) return REDIO_ERR_ARG;
    *host = *dev = nullptr;
    void *h = nullptr, *d = nullptr;
    RD_TRY(hipHostMalloc(&h, bytes ? bytes : 1, hipHostMallocMapped));
    hipError_t e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) { hipHostFree(h); return hip_rc(e); }
    *host = h; *dev = d;
    return REDIO_OK;
}
extern "C" int redio_host_free(void *host) { return host ? hip_rc(hipHostFree(host)) : REDIO_OK; }

extern "C" int redio_stream_create(void **stream)
{
    if (!stream) return REDIO_ERR_ARG;
    hipStream_t s;
    RD_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return REDIO_OK;
}
extern "C" int redio_stream_destroy(void *stream) { return stream ? hip_rc(hipStreamDestroy((hipStream_t)stream)) : REDIO_OK; }
extern "C" int redio_stream_sync(void *stream) { return hip_rc(hipStreamSynchronize((hipStream_t)stream)); }
// A 32-bit value written to device-addressable memory (e.g. the device alias of a redio_host_alloc buffer) when the stream reaches this
// point: a host thread that polls the word learns that everything enqueued before has finished without entering hipStreamSynchronize --
// what the per-message drop-ins (kiss_fft, src_process) wait on.
extern "C" int redio_stream_signal(void *stream, void *d_word, uint32_t value)
{
    if (!d_word) return REDIO_ERR_ARG;
    return hip_rc(hipStreamWriteValue32((hipStream_t)stream, d_word, value, 0));
}
// ---- launch graphs: record the *_enqueue calls of a block pipeline once, replay them with one submission ----
struct redio_graph {
    hipGraph_t graph;
    hipGraphExec_t exec;
};
extern "C" int redio_graph_begin(void *stream)
{
    if (!stream) return REDIO_ERR_ARG; // the default stream cannot be captured
    return hip_rc(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
}
extern "C" int redio_graph_end(void *stream, redio_graph **g)
{
    if (!stream || !g) return REDIO_ERR_ARG;
    *g = nullptr;
    hipGraph_t graph = nullptr;
    RD_TRY(hipStreamEndCapture((hipStream_t)stream, &graph));
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e != hipSuccess) { hipGraphDestroy(graph); return hip_rc(e); }
    redio_graph *r = new (std::nothrow) redio_graph();
    if (!r) { hipGraphExecDestroy(exec); hipGraphDestroy(graph); return REDIO_ERR_NOMEM; }
    r->graph = graph; r->exec = exec;
    *g = r;
    return REDIO_OK;
}
extern "C" int redio_graph_launch(redio_graph *g, void *stream)
{
    if (!g) return REDIO_ERR_ARG;
    return hip_rc(hipGraphLaunch(g->exec, (hipStream_t)stream));
}
extern "C" int redio_graph_destroy(redio_graph *g)
{
    if (!g) return REDIO_OK;
    hipGraphExecDestroy(g->exec);
    hipGraphDestroy(g->graph);
    delete g;
    return REDIO_OK;
}

extern "C" int redio_event_create(void **event)
{
    if (!event) return REDIO_ERR_ARG;
    hipEvent_t e;
    RD_TRY(hipEventCreate(&e));
    *event = e;
    return REDIO_OK;
}
extern "C" int redio_event_create_sync(void **event)
{
    if (!event) return REDIO_ERR_ARG;
    hipEvent_t e;
    RD_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    *event = e;
    return REDIO_OK;
}
extern "C" int redio_stream_wait_event(void *stream, void *event)
{
    if (!event) return REDIO_ERR_ARG;
    return hip_rc(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)event, 0));
}
extern "C" int redio_event_sync(void *event) { return event ? hip_rc(hipEventSynchronize((hipEvent_t)event)) : REDIO_ERR_ARG; }
extern "C" int redio_event_destroy(void *event) { return event ? hip_rc(hipEventDestroy((hipEvent_t)event)) : REDIO_OK; }
extern "C" int redio_event_record(void *event, void *stream) { return hip_rc(hipEventRecord((hipEvent_t)event, (hipStream_t)stream)); }
extern "C" int redio_event_elapsed_ms(void *start, void *stop, float *ms)
{
    if (!ms) return REDIO_ERR_ARG;
    RD_TRY(hipEventSynchronize((hipEvent_t)stop));
    return hip_rc(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
}

static int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return -1;
    return d;
}

// ---------------------------------------------------------------- FIR plan
struct redio_fir {
    int device;
    size_t ntaps, decim;
    unsigned flags;
    float *d_taps;
};

extern "C" int redio_fir_create(redio_fir **h, const float *taps, size_t ntaps, size_t decim, unsigned flags)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (!taps) return REDIO_ERR_ARG;
    if (ntaps == 0 || decim == 0) return REDIO_ERR_ASSERT; // windows(0) panics in the reference
    if (ntaps > (1u << 24)) return REDIO_ERR_UNSUPPORTED;
    int dev = current_device();
    if (dev < 0) return REDIO_ERR_NO_DEVICE;
    redio_fir *p = new (std::nothrow) redio_fir();
    if (!p) return REDIO_ERR_NOMEM;
    p->device = dev; p->ntaps = ntaps; p->decim = decim; p->flags = flags; p->d_taps = nullptr;
    // pad the tap table to a multiple of 16 floats so wide scalar loads never run off the end
    size_t padded = (ntaps + 15) & ~(size_t)15;
    std::vector<float> tmp(padded, 0.0f);
    memcpy(tmp.data(), taps, ntaps * sizeof(float));
    hipError_t e = hipMalloc((void **)&p->d_taps, padded * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_taps, tmp.data(), padded * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (p->d_taps) hipFree(p->d_taps); delete p; return hip_rc(e); }
    *h = p;
    return REDIO_OK;
}
extern "C" int redio_fir_destroy(redio_fir *h)
{
    if (!h) return REDIO_OK;
    hipFree(h->d_taps);
    delete h;
    return REDIO_OK;
}
extern "C" size_t redio_fir_nout(const redio_fir *h, size_t n_in)
{
    if (!h || n_in < h->ntaps) return 0;
    return (n_in - h->ntaps) / h->decim + 1;
}
extern "C" int redio_fir_enqueue(redio_fir *h, const void *d_in, size_t n_in, void *d_out, void *stream)
{
    if (!h) return REDIO_ERR_ARG;
    size_t nout = redio_fir_nout(h, n_in);
    if (nout == 0) return REDIO_OK;
    if (!d_in || !d_out || d_in == d_out) return REDIO_ERR_ARG;
    RD_TRY(hipSetDevice(h->device));
    return hip_rc(launch_fir(d_in, (long)n_in, h->d_taps, (int)h->ntaps, (long)h->decim, d_out, (long)nout,
                             (h->flags & REDIO_FIR_COMPLEX) != 0, (h->flags & REDIO_FIR_FUSED) != 0, (hipStream_t)stream));
}

// ---------------------------------------------------------------- A1 host drop-in
// Per-thread state of the host-buffer convolve: the reference calls convolve once per message from a block thread
// (src/ratpak.rs:60-185: one OS thread per block), with the same taps every time, so the plan, a stream and the
// message buffers are kept per calling thread instead of being created and freed on every call.  Small messages go
// through pinned, device-addressable buffers (the kernel reads and writes them across PCIe itself); large ones
// through device buffers with explicit copies.  The cache lives as long as the thread's process (never freed at
// thread exit: the HIP runtime may already be shutting down there).
namespace {
struct ConvCache {
    int device = -1;
    hipStream_t stream = nullptr;
    redio_fir *plan = nullptr;
    std::vector<float> taps;
    float *pin_in = nullptr, *pin_in_dev = nullptr, *pin_out = nullptr, *pin_out_dev = nullptr; // zero-copy path
    size_t pin_cap = 0;
    float *d_in = nullptr, *d_out = nullptr; // copy path
    size_t d_cap = 0;
};
constexpr size_t CONV_ZERO_COPY_MAX = 1 << 16; // floats
} // namespace

extern "C" int redio_convolve_f32(const float *u, size_t nu, const float *v, size_t nv, float *out, size_t *nout)
{
    if (nout) *nout = 0;
    if (nv == 0) return REDIO_ERR_ASSERT;
    if (nu < nv) return REDIO_OK;
    if (!u || !v || !out) return REDIO_ERR_ARG;
    static thread_local ConvCache *cache = nullptr;
    int dev = current_device();
    if (dev < 0) return REDIO_ERR_NO_DEVICE;
    if (!cache) cache = new (std::nothrow) ConvCache();
    if (!cache) return REDIO_ERR_NOMEM;
    ConvCache &c = *cache;
    if (c.device != dev) { // first call of this thread, or the thread moved to another device: release what the
        if (c.device >= 0) { // old device holds (bound to it while freeing), then start over on the new one
            hipSetDevice(c.device);
            if (c.plan) redio_fir_destroy(c.plan);
            if (c.pin_in) hipHostFree(c.pin_in);
            if (c.pin_out) hipHostFree(c.pin_out);
            if (c.d_in) hipFree(c.d_in);
            if (c.d_out) hipFree(c.d_out);
            if (c.stream) hipStreamDestroy(c.stream);
            hipSetDevice(dev);
        }
        c = ConvCache();
        c.device = dev;
        RD_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    }
    if (!c.plan || c.taps.size() != nv || memcmp(c.taps.data(), v, nv * sizeof(float)) != 0) {
        if (c.plan) redio_fir_destroy(c.plan);
        c.plan = nullptr;
        int rc = redio_fir_create(&c.plan, v, nv, 1, 0);
        if (rc) return rc;
        c.taps.assign(v, v + nv);
    }
    const size_t n = nu - nv + 1;
    if (nu <= CONV_ZERO_COPY_MAX) {
        if (nu > c.pin_cap) {
            if (c.pin_in) hipHostFree(c.pin_in);
            if (c.pin_out) hipHostFree(c.pin_out);
            c.pin_in = c.pin_out = nullptr; c.pin_cap = 0;
            const size_t cap = nu < 4096 ? 4096 : nu;
            RD_TRY(hipHostMalloc((void **)&c.pin_in, cap * sizeof(float), hipHostMallocMapped));
            RD_TRY(hipHostMalloc((void **)&c.pin_out, cap * sizeof(float), hipHostMallocMapped));
            RD_TRY(hipHostGetDevicePointer((void **)&c.pin_in_dev, c.pin_in, 0));
            RD_TRY(hipHostGetDevicePointer((void **)&c.pin_out_dev, c.pin_out, 0));
            c.pin_cap = cap;
        }
        memcpy(c.pin_in, u, nu * sizeof(float));
        int rc = redio_fir_enqueue(c.plan, c.pin_in_dev, nu, c.pin_out_dev, c.stream);
        if (rc) return rc;
        RD_TRY(hipStreamSynchronize(c.stream));
        memcpy(out, c.pin_out, n * sizeof(float));
    } else {
        if (nu > c.d_cap) {
            if (c.d_in) hipFree(c.d_in);
            if (c.d_out) hipFree(c.d_out);
            c.d_in = c.d_out = nullptr; c.d_cap = 0;
            RD_TRY(hipMalloc((void **)&c.d_in, nu * sizeof(float)));
            RD_TRY(hipMalloc((void **)&c.d_out, nu * sizeof(float)));
            c.d_cap = nu;
        }
        RD_TRY(hipMemcpyAsync(c.d_in, u, nu * sizeof(float), hipMemcpyHostToDevice, c.stream));
        int rc = redio_fir_enqueue(c.plan, c.d_in, nu, c.d_out, c.stream);
        if (rc) return rc;
        RD_TRY(hipMemcpyAsync(out, c.d_out, n * sizeof(float), hipMemcpyDeviceToHost, c.stream));
        RD_TRY(hipStreamSynchronize(c.stream));
    }
    if (nout) *nout = n;
    return REDIO_OK;
}

// ---------------------------------------------------------------- FFT plan
struct redio_fft {
    int device;
    FftPlanDev dev;
    float2 *d_tw, *d_tw_pass;
    int *d_leaf, *d_leaf_pos;
    float2 *d_tmp; // for in-place calls on the global-memory path
    size_t tmp_elems;
};

extern "C" int redio_fft_create(redio_fft **h, int nfft, int inverse)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (nfft <= 0) return REDIO_ERR_ARG;
    if (nfft > (1 << 26)) return REDIO_ERR_UNSUPPORTED;
    int dev = current_device();
    if (dev < 0) return REDIO_ERR_NO_DEVICE;
    redio_fft *p = new (std::nothrow) redio_fft();
    if (!p) return REDIO_ERR_NOMEM;
    memset(&p->dev, 0, sizeof(p->dev));
    p->device = dev; p->d_tw = nullptr; p->d_tw_pass = nullptr; p->d_leaf = p->d_leaf_pos = nullptr; p->d_tmp = nullptr; p->tmp_elems = 0;
    p->dev.nfft = nfft; p->dev.inverse = inverse ? 1 : 0;
    p->dev.nstages = fft_plan_stages(nfft, p->dev.st, FFT_MAX_STAGES);
    if (p->dev.nstages < 0) { delete p; return REDIO_ERR_UNSUPPORTED; }
    // twiddles evaluated in double and rounded once, phase = -+ 2 pi i / nfft (kiss_fft_alloc)
    std::vector<float2> tw((size_t)nfft);
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    for (int i = 0; i < nfft; ++i) {
        double phase = -2 * pi * i / nfft;
        if (inverse) phase *= -1;
        tw[i] = make_float2((float)cos(phase), (float)sin(phase));
    }
    std::vector<int> leaf((size_t)nfft);
    for (int P = 0; P < nfft; ++P) leaf[P] = fft_leaf_source(P, p->dev.st, p->dev.nstages);
    std::vector<int> pos((size_t)nfft);
    for (int P = 0; P < nfft; ++P) pos[(size_t)leaf[P]] = P;
    // ceil(2^32 / d); 0 stands for d == 1 (the quotient is the dividend)
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned)d - 1) / (unsigned)d); };
    for (int i = 0; i < p->dev.nstages; ++i) p->dev.magic_m[i] = magic(p->dev.st[i].m);
    p->dev.magic_n = magic(nfft);
    hipError_t e = hipMalloc((void **)&p->d_tw, (size_t)nfft * sizeof(float2));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_leaf, (size_t)nfft * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_leaf_pos, (size_t)nfft * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(p->d_leaf_pos, pos.data(), (size_t)nfft * sizeof(int), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_tw, tw.data(), (size_t)nfft * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(p->d_leaf, leaf.data(), (size_t)nfft * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        if (p->d_tw) hipFree(p->d_tw);
        if (p->d_leaf) hipFree(p->d_leaf);
        if (p->d_leaf_pos) hipFree(p->d_leaf_pos);
        delete p;
        return hip_rc(e);
    }
    p->dev.tw = p->d_tw;
    p->dev.tw_pass = nullptr;
    bool smooth = (nfft & (nfft - 1)) != 0; // every radix-2/3/4/5 size that is not a power of two (compile-time kernels, tile passes)
    for (int i = 0; i < p->dev.nstages; ++i) smooth = smooth && p->dev.st[i].p <= 5;
    if (smooth) { // stage-ordered copy, FftCt<N>::toff: stage s holds T[(n - 1) m + k] = tw[n k fstride]
        std::vector<float2> ord;
        for (int i = 0; i < p->dev.nstages; ++i) {
            const FftStage &st = p->dev.st[i];
            for (int n = 1; n < st.p; ++n)
                for (int k = 0; k < st.m; ++k) ord.push_back(tw[(size_t)n * k * st.fstride]);
        }
        e = hipMalloc((void **)&p->d_tw_pass, ord.size() * sizeof(float2));
        if (e == hipSuccess) e = hipMemcpy(p->d_tw_pass, ord.data(), ord.size() * sizeof(float2), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            hipFree(p->d_tw); hipFree(p->d_leaf); hipFree(p->d_leaf_pos); hipFree(p->d_tw_pass);
            delete p;
            return hip_rc(e);
        }
        p->dev.tw_pass = p->d_tw_pass;
    } else if (const size_t ne = fftbig_tables_elems(nfft)) { // the multi-pass sizes read their twiddles in pass order
        e = hipMalloc((void **)&p->d_tw_pass, ne * sizeof(float2));
        if (e == hipSuccess) e = fftbig_tables_build(p->d_tw, p->d_tw_pass, nfft, nullptr);
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) {
            hipFree(p->d_tw); hipFree(p->d_leaf); hipFree(p->d_leaf_pos); hipFree(p->d_tw_pass);
            delete p;
            return hip_rc(e);
        }
        p->dev.tw_pass = p->d_tw_pass;
    }
    p->dev.leaf_src = p->d_leaf;
    p->dev.leaf_pos = p->d_leaf_pos;
    *h = p;
    return REDIO_OK;
}
extern "C" int redio_fft_destroy(redio_fft *h)
{
    if (!h) return REDIO_OK;
    hipFree(h->d_tw);
    if (h->d_tw_pass) hipFree(h->d_tw_pass);
    hipFree(h->d_leaf);
    hipFree(h->d_leaf_pos);
    if (h->d_tmp) hipFree(h->d_tmp);
    delete h;
    return REDIO_OK;
}
// sizes the staging buffer of the sizes that need one (in-place calls on the global-memory path, prime factors above 5
// beyond LDS) for up to nbatch messages per call; every other size never stages and needs no reservation
extern "C" int redio_fft_reserve(redio_fft *h, size_t nbatch)
{
    if (!h) return REDIO_ERR_ARG;
    const size_t need = 2 * nbatch * (size_t)h->dev.nfft;
    if (need <= h->tmp_elems) return REDIO_OK;
    RD_TRY(hipSetDevice(h->device));
    if (h->d_tmp) RD_TRY(hipFree(h->d_tmp));
    h->d_tmp = nullptr; h->tmp_elems = 0;
    RD_TRY(hipMalloc((void **)&h->d_tmp, need * sizeof(float2)));
    h->tmp_elems = need;
    return REDIO_OK;
}
// the retry of a launch that asked for staging: tmp = [input copy | work], grown on first use / growth only
static int fft_enqueue_staged(redio_fft *h, const float2 *d_in, float2 *d_out, size_t nbatch, long in_stride, hipStream_t st)
{
    const size_t need = nbatch * (size_t)h->dev.nfft;
    if (2 * need > h->tmp_elems) { // grown on first use unless redio_fft_reserve() sized it; never during graph capture
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return REDIO_ERR_NOT_RESERVED;
        int rc = redio_fft_reserve(h, nbatch);
        if (rc) return rc;
    }
    const float2 *src = d_in;
    if (d_in == d_out) { // only reachable with contiguous messages (in_stride == nfft)
        RD_TRY(hipMemcpyAsync(h->d_tmp, d_in, need * sizeof(float2), hipMemcpyDeviceToDevice, st));
        src = h->d_tmp;
    }
    hipError_t e = launch_fft(h->dev, src, d_out, (long)nbatch, st, in_stride, h->d_tmp + need);
    if (e == hipErrorNotSupported) return REDIO_ERR_UNSUPPORTED;
    return hip_rc(e);
}

extern "C" int redio_fft_enqueue(redio_fft *h, const void *d_in, void *d_out, size_t nbatch, void *stream)
{
    if (!h) return REDIO_ERR_ARG;
    if (nbatch == 0) return REDIO_OK;
    if (!d_in || !d_out) return REDIO_ERR_ARG;
    RD_TRY(hipSetDevice(h->device));
    hipError_t e = launch_fft(h->dev, (const float2 *)d_in, (float2 *)d_out, (long)nbatch, (hipStream_t)stream);
    // global-memory path in place, or a prime factor above 5 in a size that does not fit LDS: stage through the
    // plan-owned temporary (a caller that needs graph capture runs the sequence once before capturing)
    if (e == hipErrorNotSupported) return fft_enqueue_staged(h, (const float2 *)d_in, (float2 *)d_out, nbatch, 0, (hipStream_t)stream);
    return hip_rc(e);
}

// messages that start every in_stride samples (in_stride < nfft: overlapping blocks, as overlap-save
// needs); d_in and d_out must not alias
extern "C" int redio_fft_enqueue_strided(redio_fft *h, const void *d_in, void *d_out, size_t nbatch, long in_stride, void *stream)
{
    if (!h) return REDIO_ERR_ARG;
    if (nbatch == 0) return REDIO_OK;
    if (!d_in || !d_out || d_in == d_out || in_stride <= 0) return REDIO_ERR_ARG;
    RD_TRY(hipSetDevice(h->device));
    hipError_t e = launch_fft(h->dev, (const float2 *)d_in, (float2 *)d_out, (long)nbatch, (hipStream_t)stream, in_stride);
    if (e == hipErrorNotSupported) return fft_enqueue_staged(h, (const float2 *)d_in, (float2 *)d_out, nbatch, in_stride, (hipStream_t)stream);
    return hip_rc(e);
}

const FftPlanDev *redio_fft_plan_dev(const redio_fft *h) { return h ? &h->dev : nullptr; }
const float2 *redio_fft_twiddles_dev(const redio_fft *h) { return h ? h->dev.tw : nullptr; }
const float2 *redio_fft_twiddles_pass_dev(const redio_fft *h) { return h ? h->dev.tw_pass : nullptr; }

// ---------------------------------------------------------------- chain plan
struct redio_chain {
    redio_fir *fir;
    redio_fft *fft;
    int nfft;
    int fused_ok;
    int force_unfused;
    float2 *d_mid; // intermediate for the two-kernel path (redio_chain_reserve)
    size_t mid_elems;
    unsigned long long *d_stamps; // diagnostic per-wave stamps of THIS plan's launches (redio_chain_set_debug_stamps), else null
    size_t stamp_waves;           // records (4 x u64) that buffer holds: wavefronts beyond it leave no stamp
    char kernel_name[96];         // redio_chain_kernel_name
    float2 *d_conv; // converted samples for redio_chain_enqueue_u8 on shapes / pointers without the one-kernel form (grown on first use)
    size_t conv_elems;
};

extern "C" int redio_chain_create(redio_chain **h, const float *taps, size_t ntaps, size_t decim, int nfft, unsigned flags)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    redio_chain *c = new (std::nothrow) redio_chain();
    if (!c) return REDIO_ERR_NOMEM;
    memset(c, 0, sizeof(*c));
    int rc = redio_fir_create(&c->fir, taps, ntaps, decim, flags | REDIO_FIR_COMPLEX);
    if (rc == REDIO_OK) rc = redio_fft_create(&c->fft, nfft, 0);
    if (rc) { redio_fir_destroy(c->fir); redio_fft_destroy(c->fft); delete c; return rc; }
    c->nfft = nfft;
    c->fused_ok = chain_supported((int)ntaps, (long)decim, nfft) ? 1 : 0;
    *h = c;
    return REDIO_OK;
}
extern "C" int redio_chain_destroy(redio_chain *h)
{
    if (!h) return REDIO_OK;
    redio_fir_destroy(h->fir);
    redio_fft_destroy(h->fft);
    if (h->d_mid) hipFree(h->d_mid);
    if (h->d_conv) hipFree(h->d_conv);
    delete h;
    return REDIO_OK;
}
extern "C" size_t redio_chain_nblocks(const redio_chain *h, size_t n_in)
{
    if (!h) return 0;
    return redio_fir_nout(h->fir, n_in) / (size_t)h->nfft;
}
extern "C" int redio_chain_is_fused(const redio_chain *h) { return h && h->fused_ok && !h->force_unfused; }
extern "C" int redio_chain_set_unfused(redio_chain *h, int unfused)
{
    if (!h) return REDIO_ERR_ARG;
    h->force_unfused = unfused ? 1 : 0;
    return REDIO_OK;
}
extern "C" int redio_chain_set_debug_stamps(redio_chain *h, void *d_buf, size_t capacity_waves)
{
    if (!h || (d_buf && capacity_waves == 0)) return REDIO_ERR_ARG;
    h->d_stamps = (unsigned long long *)d_buf;
    h->stamp_waves = d_buf ? capacity_waves : 0;
    return REDIO_OK;
}
// launch geometry of the fused kernel for a call that yields nblocks blocks: consecutive blocks per wavefront, and the number of
// wavefronts (= workgroups) of the launch; 0 when the plan runs as two kernels
extern "C" size_t redio_chain_blocks_per_wave(const redio_chain *h, size_t nblocks)
{
    if (!redio_chain_is_fused(h) || nblocks == 0) return 0;
    if (hipSetDevice(h->fir->device) != hipSuccess) return 0; // the rule depends on the device's CU count
    return (size_t)chain_v4_blocks_per_wave((long)nblocks);
}
extern "C" size_t redio_chain_launch_waves(const redio_chain *h, size_t nblocks)
{
    const size_t bpw = redio_chain_blocks_per_wave(h, nblocks);
    return bpw ? (nblocks + bpw - 1) / bpw : 0;
}
// name of the fused kernel this plan launches for cf32 input, as rocprofv3 reports it with the spaces removed; NULL for a two-kernel plan
extern "C" const char *redio_chain_kernel_name(redio_chain *h)
{
    if (!redio_chain_is_fused(h)) return nullptr;
    return chain_kernel_name((int)h->fir->ntaps, (long)h->fir->decim, (h->fir->flags & REDIO_FIR_FUSED) != 0, h->kernel_name, sizeof(h->kernel_name));
}
// sizes the two-kernel path's intermediate for inputs of up to n_in samples (allocation; may free a smaller one)
extern "C" int redio_chain_reserve(redio_chain *h, size_t n_in)
{
    if (!h) return REDIO_ERR_ARG;
    const size_t ny = redio_chain_nblocks(h, n_in) * (size_t)h->nfft;
    if (ny <= h->mid_elems) return REDIO_OK;
    RD_TRY(hipSetDevice(h->fir->device));
    if (h->d_mid) RD_TRY(hipFree(h->d_mid));
    h->d_mid = nullptr; h->mid_elems = 0;
    RD_TRY(hipMalloc((void **)&h->d_mid, ny * sizeof(float2)));
    h->mid_elems = ny;
    return REDIO_OK;
}
static bool stream_is_capturing(hipStream_t st)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    return hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
}
extern "C" int redio_chain_enqueue(redio_chain *h, const void *d_in, size_t n_in, void *d_out, void *stream)
{
    if (!h) return REDIO_ERR_ARG;
    size_t nblk = redio_chain_nblocks(h, n_in);
    if (nblk == 0) return REDIO_OK;
    if (!d_in || !d_out || d_in == d_out) return REDIO_ERR_ARG;
    RD_TRY(hipSetDevice(h->fir->device));
    const bool fused_math = (h->fir->flags & REDIO_FIR_FUSED) != 0;
    if (redio_chain_is_fused(h)) {
        hipError_t e = launch_chain(h->fft->dev, (const float2 *)d_in, (long)n_in, h->fir->d_taps, (int)h->fir->ntaps,
                                    (long)h->fir->decim, (float2 *)d_out, (long)nblk, fused_math, (hipStream_t)stream, h->d_stamps, (long)h->stamp_waves);
        if (e != hipErrorNotSupported) return hip_rc(e);
        // e.g. an input pointer the fused kernel cannot take: same results through the two kernels below
    }
    // two kernels through the plan-owned intermediate.  Sized by redio_chain_reserve(); an un-reserved plan grows it
    // here on first use -- an allocation, so never while the stream is being captured into a graph.
    size_t ny = nblk * (size_t)h->nfft;
    if (ny > h->mid_elems) {
        if (stream_is_capturing((hipStream_t)stream)) return REDIO_ERR_NOT_RESERVED;
        int rc = redio_chain_reserve(h, n_in);
        if (rc) return rc;
    }
    size_t need_in = (ny - 1) * h->fir->decim + h->fir->ntaps; // inputs feeding the kept blocks
    RD_TRY(launch_fir(d_in, (long)need_in, h->fir->d_taps, (int)h->fir->ntaps, (long)h->fir->decim, h->d_mid, (long)ny,
                      true, fused_math, (hipStream_t)stream));
    return redio_fft_enqueue(h->fft, h->d_mid, d_out, nblk, stream);
}

// sizes what redio_chain_enqueue_u8 needs beyond the one-kernel form for messages of up to nbytes bytes: the converted samples,
// and the two-kernel path's intermediate behind them
extern "C" int redio_chain_reserve_u8(redio_chain *h, size_t nbytes)
{
    if (!h) return REDIO_ERR_ARG;
    const size_t n_in = nbytes / 2;
    RD_TRY(hipSetDevice(h->fir->device));
    if (n_in > h->conv_elems) {
        if (h->d_conv) RD_TRY(hipFree(h->d_conv));
        h->d_conv = nullptr; h->conv_elems = 0;
        RD_TRY(hipMalloc((void **)&h->d_conv, n_in * sizeof(float2)));
        h->conv_elems = n_in;
    }
    return redio_chain_reserve(h, n_in);
}

// rtlsdr::data_to_samples (rtlsdr.rs:159-162) -> the chain, from the receiver's u8 I/Q bytes
extern "C" int redio_chain_enqueue_u8(redio_chain *h, const void *d_bytes, size_t nbytes, void *d_out, void *stream)
{
    if (!h || (nbytes & 1)) return REDIO_ERR_ARG;
    const size_t n_in = nbytes / 2;
    const size_t nblk = redio_chain_nblocks(h, n_in);
    if (nblk == 0) return REDIO_OK;
    if (!d_bytes || !d_out || d_bytes == d_out) return REDIO_ERR_ARG;
    RD_TRY(hipSetDevice(h->fir->device));
    if (redio_chain_is_fused(h)) {
        hipError_t e = launch_chain_u8(h->fft->dev, d_bytes, h->fir->d_taps, (int)h->fir->ntaps, (long)h->fir->decim, (float2 *)d_out, (long)nblk,
                                       (h->fir->flags & REDIO_FIR_FUSED) != 0, (hipStream_t)stream);
        if (e != hipErrorNotSupported) return hip_rc(e);
    }
    // other shapes, or bytes that are not 4-byte aligned: convert into a plan-owned buffer, then the cf32 entry point (same results)
    if (n_in > h->conv_elems) { // un-reserved (redio_chain_reserve_u8): grow on first use, never inside a capture
        if (stream_is_capturing((hipStream_t)stream)) return REDIO_ERR_NOT_RESERVED;
        RD_TRY(hipStreamSynchronize((hipStream_t)stream)); // launches that still read the old buffer
        int rc = redio_chain_reserve_u8(h, nbytes);
        if (rc) return rc;
    }
    int rc = redio_data_to_samples(d_bytes, nbytes, h->d_conv, stream);
    if (rc) return rc;
    return redio_chain_enqueue(h, h->d_conv, n_in, d_out, stream);
}

// shapes of the plans, for the carried-history layer (stream_carry.hip)
void redio_fir_shape(const redio_fir *h, size_t *ntaps, size_t *decim, unsigned *flags, int *device)
{
    *ntaps = h->ntaps; *decim = h->decim; *flags = h->flags; *device = h->device;
}
void redio_chain_shape(const redio_chain *h, size_t *ntaps, size_t *decim, int *nfft, int *device)
{
    *ntaps = h->fir->ntaps; *decim = h->fir->decim; *nfft = h->nfft; *device = h->fir->device;
}

// ---------------------------------------------------------------- synthetic input
extern "C" int redio_synth_iq(void *d_out, uint32_t seed, uint64_t first, size_t n, void *stream)
{
    if (n && !d_out) return REDIO_ERR_ARG;
    return hip_rc(launch_synth_iq((float2 *)d_out, seed, first, (long)n, (hipStream_t)stream));
}
extern "C" int redio_synth_f32(void *d_out, uint32_t seed, uint64_t first, size_t n, void *stream)
{
    if (n && !d_out) return REDIO_ERR_ARG;
    return hip_rc(launch_synth_f32((float *)d_out, seed, first, (long)n, (hipStream_t)stream));
}

// test hook (not in include/redio.h): the gfx950 buffer range-check rule the u8 channelizer's row-pair loads rely on
namespace redio { hipError_t launch_buffer_load_probe(const void *, uint32_t, uint32_t, uint32_t, void *, hipStream_t); }
extern "C" int redio_debug_buffer_load_probe(const void *d_base, uint32_t num_records, uint32_t voffset, uint32_t soffset, void *d_out64, void *stream)
{
    if (!d_base || !d_out64) return REDIO_ERR_ARG;
    return hip_rc(launch_buffer_load_probe(d_base, num_records, voffset, soffset, d_out64, (hipStream_t)stream));
}
