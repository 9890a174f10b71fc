// kpn_dev.hpp -- device-resident messages for kpn graphs: the steps either side of
// every hot block without host round trips.  A message is a kpn::dev::View = shared ownership of a
// device allocation + (offset, length) in samples, so the reshaping blocks of src/kpn/src/kpn.rs become
// metadata operations:
//   fork (kpn.rs:182-189)            clones the handle, not the samples (kpn::fork works unchanged on Views)
//   shaper (kpn.rs:278-282)          dev::shaper: re-chunks a stream of Views into Views of length l,
//                                    zero-copy when a chunk lies inside one allocation
//   unpacketizer / shaper_vecs       the inverse is the identity on a View stream
// and the hot blocks consume and produce Views through the device plans of include/redio.h.  Channels are the
// same FIFOs; use kpn::bounded_channel<View<T>>(n) between device blocks to cap the HBM a fast producer can pin
// (the reference's unbounded mpsc has no back-pressure -- a documented deviation).
#pragma once
#include "kpn.hpp"
#include <complex>
#include <memory>
#include <vector>

namespace kpn {
namespace dev {

struct Alloc {
    void *ptr = nullptr;
    size_t bytes = 0;
    explicit Alloc(size_t b) : bytes(b)
    {
        int rc = redio_malloc(&ptr, b ? b : 1);
        if (rc != REDIO_OK) throw std::runtime_error(redio_strerror(rc));
    }
    ~Alloc() { redio_free(ptr); }
    Alloc(const Alloc &) = delete;
    Alloc &operator=(const Alloc &) = delete;
};

template <typename T>
struct View {
    std::shared_ptr<Alloc> mem;
    size_t off = 0, len = 0; // in elements of T
    T *data() const { return reinterpret_cast<T *>(mem->ptr) + off; }
    View sub(size_t o, size_t n) const { return View{mem, off + o, n}; }
};

template <typename T>
View<T> make(size_t n) { return View<T>{std::make_shared<Alloc>(n * sizeof(T)), 0, n}; }

inline void check(int rc)
{
    if (rc != REDIO_OK) throw std::runtime_error(redio_strerror(rc));
}

// Every block owns a HIP stream for its life (one OS thread per block, src/ratpak.rs:60-185): blocks on
// different threads then overlap on the device instead of serialising on the default stream.  A block
// synchronises its stream before it sends a message, so ordering between blocks is the channel's.
struct BlockStream {
    void *s = nullptr;
    BlockStream() { check(redio_stream_create(&s)); }
    ~BlockStream() { redio_stream_destroy(s); }
    BlockStream(const BlockStream &) = delete;
    BlockStream &operator=(const BlockStream &) = delete;
    operator void *() const { return s; }
};

// host Vec<T> -> device View<T> and back (the only PCIe crossings of a device-resident graph)
template <typename T>
void to_device(Receiver<std::vector<T>> u, Sender<View<T>> v)
{
    BlockStream st;
    for (;;) {
        auto x = u.recv();
        auto d = make<T>(x.size());
        check(redio_upload(d.data(), x.data(), x.size() * sizeof(T), st));
        check(redio_stream_sync(st));
        v.send_unwrap(std::move(d));
    }
}
template <typename T>
void to_host(Receiver<View<T>> u, Sender<std::vector<T>> v)
{
    BlockStream st;
    for (;;) {
        auto d = u.recv();
        std::vector<T> x(d.len);
        check(redio_download(x.data(), d.data(), d.len * sizeof(T), st));
        check(redio_stream_sync(st));
        v.send_unwrap(std::move(x));
    }
}

// kpn::shaper on a View stream: chunks of exactly l samples; zero-copy when possible, otherwise the
// straddling chunk is assembled once on the device.  A trailing partial chunk is dropped at hang-up.
template <typename T>
void shaper(Receiver<View<T>> u, Sender<View<T>> v, size_t l)
{
    BlockStream st;
    View<T> pend;           // partially filled chunk (owned copy)
    size_t have = 0;
    for (;;) {
        auto d = u.recv();
        size_t pos = 0;
        while (pos < d.len) {
            if (have == 0 && d.len - pos >= l) { // whole chunk inside this message: a view
                v.send_unwrap(d.sub(pos, l));
                pos += l;
                continue;
            }
            if (have == 0) pend = make<T>(l);
            const size_t take = std::min(l - have, d.len - pos);
            check(redio_copy(pend.data() + have, d.data() + pos, take * sizeof(T), st));
            have += take;
            pos += take;
            if (have == l) {
                check(redio_stream_sync(st));
                v.send_unwrap(pend);
                have = 0;
            }
        }
    }
}

// dsputils::convolve semantics on device streams (complex samples x real taps, optional decimation)
inline void fir(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps, size_t decim, bool fused)
{
    BlockStream st;
    redio_fir *h = nullptr;
    check(redio_fir_create(&h, taps.data(), taps.size(), decim, REDIO_FIR_COMPLEX | (fused ? REDIO_FIR_FUSED : 0)));
    struct G { redio_fir *h; ~G() { redio_fir_destroy(h); } } g{h};
    for (;;) {
        auto d = u.recv();
        auto o = make<std::complex<float>>(redio_fir_nout(h, d.len));
        check(redio_fir_enqueue(h, d.data(), d.len, o.data(), st));
        check(redio_stream_sync(st));
        v.send_unwrap(std::move(o));
    }
}

// kissfft::fft semantics: every message must be a whole number of block_size-sample blocks
inline void fft(Receiver<View<std::complex<float>>> pin, Sender<View<std::complex<float>>> cout, uint32_t block_size, uint32_t inv)
{
    BlockStream st;
    redio_fft *h = nullptr;
    check(redio_fft_create(&h, (int)block_size, (int)inv));
    struct G { redio_fft *h; ~G() { redio_fft_destroy(h); } } g{h};
    for (;;) {
        auto d = pin.recv();
        if (d.len % block_size) throw std::runtime_error("assert!(din.len() == block_size) (kissfft.rs:24)");
        auto o = make<std::complex<float>>(d.len);
        check(redio_fft_enqueue(h, d.data(), o.data(), d.len / block_size, st));
        check(redio_stream_sync(st));
        cout.send_unwrap(std::move(o));
    }
}

// the fused north-star chain as one block
inline void fir_fft_chain(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps,
                          size_t decim, int nfft, bool fused)
{
    BlockStream st;
    redio_chain *h = nullptr;
    check(redio_chain_create(&h, taps.data(), taps.size(), decim, nfft, fused ? REDIO_FIR_FUSED : 0));
    struct G { redio_chain *h; ~G() { redio_chain_destroy(h); } } g{h};
    for (;;) {
        auto d = u.recv();
        auto o = make<std::complex<float>>(redio_chain_nblocks(h, d.len) * (size_t)nfft);
        check(redio_chain_enqueue(h, d.data(), d.len, o.data(), st));
        check(redio_stream_sync(st));
        v.send_unwrap(std::move(o));
    }
}

// the receiver's messages straight into the chain: u8 I/Q bytes (rtlsdr::rtlSource, rtlsdr.rs:127-152) -> data_to_samples
// (rtlsdr.rs:159-162) -> FIR -> FFT as ONE kernel for the north-star shape (redio_chain_enqueue_u8)
inline void bytes_fir_fft_chain(Receiver<View<uint8_t>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps, size_t decim, int nfft,
                                bool fused)
{
    BlockStream st;
    redio_chain *h = nullptr;
    check(redio_chain_create(&h, taps.data(), taps.size(), decim, nfft, fused ? REDIO_FIR_FUSED : 0));
    struct G { redio_chain *h; ~G() { redio_chain_destroy(h); } } g{h};
    for (;;) {
        auto d = u.recv();
        auto o = make<std::complex<float>>(redio_chain_nblocks(h, d.len / 2) * (size_t)nfft);
        check(redio_chain_enqueue_u8(h, d.data(), d.len, o.data(), st));
        check(redio_stream_sync(st));
        v.send_unwrap(std::move(o));
    }
}

// the front end of the shipped graph: u8 IQ bytes -> |x| (rtlsdr.rs:159-162 + ratpak.rs:64-68)
inline void ingest_mag(Receiver<View<uint8_t>> u, Sender<View<float>> v)
{
    BlockStream st;
    for (;;) {
        auto d = u.recv();
        auto o = make<float>(d.len / 2);
        check(redio_ingest_u8_mag(d.data(), d.len, o.data(), st));
        check(redio_stream_sync(st));
        v.send_unwrap(std::move(o));
    }
}

// kpn::mul_vecs / sum_vecs (kpn.rs:198-203, 227-231) with the constant vector resident on the device
namespace detail {
inline int zip_call(bool add, const float *a, const float *b, float *o, size_t n, void *st) { return add ? redio_add_f32(a, b, o, n, st) : redio_mul_f32(a, b, o, n, st); }
inline int zip_call(bool add, const std::complex<float> *a, const std::complex<float> *b, std::complex<float> *o, size_t n, void *st)
{
    return add ? redio_add_c32(a, b, o, n, st) : redio_mul_c32(a, b, o, n, st);
}
template <typename T>
void zip_vecs(Receiver<View<T>> u, Sender<View<T>> v, const std::vector<T> &c, bool add)
{
    BlockStream st;
    auto dc = make<T>(c.size());
    check(redio_upload(dc.data(), c.data(), c.size() * sizeof(T), st));
    check(redio_stream_sync(st));
    for (;;) {
        auto x = u.recv();
        const size_t n = x.len < c.size() ? x.len : c.size();
        auto o = make<T>(n);
        check(zip_call(add, x.data(), dc.data(), o.data(), n, st));
        check(redio_stream_sync(st));
        v.send_unwrap(std::move(o));
    }
}
} // namespace detail
template <typename T>
void mul_vecs(Receiver<View<T>> u, Sender<View<T>> v, std::vector<T> c) { detail::zip_vecs<T>(std::move(u), std::move(v), c, false); }
template <typename T>
void sum_vecs(Receiver<View<T>> u, Sender<View<T>> v, std::vector<T> c) { detail::zip_vecs<T>(std::move(u), std::move(v), c, true); }

// samplerate::resample semantics (samplerate.rs:59-87) on a device stream: one converter state for the
// life of the block, output capacity floor(ratio*len + 1) per message, output_frames_gen samples sent
inline void resample(Receiver<View<float>> din, Sender<View<float>> dout, double ratio)
{
    BlockStream st;
    redio_src *h = nullptr;
    int rc = redio_src_create(&h, 1 /* SRC_SINC_MEDIUM_QUALITY, samplerate.rs:27 */, 1);
    if (rc != REDIO_OK) throw std::runtime_error(redio_strerror(rc));
    struct G { redio_src *h; ~G() { redio_src_destroy(h); } } g{h};
    for (;;) {
        auto d = din.recv();
        const long lout = (long)(ratio * (double)d.len + 1.0);
        auto o = make<float>((size_t)lout);
        long used = 0, gen = 0;
        rc = redio_src_process(h, d.data(), (long)d.len, (long)d.len, o.data(), lout, lout, ratio, 0, &used, &gen, st);
        if (rc != REDIO_OK) throw std::runtime_error(redio_strerror(rc)); // the reference panics with src_strerror's text
        dout.send_unwrap(o.sub(0, (size_t)gen));
    }
}

// the 64-channel polyphase filterbank (BASELINE configs[3]) as a block: rows of 64 channel samples out
inline void channelizer(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> proto, int nchan,
                        int taps_per_branch, bool fused)
{
    BlockStream st;
    redio_pfb *h = nullptr;
    check(redio_pfb_create(&h, proto.data(), nchan, taps_per_branch, fused ? REDIO_FIR_FUSED : 0));
    struct G { redio_pfb *h; ~G() { redio_pfb_destroy(h); } } g{h};
    for (;;) {
        auto d = u.recv();
        auto o = make<std::complex<float>>(redio_pfb_nrows(h, d.len) * (size_t)nchan);
        check(redio_pfb_enqueue(h, d.data(), d.len, o.data(), 1, st));
        check(redio_stream_sync(st));
        v.send_unwrap(std::move(o));
    }
}

// overlap-save FFT convolution (BASELINE configs[4]) with dsputils::convolve's valid-mode semantics per message
inline void overlap_save(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps, int nfft)
{
    BlockStream st;
    redio_ovsave *h = nullptr;
    check(redio_ovsave_create(&h, taps.data(), taps.size(), nfft));
    struct G { redio_ovsave *h; ~G() { redio_ovsave_destroy(h); } } g{h};
    for (;;) {
        auto d = u.recv();
        auto o = make<std::complex<float>>(redio_ovsave_nout(h, d.len));
        check(redio_ovsave_enqueue(h, d.data(), d.len, o.data(), st));
        check(redio_stream_sync(st));
        v.send_unwrap(std::move(o));
    }
}

// ---- the windowed blocks as STREAMS: history carried across messages on the device (redio_*_stream_*, include/redio.h) ----
// The blocks above keep the reference's per-message semantics (dsputils::convolve is stateless, dsputils.rs:30-32: a
// message seam costs ntaps - 1 outputs).  These carry the stream's unconsumed tail in HBM, so the concatenation of their
// output messages does not depend on how the input was cut into messages; a message that completes no
// output unit sends nothing.
inline void fir_stream(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps, size_t decim, bool fused)
{
    BlockStream st;
    redio_fir *h = nullptr;
    check(redio_fir_create(&h, taps.data(), taps.size(), decim, REDIO_FIR_COMPLEX | (fused ? REDIO_FIR_FUSED : 0)));
    redio_fir_stream *s = nullptr;
    const int rc = redio_fir_stream_create(&s, h);
    struct G { redio_fir *h; redio_fir_stream *s; ~G() { redio_fir_stream_destroy(s); redio_fir_destroy(h); } } g{h, s};
    check(rc);
    for (;;) {
        auto d = u.recv();
        const size_t n = redio_fir_stream_nout(s, d.len);
        auto o = make<std::complex<float>>(n);
        size_t got = 0;
        check(redio_fir_stream_enqueue(s, d.data(), d.len, o.data(), &got, st));
        check(redio_stream_sync(st)); // the input view may be released once the seam has been staged
        if (n) v.send_unwrap(std::move(o));
    }
}

inline void fir_fft_chain_stream(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps,
                                 size_t decim, int nfft, bool fused)
{
    BlockStream st;
    redio_chain *h = nullptr;
    check(redio_chain_create(&h, taps.data(), taps.size(), decim, nfft, fused ? REDIO_FIR_FUSED : 0));
    redio_chain_stream *s = nullptr;
    const int rc = redio_chain_stream_create(&s, h);
    struct G { redio_chain *h; redio_chain_stream *s; ~G() { redio_chain_stream_destroy(s); redio_chain_destroy(h); } } g{h, s};
    check(rc);
    for (;;) {
        auto d = u.recv();
        const size_t n = redio_chain_stream_nout(s, d.len);
        auto o = make<std::complex<float>>(n);
        size_t got = 0;
        check(redio_chain_stream_enqueue(s, d.data(), d.len, o.data(), &got, st));
        check(redio_stream_sync(st));
        if (n) v.send_unwrap(std::move(o));
    }
}

inline void channelizer_stream(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> proto, int nchan,
                               int taps_per_branch, bool fused)
{
    BlockStream st;
    redio_pfb *h = nullptr;
    check(redio_pfb_create(&h, proto.data(), nchan, taps_per_branch, fused ? REDIO_FIR_FUSED : 0));
    redio_pfb_stream *s = nullptr;
    const int rc = redio_pfb_stream_create(&s, h);
    struct G { redio_pfb *h; redio_pfb_stream *s; ~G() { redio_pfb_stream_destroy(s); redio_pfb_destroy(h); } } g{h, s};
    check(rc);
    for (;;) {
        auto d = u.recv();
        const size_t n = redio_pfb_stream_nout(s, d.len);
        auto o = make<std::complex<float>>(n);
        size_t got = 0;
        check(redio_pfb_stream_enqueue(s, d.data(), d.len, o.data(), &got, st));
        check(redio_stream_sync(st));
        if (n) v.send_unwrap(std::move(o));
    }
}

inline void overlap_save_stream(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps, int nfft)
{
    BlockStream st;
    redio_ovsave *h = nullptr;
    check(redio_ovsave_create(&h, taps.data(), taps.size(), nfft));
    redio_ovsave_stream *s = nullptr;
    const int rc = redio_ovsave_stream_create(&s, h);
    struct G { redio_ovsave *h; redio_ovsave_stream *s; ~G() { redio_ovsave_stream_destroy(s); redio_ovsave_destroy(h); } } g{h, s};
    check(rc);
    for (;;) {
        auto d = u.recv();
        const size_t n = redio_ovsave_stream_nout(s, d.len);
        auto o = make<std::complex<float>>(n);
        size_t got = 0;
        check(redio_ovsave_stream_enqueue(s, d.data(), d.len, o.data(), &got, st));
        check(redio_stream_sync(st));
        if (n) v.send_unwrap(std::move(o));
    }
}

} // namespace dev
} // namespace kpn