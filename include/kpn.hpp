// kpn.hpp -- C++ twin of LibRedio's dataflow surface (src/kpn/src/kpn.rs) for hosts without Rust.
//
// Same convention as the reference: a block is a free function `name(inputs..., outputs..., params...)`
// run on its own thread, inputs are Receiver<T>, outputs are Sender<U>, channels are unbounded FIFOs
// with non-blocking send and blocking receive (README.mkd:3, std::sync::mpsc).  Names, argument order
// and per-block quirks follow kpn.rs line by line (cited at each block).  The one deliberate
// difference is failure handling: where the Rust does `recv().unwrap()` and panics
// when the upstream hangs up, `Receiver::recv()` throws kpn::hangup, which `kpn::spawn` catches -- the
// block ends, its endpoints drop, and the hang-up propagates downstream exactly like the panic
// cascade, but without aborting the process.
//
// The three hot blocks call the MI355X library through its C ABI:
//   dsputils::convolve  -> redio_convolve_f32      (include/redio.h)
//   kissfft::fft        -> kiss_fft_alloc/kiss_fft  (include/kiss_fft.h)
//   samplerate::resample-> src_new/src_process      (include/samplerate.h)
#pragma once
#include <complex>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <iostream>
#include <memory>
#include <mutex>
#include <optional>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "kiss_fft.h"
#include "redio.h"
#include "samplerate.h"

namespace kpn {

struct hangup : std::runtime_error {
    hangup() : std::runtime_error("channel hung up") {}
};

namespace detail {
template <typename T>
struct Chan {
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    int senders = 0;
    bool receiver_alive = true;
    size_t capacity = 0; // 0: unbounded (std::sync::mpsc::channel); > 0: send blocks while the queue is full
};
} // namespace detail

template <typename T>
class Sender {
    std::shared_ptr<detail::Chan<T>> c_;
    void attach()
    {
        if (c_) { std::lock_guard<std::mutex> l(c_->m); ++c_->senders; }
    }
    void detach()
    {
        if (!c_) return;
        {
            std::lock_guard<std::mutex> l(c_->m);
            --c_->senders;
        }
        c_->cv.notify_all();
        c_.reset();
    }

public:
    Sender() = default;
    explicit Sender(std::shared_ptr<detail::Chan<T>> c) : c_(std::move(c)) { attach(); }
    Sender(const Sender &o) : c_(o.c_) { attach(); } // Sender::clone()
    Sender(Sender &&o) noexcept : c_(std::move(o.c_)) {}
    Sender &operator=(Sender o) { std::swap(c_, o.c_); return *this; }
    ~Sender() { detach(); }
    // mpsc send: never blocks on an unbounded channel; false when the receiver is gone (Err in Rust; blocks
    // decide to unwrap).  On a bounded channel (bounded_channel) it waits for a free slot: back-pressure.
    bool send(T v) const
    {
        {
            std::unique_lock<std::mutex> l(c_->m);
            if (c_->capacity) c_->cv.wait(l, [&] { return !c_->receiver_alive || c_->q.size() < c_->capacity; });
            if (!c_->receiver_alive) return false;
            c_->q.push_back(std::move(v));
        }
        c_->cv.notify_all();
        return true;
    }
    void send_unwrap(T v) const
    {
        if (!send(std::move(v))) throw hangup();
    }
};

template <typename T>
class Receiver {
    std::shared_ptr<detail::Chan<T>> c_;

public:
    Receiver() = default;
    explicit Receiver(std::shared_ptr<detail::Chan<T>> c) : c_(std::move(c)) {}
    Receiver(const Receiver &) = delete;
    Receiver(Receiver &&o) noexcept : c_(std::move(o.c_)) {}
    Receiver &operator=(Receiver &&o) noexcept
    {
        close();
        c_ = std::move(o.c_);
        return *this;
    }
    ~Receiver() { close(); }
    void close()
    {
        if (!c_) return;
        {
            std::lock_guard<std::mutex> l(c_->m);
            c_->receiver_alive = false;
            c_->q.clear();
        }
        c_->cv.notify_all(); // senders blocked on a full bounded channel see the hang-up
        c_.reset();
    }
    // blocking receive; nullopt once every sender is gone and the queue is drained (Err(RecvError))
    std::optional<T> try_recv_blocking() const
    {
        std::unique_lock<std::mutex> l(c_->m);
        c_->cv.wait(l, [&] { return !c_->q.empty() || c_->senders == 0; });
        if (c_->q.empty()) return std::nullopt;
        T v = std::move(c_->q.front());
        c_->q.pop_front();
        if (c_->capacity) { l.unlock(); c_->cv.notify_all(); } // a slot is free
        return v;
    }
    T recv() const // recv().unwrap()
    {
        auto v = try_recv_blocking();
        if (!v) throw hangup();
        return std::move(*v);
    }
    // try_recv(): nullopt when empty (the reference's grapes() unwraps this and panics, kpn.rs:245)
    std::optional<T> try_recv() const
    {
        std::unique_lock<std::mutex> l(c_->m);
        if (c_->q.empty()) return std::nullopt;
        T v = std::move(c_->q.front());
        c_->q.pop_front();
        if (c_->capacity) { l.unlock(); c_->cv.notify_all(); }
        return v;
    }
};

template <typename T>
std::pair<Sender<T>, Receiver<T>> channel()
{
    auto c = std::make_shared<detail::Chan<T>>();
    return {Sender<T>(c), Receiver<T>(c)};
}

// A deliberate deviation for device-resident graphs: the reference's channels are unbounded, which
// is harmless for small host Vecs but lets a fast producer pin an unbounded amount of HBM.  A bounded channel holds
// at most `capacity` messages; send waits for a slot (credit), everything else behaves like channel().
template <typename T>
std::pair<Sender<T>, Receiver<T>> bounded_channel(size_t capacity)
{
    auto c = std::make_shared<detail::Chan<T>>();
    c->capacity = capacity ? capacity : 1;
    return {Sender<T>(c), Receiver<T>(c)};
}

// one named thread per block (src/ratpak.rs:60-185); a hang-up ends the block quietly.  Any other
// exception a block throws (an assert of the reference: eat/binconv overrun, a wrong fft message size,
// a converter error, a HIP failure under a kpn::dev block ...) is the reference's panic of THAT task only:
// the text goes to stderr, the block's endpoints drop with the closure and the hang-up cascades downstream.
template <typename F>
std::thread spawn(F &&f)
{
    return std::thread([fn = std::forward<F>(f)]() mutable {
        try {
            fn();
        } catch (const hangup &) {
        } catch (const std::exception &e) {
            std::fprintf(stderr, "kpn: block panicked: %s\n", e.what());
        } catch (...) {
            std::fprintf(stderr, "kpn: block panicked (unknown exception)\n");
        }
    });
}

// ---------------------------------------------------------------- kpn.rs blocks, in file order

// run length encoding, kpn.rs:17-29 (the last run is never flushed)
template <typename T>
void rle(Receiver<T> u, Sender<std::pair<T, size_t>> v)
{
    T x = u.recv();
    size_t i = 1;
    for (;;) {
        T y = u.recv();
        if (y != x) {
            v.send_unwrap({x, i});
            i = 1;
        } else {
            i = i + 1;
        }
        x = y;
    }
}

// counts -> seconds, kpn.rs:32-38
template <typename T>
void dle(Receiver<std::pair<T, size_t>> u, Sender<std::pair<T, float>> v, size_t s_rate)
{
    for (;;) {
        auto p = u.recv();
        v.send_unwrap({p.first, (float)p.second / (float)s_rate});
    }
}

// duration length decoding, kpn.rs:41-47
template <typename T>
void dld(Receiver<std::pair<T, float>> u, Sender<T> v, float s_rate)
{
    for (;;) {
        auto p = u.recv();
        const size_t n = (size_t)(p.second * s_rate);
        for (size_t k = 0; k < n; ++k) v.send_unwrap(p.first);
    }
}

// run length decoding, kpn.rs:50-56
template <typename T>
void rld(Receiver<std::pair<T, size_t>> u, Sender<T> v)
{
    for (;;) {
        auto p = u.recv();
        for (size_t k = 0; k < p.second; ++k) v.send_unwrap(p.first);
    }
}

// drop repeated samples, kpn.rs:73-82 (the first value is never emitted)
template <typename T>
void differentiator(Receiver<T> u, Sender<T> v)
{
    T x = u.recv();
    for (;;) {
        T y = u.recv();
        if (x != y) {
            x = y;
            v.send_unwrap(x);
        }
    }
}

// first-order difference, kpn.rs:85-92 (keeps the DIFFERENCE as the next x, as written)
template <typename T>
void dxdt(Receiver<T> u, Sender<T> v)
{
    T x = u.recv();
    for (;;) {
        T y = u.recv();
        x = y - x;
        v.send_unwrap(x);
    }
}

// unpack vecs to a list of elements, kpn.rs:95-101
template <typename T>
void unpacketizer(Receiver<std::vector<T>> u, Sender<T> v)
{
    for (;;)
        for (auto &x : u.recv()) v.send_unwrap(x);
}

// kpn.rs:104-108
template <typename T>
void print_sink(Receiver<T> u)
{
    for (;;) std::cout << u.recv() << std::endl;
}

// MSB-first binary digits -> unsigned, kpn.rs:111-113
inline size_t b2d(const std::vector<size_t> &xs)
{
    size_t acc = 0;
    for (size_t i = 0; i < xs.size(); ++i) acc += ((size_t)1 << (xs.size() - i - 1)) * xs[i];
    return acc;
}

// split by a list of widths, kpn.rs:116-124 (an overrun is the reference's slice panic)
inline std::vector<size_t> eat(const std::vector<size_t> &x, const std::vector<size_t> &is)
{
    size_t i = 0;
    std::vector<size_t> out;
    for (size_t index : is) {
        if (i + index > x.size()) throw std::out_of_range("eat: widths overrun the input");
        out.push_back(b2d(std::vector<size_t>(x.begin() + (long)i, x.begin() + (long)(i + index))));
        i = i + index;
    }
    return out;
}

// map |T|->T across Channel<T>, kpn.rs:127-131
template <typename T, typename F>
void applicator(Receiver<T> u, Sender<T> v, F f)
{
    for (;;) v.send_unwrap(f(u.recv()));
}

// map |&T|->T across Channel<Vec<T>>, kpn.rs:134-138
template <typename T, typename F>
void applicator_vecs(Receiver<std::vector<T>> u, Sender<std::vector<T>> v, F f)
{
    for (;;) {
        auto in = u.recv();
        std::vector<T> out;
        out.reserve(in.size());
        for (auto &x : in) out.push_back(f(x));
        v.send_unwrap(std::move(out));
    }
}

// run f(Sender), then keep the sender alive forever, kpn.rs:141-145 (here: until `stop` is closed)
template <typename T, typename F>
void soft_source(Sender<T> v, F f)
{
    f(v);
    auto park = channel<int>();
    park.second.try_recv_blocking(); // the reference parks forever; a dropped sender lets this return
}

// hand the whole stream to a closure, kpn.rs:148-150
template <typename T, typename U, typename F>
void looper(Receiver<T> u, Sender<U> v, F f)
{
    f(u, v);
}

// take maybe-T to T, kpn.rs:153-160
template <typename T>
void looper_optional(Receiver<std::optional<T>> u, Sender<T> v)
{
    for (;;) {
        auto d = u.recv();
        if (d) v.send_unwrap(*d);
    }
}

// map |T|->U, kpn.rs:163-167
template <typename T, typename U, typename F>
void cross_applicator(Receiver<T> u, Sender<U> v, F f)
{
    for (;;) v.send_unwrap(f(u.recv()));
}

// map |&T|->U across Vec<T>, kpn.rs:170-174
template <typename T, typename U, typename F>
void cross_applicator_vecs(Receiver<std::vector<T>> u, Sender<std::vector<U>> v, F f)
{
    for (;;) {
        auto in = u.recv();
        std::vector<U> out;
        out.reserve(in.size());
        for (auto &x : in) out.push_back(f(x));
        v.send_unwrap(std::move(out));
    }
}

// kpn.rs:177-179
template <typename T>
std::vector<T> vec(const T *u, size_t n) { return std::vector<T>(u, u + n); }

// duplicate a stream, kpn.rs:182-189
template <typename T>
void fork(Receiver<T> u, std::vector<Sender<T>> v)
{
    for (;;) {
        T x = u.recv();
        for (auto &y : v) y.send_unwrap(x);
    }
}

// scale by a constant, kpn.rs:192-196
template <typename T>
void mul(Receiver<T> u, Sender<T> v, T c)
{
    for (;;) v.send_unwrap(u.recv() * c);
}

// scale vectors by a vector of constants, kpn.rs:199-203 (zip truncates to the shorter)
template <typename T>
void mul_vecs(Receiver<std::vector<T>> u, Sender<std::vector<T>> v, std::vector<T> c)
{
    for (;;) {
        auto x = u.recv();
        const size_t n = x.size() < c.size() ? x.size() : c.size();
        std::vector<T> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = x[i] * c[i];
        v.send_unwrap(std::move(out));
    }
}

// lock-step N-input sum seeded with c, kpn.rs:206-210
template <typename T>
void sum_across(std::vector<Receiver<T>> u, Sender<T> v, T c)
{
    for (;;) {
        T b = c;
        for (auto &y : u) b = b + y.recv();
        v.send_unwrap(b);
    }
}

// kpn.rs:213-217
template <typename T>
void mul_across(std::vector<Receiver<T>> u, Sender<T> v, T c)
{
    for (;;) {
        T b = c;
        for (auto &y : u) b = b * y.recv();
        v.send_unwrap(b);
    }
}

// kpn.rs:220-224
template <typename T>
void sum_across_vecs(std::vector<Receiver<std::vector<T>>> u, Sender<std::vector<T>> v, std::vector<T> c)
{
    for (;;) {
        std::vector<T> b = c;
        for (auto &y : u) {
            auto a = y.recv();
            const size_t n = a.size() < b.size() ? a.size() : b.size();
            std::vector<T> nb(n);
            for (size_t i = 0; i < n; ++i) nb[i] = a[i] + b[i];
            b = std::move(nb);
        }
        v.send_unwrap(std::move(b));
    }
}

// offset vectors by a vector of constants, kpn.rs:227-231
template <typename T>
void sum_vecs(Receiver<std::vector<T>> u, Sender<std::vector<T>> v, std::vector<T> c)
{
    for (;;) {
        auto x = u.recv();
        const size_t n = x.size() < c.size() ? x.size() : c.size();
        std::vector<T> out(n);
        for (size_t i = 0; i < n; ++i) out[i] = x[i] + c[i];
        v.send_unwrap(std::move(out));
    }
}

// "accumulator" that is an offset, kpn.rs:234-238
template <typename T>
void sum(Receiver<T> u, Sender<T> v, T c)
{
    for (;;) v.send_unwrap(u.recv() + c);
}

// polling merge, kpn.rs:241-251: try_recv().unwrap() -- an empty input is the reference's panic
template <typename T>
void grapes(std::vector<Receiver<T>> u, Sender<T> v)
{
    for (;;)
        for (auto &x : u) {
            auto d = x.try_recv();
            if (!d) throw std::runtime_error("grapes: try_recv on an empty channel (kpn.rs:245 panics)");
            v.send_unwrap(std::move(*d));
            std::this_thread::sleep_for(std::chrono::nanoseconds(10));
        }
}

// emit c then pass through, kpn.rs:254-259 (send errors after the first are ignored)
template <typename T>
void delay(Receiver<T> u, Sender<T> v, T c)
{
    v.send_unwrap(c);
    for (;;) v.send(u.recv());
}
template <typename T>
void delay_vecs(Receiver<T> u, Sender<T> v, T c) { delay(std::move(u), std::move(v), std::move(c)); } // kpn.rs:261-263

// collect Somes; on None emit iff exactly l were collected, kpn.rs:266-275
template <typename T>
void shaper_optional(Receiver<std::optional<T>> u, Sender<std::vector<T>> v, size_t l)
{
    std::vector<T> x;
    for (;;) {
        auto y = u.recv();
        if (y) {
            x.push_back(*y);
        } else if (x.size() == l) {
            v.send(x);
            x.clear();
        } else {
            x.clear();
        }
    }
}

// T -> Vec<T> of length l, kpn.rs:278-282 (a hang-up mid-block drops the partial block)
template <typename T>
void shaper(Receiver<T> u, Sender<std::vector<T>> v, size_t l)
{
    for (;;) {
        std::vector<T> blk;
        blk.reserve(l);
        for (size_t i = 0; i < l; ++i) blk.push_back(u.recv());
        v.send_unwrap(std::move(blk));
    }
}

// Vec<T> -> T, ends cleanly on hang-up (u.iter()), kpn.rs:285-291
template <typename T>
void shaper_vecs(Receiver<std::vector<T>> u, Sender<T> v)
{
    while (auto x = u.try_recv_blocking())
        for (auto &y : *x) v.send_unwrap(y);
}

// eat per message, kpn.rs:295-299
inline void binconv(Receiver<std::vector<size_t>> u, Sender<std::vector<size_t>> v, std::vector<size_t> l)
{
    for (;;) v.send_unwrap(eat(u.recv(), l));
}

} // namespace kpn

// ---------------------------------------------------------------- the hot blocks (other crates)

namespace dsputils {
// dsputils::convolve, src/dsputils/src/dsputils.rs:30-32 -- runs on the MI355X
inline std::vector<float> convolve(const std::vector<float> &u, const std::vector<float> &v)
{
    if (v.empty()) throw std::invalid_argument("convolve: windows(0) panics in the reference");
    std::vector<float> out(u.size() >= v.size() ? u.size() - v.size() + 1 : 0);
    size_t n = 0;
    float dummy = 0.f;
    int rc = redio_convolve_f32(u.data(), u.size(), v.data(), v.size(), out.empty() ? &dummy : out.data(), &n);
    if (rc != REDIO_OK) throw std::runtime_error(redio_strerror(rc));
    out.resize(n);
    return out;
}
inline std::vector<float> lpf(size_t m, float fc) // dsputils.rs:66-71 (as written: tap 1 is NaN)
{
    std::vector<float> t(m);
    if (redio_lpf(m, fc, t.data()) != REDIO_OK) throw std::invalid_argument("lpf: fc < 0.5 asserted");
    return t;
}
inline std::vector<float> lpf_corrected(size_t m, float fc)
{
    std::vector<float> t(m);
    if (redio_lpf_corrected(m, fc, t.data()) != REDIO_OK) throw std::invalid_argument("lpf_corrected");
    return t;
}
} // namespace dsputils

namespace kissfft {
// kissfft::fft(pin, cout, block_size, inv), src/kissfft/src/kissfft.rs:18-31
inline void fft(kpn::Receiver<std::vector<std::complex<float>>> pin, kpn::Sender<std::vector<std::complex<float>>> cout,
                uint32_t block_size, uint32_t inv)
{
    kiss_fft_cfg cfg = kiss_fft_alloc((int)block_size, (int)inv, nullptr, nullptr); // :19
    if (!cfg) throw std::runtime_error("kiss_fft_alloc failed (no HIP device?)");
    struct Guard { kiss_fft_cfg c; ~Guard() { kiss_fft_free(c); kiss_fft_cleanup(); } } g{cfg};
    for (;;) {
        std::vector<std::complex<float>> fout(block_size); // :21-22
        auto din = pin.recv();                             // :23
        if (din.size() != block_size) throw std::runtime_error("assert!(din.len() == block_size) (kissfft.rs:24)");
        kiss_fft(cfg, reinterpret_cast<const kiss_fft_cpx *>(din.data()), reinterpret_cast<kiss_fft_cpx *>(fout.data())); // :26
        cout.send_unwrap(std::move(fout));                 // :27
    }
}
} // namespace kissfft

namespace samplerate {
// samplerate::resample(din, dout, ratio), src/samplerate/src/samplerate.rs:59-87
inline void resample(kpn::Receiver<std::vector<float>> din, kpn::Sender<std::vector<float>> dout, double ratio)
{
    int error = 0;
    SRC_STATE *ctx = src_new(1, 1, &error); // :61
    if (!ctx) throw std::runtime_error(src_strerror(error) ? src_strerror(error) : "src_new failed");
    struct Guard { SRC_STATE *s; ~Guard() { src_delete(s); } } g{ctx};
    for (;;) {
        auto vin = din.recv();                                              // :63
        const size_t lout = (size_t)((ratio * (double)vin.size()) + 1.0);   // :64
        std::vector<float> vout(lout);                                      // :65
        SRC_DATA d;
        d.data_in = vin.data(); d.data_out = vout.data();
        d.input_frames = (long)vin.size(); d.output_frames = (long)lout;
        d.input_frames_used = 0; d.output_frames_gen = 0; d.end_of_input = 0; d.src_ratio = ratio; // :66-75
        const int err = src_process(ctx, &d);                               // :76
        if (err != 0) throw std::runtime_error(src_strerror(err) ? src_strerror(err) : "src_process failed"); // :77-83
        vout.resize((size_t)d.output_frames_gen);                           // :84
        dout.send_unwrap(std::move(vout));                                  // :85
    }
}
} // namespace samplerate
