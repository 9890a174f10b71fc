// kpn_baseline.cpp -- the CPU baseline in the REFERENCE'S STRUCTURE (SURVEY.md 8d "CPU baseline, same run").
// TEST / MEASUREMENT INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg); nothing under libredio_amd/ links it.
//
// LibRedio runs one OS thread per block and hands one heap-allocated Vec per message from block to block through
// an unbounded mpsc queue (src/ratpak.rs:60-185; src/kissfft/src/kissfft.rs:18-31: recv a Vec of block_size
// samples, allocate the output Vec, kiss_fft, send).  This file times BASELINE.json configs[1] built that way
// from the oracle's C functions:
//     source --Vec(5120+126 cf32)--> [fir: convolve + keep every 5th] --Vec(1024)--> [fft: kiss_fft] --Vec(1024)--> [sink]
// three block threads plus the source, each message its own malloc/free, mutex + condvar queues.  The source
// frames the stream with the 126-sample overlap a host needs so that consecutive messages give consecutive
// decimated blocks (per-message convolve is stateless, dsputils.rs:30-32).  Scalar code, gcc -O2, no FMA.
#include "redio_oracle.h"
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace {
template <typename T>
struct Queue { // std::sync::mpsc::channel(): unbounded FIFO, non-blocking send, blocking recv
    std::mutex m;
    std::condition_variable cv;
    std::deque<T> q;
    bool closed = false;
    void send(T v)
    {
        { std::lock_guard<std::mutex> l(m); q.push_back(std::move(v)); }
        cv.notify_one();
    }
    void close()
    {
        { std::lock_guard<std::mutex> l(m); closed = true; }
        cv.notify_all();
    }
    bool recv(T &v)
    {
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return false;
        v = std::move(q.front());
        q.pop_front();
        return true;
    }
    size_t depth() { std::lock_guard<std::mutex> l(m); return q.size(); }
};
using Msg = std::vector<orc_cpx>;
} // namespace

// Runs the pipeline for about `seconds` of wall time over a 2^log2n-sample hash-generated stream (re-read in a
// loop); returns the wall time and writes the number of input samples whose spectra reached the sink.
extern "C" double orc_kpn_chain_baseline(double seconds, int log2n, uint32_t seed, const float *taps, size_t ntaps, size_t decim,
                                         int nfft, uint64_t *samples_done, uint64_t *messages_done)
{
    const size_t n = (size_t)1 << log2n;
    const size_t hop = (size_t)nfft * decim, win = hop + ntaps - decim; // inputs per message (valid mode -> nfft outputs)
    std::vector<orc_cpx> stream(n);
    orc_synth_iq(seed, 0, n, stream.data());
    const size_t per_pass = (n - win) / hop + 1;
    Queue<Msg> q0, q1, q2;
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> done{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::thread source([&] {
        for (size_t m = 0; !stop.load(std::memory_order_relaxed); m = (m + 1) % per_pass) {
            while (q0.depth() > 64 && !stop.load(std::memory_order_relaxed)) std::this_thread::yield(); // a file / SDR source is paced; do not let the queue eat the host
            Msg v(stream.begin() + m * hop, stream.begin() + m * hop + win); // one heap Vec per message
            q0.send(std::move(v));
        }
        q0.close();
    });
    std::thread fir([&] { // cross_applicator_vecs(convolve) + decimation
        Msg in;
        while (q0.recv(in)) {
            Msg out((size_t)nfft);
            orc_fir_c32(in.data(), in.size(), taps, ntaps, decim, 0, out.data());
            q1.send(std::move(out));
        }
        q1.close();
    });
    std::thread fft([&] { // kissfft::fft, kissfft.rs:18-31: one cfg for the life of the block
        orc_kiss_state *cfg = orc_kiss_fft_alloc(nfft, 0);
        Msg in;
        while (q1.recv(in)) {
            Msg out((size_t)nfft);
            orc_kiss_fft(cfg, in.data(), out.data());
            q2.send(std::move(out));
        }
        orc_kiss_fft_free(cfg);
        q2.close();
    });
    std::thread sink([&] {
        Msg in;
        volatile float keep = 0.f;
        while (q2.recv(in)) { keep = keep + in[0].r; done.fetch_add(1, std::memory_order_relaxed); }
    });
    std::this_thread::sleep_for(std::chrono::duration<double>(seconds));
    const uint64_t msgs = done.load();
    const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    stop.store(true);
    source.join(); fir.join(); fft.join(); sink.join();
    if (samples_done) *samples_done = msgs * hop;
    if (messages_done) *messages_done = msgs;
    return wall;
}
