// kpn_dev.hpp -- device-resident messages for kpn graphs: the steps either side of
// every hot block without host round trips.  A message is a kpn::dev::View = shared ownership of a
// device buffer + (offset, length) in samples, so the reshaping blocks of src/kpn/src/kpn.rs become
// metadata operations:
//   fork (kpn.rs:182-189)            clones the handle, not the samples (kpn::fork works unchanged on Views)
//   shaper (kpn.rs:278-282)          dev::shaper: re-chunks a stream of Views into Views of length l,
//                                    zero-copy when a chunk lies inside one buffer
//   unpacketizer / shaper_vecs       the inverse is the identity on a View stream
// and the hot blocks consume and produce Views through the device plans of include/redio.h.
//
// How a message travels (round 6; before, every block did hipMalloc + hipStreamSynchronize + hipFree per message and the graph ran at
// 5 % of the bare launch rate at 2^24-sample messages, profiles/r06_kpn_graph.txt):
//   * ORDER is the GPU queue's, not the host's.  The compute blocks of a graph enqueue on ONE HIP stream per device (graph_stream()):
//     a consumer receives a message only after its producer has enqueued the work that fills it, so its own work lands behind it in the
//     same queue -- no event, no wait, no packet beyond the kernels themselves.  The kernels here fill the chip and are bound by HBM, so
//     two of them side by side would only share the bandwidth; the host threads still overlap their launch work.  Blocks that move data
//     over PCIe or synchronise with the host (to_device, to_host, resample) own a private stream so that a copy never sits in front of
//     a kernel; set_stream_policy(PER_BLOCK) gives every block its own (graphs whose branches are many small kernels).
//     Between DIFFERENT streams the order is made on demand, by the side that needs it: a buffer remembers the stream that wrote it
//     and the streams that read it; a consumer on another stream records an event on the WRITER's stream at that moment (everything
//     enqueued there so far, a superset of the message's work) and makes its own stream wait for it (redio_stream_wait_event: a
//     queue-side dependency, the host thread does not block); a recycled buffer's new writer does the same with its previous readers.
//     Host synchronisation is left where data really goes to the CPU.  set_host_sync(true) restores a host synchronisation before
//     every send (debugging: a fault then belongs to the block that reports it).
//   * MEMORY is a bounded ring per block (dev::Ring): at most `depth` output buffers, recycled when the last handle to a message
//     drops; a producer whose `depth` buffers are all still held downstream WAITS in acquire() -- that is the credit scheme SURVEY.md 8b
//     asks for (the reference's unbounded mpsc has no back-pressure: a documented deviation).  After warm-up a graph performs no device
//     allocation and no hipFree (redio_malloc_count() stays put; tests/test_kpn_cpp.py).  A sink that hoards more than `depth`
//     messages of one producer without dropping them stalls that producer: drop handles, or give that block a deeper ring.
//     A ring also has a BYTE budget (default_ring_bytes_ref below): with the compute blocks on one stream, a producer that runs k
//     messages ahead puts k outputs between a message and the kernel that reads it, and rotates k + 1 buffers -- past the last-level
//     cache both cost time (measured: 6 us each on a 128 MiB message that takes 69 us with neither).
//     The rings bound MEMORY, not how far the host threads run ahead of the GPU: a buffer is recycled when its handles drop (the queue
//     orders the reuse), so hundreds of messages' kernels may sit in the HIP queue; a block that needs a result on the CPU synchronises.
//   * Blocks written elsewhere use the same three calls: `auto o = ring.acquire<T>(n, st)`, `{ Reading<T> in(view, st); enqueue...; }`,
//     `publish(o, st)` before sending o, with `BlockStream st;`.
#pragma once
#include "kpn.hpp"
#include <algorithm>
#include <atomic>
#include <chrono>
#include <complex>
#include <map>
#include <memory>
#include <vector>

namespace kpn {
namespace dev {

inline void check(int rc)
{
    if (rc != REDIO_OK) throw std::runtime_error(redio_strerror(rc));
}

// The device calls the rings and the ordering need.  A table so that their logic (credits, recycling, who waits for whom, hand-over
// between threads) also runs where there is no GPU: tests/cpp/kpn_tests.cpp `plumbing` swaps in host memory and counting stand-ins
// and runs it under ThreadSanitizer / AddressSanitizer.  Product code never touches it: the default is libredio's C ABI.
struct DeviceApi {
    int (*malloc_)(void **, size_t) = redio_malloc;
    int (*free_)(void *) = redio_free;
    int (*event_create)(void **) = redio_event_create_sync;
    int (*event_destroy)(void *) = redio_event_destroy;
    int (*event_record)(void *, void *) = redio_event_record;
    int (*stream_wait_event)(void *, void *) = redio_stream_wait_event;
    int (*stream_create)(void **) = redio_stream_create;
    int (*stream_destroy)(void *) = redio_stream_destroy;
    int (*stream_sync)(void *) = redio_stream_sync;
    int (*get_device)(int *) = redio_get_device;
};
inline DeviceApi &api()
{
    static DeviceApi a;
    return a;
}

inline std::atomic<bool> &host_sync_flag()
{
    static std::atomic<bool> f{false};
    return f;
}
inline void set_host_sync(bool on) { host_sync_flag().store(on); }
inline std::atomic<size_t> &default_ring_depth_ref()
{
    static std::atomic<size_t> d{4};
    return d;
}
// buffers per block ring for the blocks below: >= 1 bounded (credits); 0 = no pooling, one allocation per message (the round-5 behaviour,
// kept for the before/after measurement of `kpn_tests bench_c2`)
inline void set_default_ring_depth(size_t d) { default_ring_depth_ref().store(d); }
// bytes one ring may have out with the blocks downstream (beyond the first message, which always goes).  Blocks that share the graph
// stream run in the order their threads enqueued, so a producer that runs k messages ahead puts k outputs between a message and the
// kernel that reads it and writes k + 1 buffers in rotation: past the last-level cache (256 MB of Infinity Cache on MI355X) the pair
// "transform, then checksum of its output" takes 80-82 us per 128 MiB message with the reader one or more messages behind, 75 us with
// the reader directly behind, 69 us when in addition the writer gets back the buffer just read (tools/out_buffer_reuse.py; in the
// kernel trace it is the WRITER that runs 45 instead of 52 us, the reader takes 24.5 us throughout).  Half of that cache, measured
// (profiles/r06_kpn_ring_bytes.txt): 2^24-sample messages through dev::fft run at 107 % of the bare launches (which rotate four
// buffers) with one 128 MiB output out, at 92 % with two or four; small messages keep the full depth, which hides the host threads,
// and so do messages larger than the cache, i.e. twice the budget (nothing keeps those there; measured equal or 1-3 % better unbounded);
// a message between the budget and the cache goes out alone.
inline std::atomic<size_t> &default_ring_bytes_ref()
{
    static std::atomic<size_t> b{(size_t)128 << 20};
    return b;
}
inline void set_default_ring_bytes(size_t b) { default_ring_bytes_ref().store(b); }
// The byte budget is a preference about ORDER, never a reason to stop: a consumer that keeps a message while it waits for the next one
// (a block with memory of its own) would otherwise wait for ever on a producer held back by its budget.  A budget wait that lasts this
// long ends with the message going out anyway (a consumer that sat in a long host synchronisation with a message in hand: once); after
// three such waits in a row the ring stops using its budget, and only `depth` bounds it from then on, as it always did.
inline std::atomic<int> &ring_patience_ms_ref()
{
    static std::atomic<int> ms{50};
    return ms;
}
inline void set_ring_patience_ms(int ms) { ring_patience_ms_ref().store(ms); }
inline std::atomic<unsigned long long> &budget_yields_ref() // budget waits of this process that ran out of patience (diagnostics)
{
    static std::atomic<unsigned long long> n{0};
    return n;
}

// a HIP stream with shared ownership: buffers remember the streams that touched them, so a stream outlives its block while a message
// that names it is still in flight
struct StreamRef {
    void *s = nullptr;
    StreamRef() { check(api().stream_create(&s)); }
    ~StreamRef()
    {
        api().stream_sync(s); // a block that ends (hang-up) lets its queued work finish before the stream goes
        api().stream_destroy(s);
    }
    StreamRef(const StreamRef &) = delete;
    StreamRef &operator=(const StreamRef &) = delete;
};
enum StreamPolicy { SHARED = 0, PER_BLOCK = 1 };
inline std::atomic<int> &stream_policy_ref()
{
    static std::atomic<int> p{SHARED};
    return p;
}
inline void set_stream_policy(StreamPolicy p) { stream_policy_ref().store(p); }
// the one compute stream of the calling thread's current device, created on first use, kept for the life of the process
inline std::shared_ptr<StreamRef> graph_stream()
{
    static std::mutex m;
    static std::map<int, std::shared_ptr<StreamRef>> *streams = new std::map<int, std::shared_ptr<StreamRef>>(); // never destroyed: no HIP calls at exit
    int d = 0;
    check(api().get_device(&d));
    std::lock_guard<std::mutex> l(m);
    auto &p = (*streams)[d];
    if (!p) p = std::make_shared<StreamRef>();
    return p;
}

// Every block runs on its own OS thread (src/ratpak.rs:60-185) and enqueues on the stream it holds for its life: COMPUTE blocks on
// the device's shared graph stream (PER_BLOCK policy: one of their own), TRANSFER blocks always on one of their own.
struct BlockStream {
    enum Kind { COMPUTE, TRANSFER };
    std::shared_ptr<StreamRef> ref;
    explicit BlockStream(Kind k = COMPUTE)
        : ref(k == COMPUTE && stream_policy_ref().load() == SHARED ? graph_stream() : std::make_shared<StreamRef>()) {}
    BlockStream(const BlockStream &) = delete;
    BlockStream &operator=(const BlockStream &) = delete;
    operator void *() const { return ref->s; }
};

// one device buffer and what orders its uses: the stream that wrote it last, the streams that have read it since
struct Buf {
    void *ptr = nullptr;
    size_t cap = 0;
    std::mutex m; // the readers of a forked message register from several threads
    std::shared_ptr<StreamRef> writer;
    std::vector<std::shared_ptr<StreamRef>> readers;
    void *event = nullptr; // used under m for a record-there / wait-here pair; may be re-recorded as soon as the wait has been enqueued
    Buf() = default;
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    int grow(size_t bytes)
    {
        if (ptr) api().free_(ptr);
        ptr = nullptr; cap = 0;
        int rc = api().malloc_(&ptr, bytes ? bytes : 1);
        if (rc == REDIO_OK) cap = bytes;
        return rc;
    }
    ~Buf()
    {
        if (ptr) api().free_(ptr);
        if (event) api().event_destroy(event);
    }
    // work enqueued on `mine` from now on runs after everything `other` holds at this moment (m held)
    void order_after(const std::shared_ptr<StreamRef> &other, void *mine)
    {
        if (!other || other->s == mine) return; // same queue: already ordered
        if (!event) check(api().event_create(&event));
        check(api().event_record(event, other->s));
        check(api().stream_wait_event(mine, event));
    }
    void wait_writer(void *mine)
    {
        std::lock_guard<std::mutex> l(m);
        order_after(writer, mine);
    }
    void add_reader(const std::shared_ptr<StreamRef> &st)
    {
        std::lock_guard<std::mutex> l(m);
        for (auto &r : readers) if (r->s == st->s) return;
        readers.push_back(st);
    }
    // ring recycling (nobody else holds the buffer): the new writer's stream waits for the previous message's readers and writer
    void begin_write(void *mine)
    {
        std::lock_guard<std::mutex> l(m);
        for (auto &r : readers) order_after(r, mine);
        if (readers.empty()) order_after(writer, mine);
        readers.clear();
        writer.reset();
    }
    void end_write(const std::shared_ptr<StreamRef> &st)
    {
        std::lock_guard<std::mutex> l(m);
        writer = st;
    }
};

namespace detail {
struct RingState {
    std::mutex m;
    std::condition_variable cv;
    std::vector<std::unique_ptr<Buf>> free; // buffers nobody holds
    size_t depth = 0, live = 0;             // live: buffers that exist (free + leased)
    size_t budget = (size_t)-1, out = 0, out_bytes = 0; // leased buffers and the message bytes in them, against the byte budget
    bool closed = false;
    unsigned long long yields = 0;          // times the byte budget gave way
    int impatient = 0;                      // ... in a row
    unsigned long long grows = 0;           // device allocations this ring made (warm-up, or a message larger than any before)
};
// size classes: the request rounded up to 1/8 of its leading power of two (<= 12.5 % slack), so that messages whose length wobbles
// (the resampler's floor(ratio * len + 1)) keep hitting the same buffers
inline size_t size_class(size_t bytes)
{
    if (bytes <= 256) return 256;
    size_t p = 1;
    while ((p << 1) <= bytes) p <<= 1;
    const size_t q = p >> 3;
    return (bytes + q - 1) / q * q;
}
} // namespace detail

// the lease of one buffer: a message's storage.  Dropping the last handle returns the buffer to its ring (or frees a standalone one).
struct Alloc {
    void *ptr = nullptr;
    size_t bytes = 0;
    std::unique_ptr<Buf> buf;
    std::shared_ptr<detail::RingState> home;
    Alloc() = default;
    // standalone buffer (constants, test fixtures, the unpooled mode): freed when the last handle drops -- hipFree, a device-wide wait
    explicit Alloc(size_t b) : bytes(b), buf(new Buf)
    {
        check(buf->grow(b ? b : 1));
        ptr = buf->ptr;
    }
    Alloc(const Alloc &) = delete;
    Alloc &operator=(const Alloc &) = delete;
    ~Alloc()
    {
        if (!home || !buf) return; // standalone: ~Buf frees
        {
            std::lock_guard<std::mutex> l(home->m);
            --home->out;
            home->out_bytes -= bytes;
            if (home->closed) { --home->live; return; } // the block is gone: ~Buf frees
            home->free.push_back(std::move(buf));
        }
        home->cv.notify_all();
    }
};

template <typename T>
struct View {
    std::shared_ptr<Alloc> mem;
    size_t off = 0, len = 0; // in elements of T
    T *data() const { return reinterpret_cast<T *>(mem->ptr) + off; }
    View sub(size_t o, size_t n) const { return View{mem, off + o, n}; }
};

// scope of a block's reads of one input message on its stream: orders the stream behind the message's writer at entry, registers the
// stream as a reader at exit (the buffer is not rewritten before what this stream holds then).  complete(): the reads are known to
// have finished (the block synchronised with the host), nothing to register.
template <typename T>
struct Reading {
    const View<T> &v;
    const BlockStream &st;
    bool finished = false;
    Reading(const View<T> &view, const BlockStream &stream) : v(view), st(stream)
    {
        if (v.mem && v.mem->buf) v.mem->buf->wait_writer(st);
    }
    void complete() { finished = true; }
    ~Reading()
    {
        if (!finished && v.mem && v.mem->buf) v.mem->buf->add_reader(st.ref);
    }
    Reading(const Reading &) = delete;
    Reading &operator=(const Reading &) = delete;
};

// standalone message (not pooled): constants, fixtures
template <typename T>
View<T> make(size_t n) { return View<T>{std::make_shared<Alloc>(n * sizeof(T)), 0, n}; }

// the per-block ring of output buffers (header comment).  depth 0: every acquire allocates (unpooled).
class Ring {
    std::shared_ptr<detail::RingState> s_;
    bool first_fill_ = true; // acquire() is called by the ring's own block thread only

public:
    explicit Ring(size_t depth = default_ring_depth_ref().load(), size_t byte_budget = default_ring_bytes_ref().load())
        : s_(std::make_shared<detail::RingState>())
    {
        s_->depth = depth;
        s_->budget = byte_budget;
    }
    Ring(const Ring &) = delete;
    Ring &operator=(const Ring &) = delete;
    ~Ring()
    {
        std::vector<std::unique_ptr<Buf>> drop;
        {
            std::lock_guard<std::mutex> l(s_->m);
            s_->closed = true;
            s_->live -= s_->free.size();
            drop.swap(s_->free);
        }
        s_->cv.notify_all();
    } // buffers still held downstream are freed by their last handle
    size_t depth() const { return s_->depth; }
    unsigned long long budget_yields() const
    {
        std::lock_guard<std::mutex> l(s_->m);
        return s_->yields;
    }
    unsigned long long grows() const
    {
        std::lock_guard<std::mutex> l(s_->m);
        return s_->grows;
    }
    // a buffer of n elements for work to be enqueued on `stream`; waits for a credit while all `depth` buffers, or the byte budget's
    // worth of messages, are held downstream
    template <typename T>
    View<T> acquire(size_t n, void *stream)
    {
        const size_t bytes = n * sizeof(T);
        if (s_->depth == 0) return make<T>(n);
        std::unique_ptr<Buf> b;
        {
            std::unique_lock<std::mutex> l(s_->m);
            // a message larger than the cache (twice the budget) cannot be kept there whatever the order: holding its producer back gains
            // nothing; one between the budget and the cache goes alone
            const bool too_big = s_->budget != (size_t)-1 && bytes / 2 > s_->budget;
            const auto in_budget = [&] { return s_->out == 0 || (s_->out < s_->depth && (too_big || s_->out_bytes + bytes <= s_->budget)); };
            while (!in_budget()) {
                if (s_->out >= s_->depth) s_->cv.wait(l); // no credit: the bound
                // (wait_until on the system clock = pthread_cond_timedwait, which ThreadSanitizer follows; wait_for goes through
                // pthread_cond_clockwait, which gcc 11's does not: it then believes the waiter still holds the mutex)
                else if (s_->cv.wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(ring_patience_ms_ref().load()), in_budget))
                    s_->impatient = 0;
                else { // comment at ring_patience_ms_ref: this message goes now (out < depth); three in a row and the budget is off
                    ++s_->yields;
                    ++budget_yields_ref();
                    if (++s_->impatient >= 3) s_->budget = (size_t)-1;
                    break;
                }
            }
            // best fit among the free buffers, the one released last among equals (the likeliest to be in the cache still); else a new
            // one while the ring is not full; else the largest free one is regrown
            size_t best = s_->free.size();
            for (size_t i = 0; i < s_->free.size(); ++i)
                if (s_->free[i]->cap >= bytes && (best == s_->free.size() || s_->free[i]->cap <= s_->free[best]->cap)) best = i;
            if (best == s_->free.size() && s_->live >= s_->depth) {
                best = 0;
                for (size_t i = 1; i < s_->free.size(); ++i) if (s_->free[i]->cap > s_->free[best]->cap) best = i;
            }
            if (best < s_->free.size()) {
                b = std::move(s_->free[best]);
                s_->free.erase(s_->free.begin() + (long)best);
            } else {
                b.reset(new Buf);
                ++s_->live;
            }
            if (b->cap < bytes || !b->ptr) ++s_->grows;
            ++s_->out;
            s_->out_bytes += bytes;
        }
        if (b->cap < bytes || !b->ptr) {
            // rare: warm-up, or the largest message so far (hipFree + hipMalloc).  One eighth of headroom on top of the size class: a block whose
            // messages wobble across a class boundary (the stream blocks send 12 or 13 spectra per 2^16-sample message) then settles at once
            const int rc = b->grow(detail::size_class(bytes + bytes / 8));
            if (rc != REDIO_OK) {
                { std::lock_guard<std::mutex> l(s_->m); --s_->live; --s_->out; s_->out_bytes -= bytes; }
                s_->cv.notify_all();
                check(rc);
            }
        }
        if (first_fill_) { // the ring's first message: the other buffers of its size class now, so that a graph allocates during its
            first_fill_ = false; // first message and never after (a fourth buffer first needed deep into a run would be a late hipMalloc).
            // As many as the byte budget lets out at this message size, and one to spare for lengths that wobble
            const size_t want = bytes / 2 > s_->budget ? s_->depth : std::min(s_->depth, s_->budget / std::max<size_t>(bytes, 1) + 1 + (bytes > s_->budget));
            for (size_t i = 1; i < want; ++i) {
                std::unique_ptr<Buf> extra(new Buf);
                if (extra->grow(b->cap) != REDIO_OK) break; // the ring then holds fewer buffers until a later acquire can allocate
                std::lock_guard<std::mutex> l(s_->m);
                ++s_->live; ++s_->grows;
                s_->free.push_back(std::move(extra));
            }
        }
        auto a = std::make_shared<Alloc>();
        a->ptr = b->ptr;
        a->bytes = bytes;
        a->buf = std::move(b);
        a->home = s_;
        a->buf->begin_write(stream); // the previous message's readers finish before this stream writes
        return View<T>{std::move(a), 0, n};
    }
};

// producer side of every block: the work that fills the message has been enqueued on the block's stream (in the debugging mode: and
// has finished)
template <typename T>
void publish(const View<T> &o, const BlockStream &st)
{
    if (host_sync_flag().load(std::memory_order_relaxed)) {
        check(redio_stream_sync(st));
        o.mem->buf->end_write(nullptr); // complete: nobody needs to wait
    } else o.mem->buf->end_write(st.ref);
}

// host Vec<T> -> device View<T> and back (the only PCIe crossings of a device-resident graph)
template <typename T>
void to_device(Receiver<std::vector<T>> u, Sender<View<T>> v)
{
    BlockStream st(BlockStream::TRANSFER);
    Ring ring(default_ring_depth_ref().load(), (size_t)-1); // no byte budget: the copy engine fills the next message while the graph works on this one
    for (;;) {
        auto x = u.recv();
        auto d = ring.acquire<T>(x.size(), st);
        check(redio_upload(d.data(), x.data(), x.size() * sizeof(T), st));
        check(redio_stream_sync(st)); // the pageable source Vec is freed when this iteration ends
        d.mem->buf->end_write(nullptr); // the copy has finished: consumers have nothing to wait for
        v.send_unwrap(std::move(d));
    }
}
template <typename T>
void to_host(Receiver<View<T>> u, Sender<std::vector<T>> v)
{
    BlockStream st(BlockStream::TRANSFER);
    for (;;) {
        auto d = u.recv();
        std::vector<T> x(d.len);
        {
            Reading<T> in(d, st);
            check(redio_download(x.data(), d.data(), d.len * sizeof(T), st));
            check(redio_stream_sync(st)); // data handed to the CPU: the one place a graph waits on the host
            in.complete();
        }
        v.send_unwrap(std::move(x));
    }
}

// test / bench source: nmsg messages of msg_len hash-generated cf32 samples, message i = samples [i*msg_len, (i+1)*msg_len)
inline void synth_iq_source(Sender<View<std::complex<float>>> v, uint32_t seed, size_t msg_len, size_t nmsg)
{
    BlockStream st;
    Ring ring;
    for (size_t i = 0; i < nmsg; ++i) {
        auto d = ring.acquire<std::complex<float>>(msg_len, st);
        check(redio_synth_iq(d.data(), seed, (uint64_t)i * msg_len, msg_len, st));
        publish(d, st);
        v.send_unwrap(std::move(d));
    }
}

// checking sink: the order-free sum of every 32-bit word received (redio_checksum_u32) and the message count, written at hang-up
template <typename T>
void checksum_sink(Receiver<View<T>> u, unsigned long long *sum, size_t *messages)
{
    BlockStream st;
    auto acc = make<unsigned long long>(1);
    const unsigned long long zero = 0;
    check(redio_upload(acc.data(), &zero, sizeof(zero), st));
    check(redio_stream_sync(st));
    size_t n = 0;
    try {
        for (;;) {
            auto d = u.recv();
            Reading<T> in(d, st);
            check(redio_checksum_u32(d.data(), d.len * sizeof(T) / 4, acc.data(), st));
            ++n;
        }
    } catch (const hangup &) {
    }
    unsigned long long s = 0;
    check(redio_download(&s, acc.data(), sizeof(s), st));
    check(redio_stream_sync(st));
    if (sum) *sum = s;
    if (messages) *messages = n;
}

// kpn::shaper on a View stream: chunks of exactly l samples; zero-copy when possible, otherwise the
// straddling chunk is assembled once on the device.  A trailing partial chunk is dropped at hang-up.
template <typename T>
void shaper(Receiver<View<T>> u, Sender<View<T>> v, size_t l)
{
    BlockStream st;
    Ring ring;
    View<T> pend;           // partially filled chunk (owned copy)
    size_t have = 0;
    for (;;) {
        auto d = u.recv();
        size_t pos = 0;
        while (pos < d.len) {
            if (have == 0 && d.len - pos >= l) { // whole chunk inside this message: a view (the producer's events travel with it)
                v.send_unwrap(d.sub(pos, l));
                pos += l;
                continue;
            }
            if (have == 0) pend = ring.acquire<T>(l, st);
            const size_t take = std::min(l - have, d.len - pos);
            {
                Reading<T> in(d, st);
                check(redio_copy(pend.data() + have, d.data() + pos, take * sizeof(T), st));
            }
            have += take;
            pos += take;
            if (have == l) {
                publish(pend, st);
                v.send_unwrap(std::move(pend));
                pend = View<T>();
                have = 0;
            }
        }
    }
}

namespace detail {
// the loop every stateless hot block shares: recv -> ring buffer of nout(d) samples -> enqueue on the block's stream between the
// input's `ready` wait and its `done` record -> `ready` of the output -> send
template <typename In, typename Out, typename NOut, typename Enqueue>
void run_block(Receiver<View<In>> &u, Sender<View<Out>> &v, NOut nout, Enqueue enqueue)
{
    BlockStream st;
    Ring ring;
    for (;;) {
        auto d = u.recv();
        auto o = ring.acquire<Out>(nout(d), st);
        {
            Reading<In> in(d, st);
            check(enqueue(d, o, (void *)st));
        }
        publish(o, st);
        v.send_unwrap(std::move(o));
    }
}
} // namespace detail

// dsputils::convolve semantics on device streams (complex samples x real taps, optional decimation)
inline void fir(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps, size_t decim, bool fused)
{
    using cf = std::complex<float>;
    redio_fir *h = nullptr;
    check(redio_fir_create(&h, taps.data(), taps.size(), decim, REDIO_FIR_COMPLEX | (fused ? REDIO_FIR_FUSED : 0)));
    struct G { redio_fir *h; ~G() { redio_fir_destroy(h); } } g{h};
    detail::run_block<cf, cf>(u, v, [&](const View<cf> &d) { return redio_fir_nout(h, d.len); },
                              [&](const View<cf> &d, const View<cf> &o, void *st) { return redio_fir_enqueue(h, d.data(), d.len, o.data(), st); });
}

// kissfft::fft semantics: every message must be a whole number of block_size-sample blocks
inline void fft(Receiver<View<std::complex<float>>> pin, Sender<View<std::complex<float>>> cout, uint32_t block_size, uint32_t inv)
{
    using cf = std::complex<float>;
    redio_fft *h = nullptr;
    check(redio_fft_create(&h, (int)block_size, (int)inv));
    struct G { redio_fft *h; ~G() { redio_fft_destroy(h); } } g{h};
    detail::run_block<cf, cf>(pin, cout,
                              [&](const View<cf> &d) {
                                  if (d.len % block_size) throw std::runtime_error("assert!(din.len() == block_size) (kissfft.rs:24)");
                                  return d.len;
                              },
                              [&](const View<cf> &d, const View<cf> &o, void *st) { return redio_fft_enqueue(h, d.data(), o.data(), d.len / block_size, st); });
}

// the fused north-star chain as one block
inline void fir_fft_chain(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps,
                          size_t decim, int nfft, bool fused)
{
    using cf = std::complex<float>;
    redio_chain *h = nullptr;
    check(redio_chain_create(&h, taps.data(), taps.size(), decim, nfft, fused ? REDIO_FIR_FUSED : 0));
    struct G { redio_chain *h; ~G() { redio_chain_destroy(h); } } g{h};
    detail::run_block<cf, cf>(u, v, [&](const View<cf> &d) { return redio_chain_nblocks(h, d.len) * (size_t)nfft; },
                              [&](const View<cf> &d, const View<cf> &o, void *st) { return redio_chain_enqueue(h, d.data(), d.len, o.data(), st); });
}

// the receiver's messages straight into the chain: u8 I/Q bytes (rtlsdr::rtlSource, rtlsdr.rs:127-152) -> data_to_samples
// (rtlsdr.rs:159-162) -> FIR -> FFT as ONE kernel for the north-star shape (redio_chain_enqueue_u8)
inline void bytes_fir_fft_chain(Receiver<View<uint8_t>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps, size_t decim, int nfft,
                                bool fused)
{
    using cf = std::complex<float>;
    redio_chain *h = nullptr;
    check(redio_chain_create(&h, taps.data(), taps.size(), decim, nfft, fused ? REDIO_FIR_FUSED : 0));
    struct G { redio_chain *h; ~G() { redio_chain_destroy(h); } } g{h};
    detail::run_block<uint8_t, cf>(u, v, [&](const View<uint8_t> &d) { return redio_chain_nblocks(h, d.len / 2) * (size_t)nfft; },
                                   [&](const View<uint8_t> &d, const View<cf> &o, void *st) { return redio_chain_enqueue_u8(h, d.data(), d.len, o.data(), st); });
}

// the front end of the shipped graph: u8 IQ bytes -> |x| (rtlsdr.rs:159-162 + ratpak.rs:64-68)
inline void ingest_mag(Receiver<View<uint8_t>> u, Sender<View<float>> v)
{
    detail::run_block<uint8_t, float>(u, v, [](const View<uint8_t> &d) { return d.len / 2; },
                                      [](const View<uint8_t> &d, const View<float> &o, void *st) { return redio_ingest_u8_mag(d.data(), d.len, o.data(), st); });
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
    auto dc = make<T>(c.size());
    check(redio_upload(dc.data(), c.data(), c.size() * sizeof(T), nullptr));
    check(redio_stream_sync(nullptr)); // once, before the first message: the constant is resident from here on
    run_block<T, T>(u, v, [&](const View<T> &x) { return x.len < c.size() ? x.len : c.size(); },
                    [&](const View<T> &x, const View<T> &o, void *st) { return zip_call(add, x.data(), dc.data(), o.data(), o.len, st); });
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
    BlockStream st(BlockStream::TRANSFER); // redio_src_process synchronises its stream (host state machine): keep that off the graph stream
    Ring ring;
    redio_src *h = nullptr;
    int rc = redio_src_create(&h, 1 /* SRC_SINC_MEDIUM_QUALITY, samplerate.rs:27 */, 1);
    if (rc != REDIO_OK) throw std::runtime_error(redio_strerror(rc));
    struct G { redio_src *h; ~G() { redio_src_destroy(h); } } g{h};
    for (;;) {
        auto d = din.recv();
        const long lout = (long)(ratio * (double)d.len + 1.0);
        auto o = ring.acquire<float>((size_t)lout, st);
        long used = 0, gen = 0;
        {
            Reading<float> in(d, st);
            // synchronises the stream itself: the converter's per-output parameters come from its host state machine
            rc = redio_src_process(h, d.data(), (long)d.len, (long)d.len, o.data(), lout, lout, ratio, 0, &used, &gen, st);
            if (rc == REDIO_OK) in.complete();
        }
        if (rc != REDIO_OK) throw std::runtime_error(redio_strerror(rc)); // the reference panics with src_strerror's text
        o.mem->buf->end_write(nullptr); // finished (synchronised above)
        dout.send_unwrap(o.sub(0, (size_t)gen));
    }
}

// the 64-channel polyphase filterbank (BASELINE configs[3]) as a block: rows of 64 channel samples out
inline void channelizer(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> proto, int nchan,
                        int taps_per_branch, bool fused)
{
    using cf = std::complex<float>;
    redio_pfb *h = nullptr;
    check(redio_pfb_create(&h, proto.data(), nchan, taps_per_branch, fused ? REDIO_FIR_FUSED : 0));
    struct G { redio_pfb *h; ~G() { redio_pfb_destroy(h); } } g{h};
    detail::run_block<cf, cf>(u, v, [&](const View<cf> &d) { return redio_pfb_nrows(h, d.len) * (size_t)nchan; },
                              [&](const View<cf> &d, const View<cf> &o, void *st) { return redio_pfb_enqueue(h, d.data(), d.len, o.data(), 1, st); });
}

// overlap-save FFT convolution (BASELINE configs[4]) with dsputils::convolve's valid-mode semantics per message
inline void overlap_save(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps, int nfft)
{
    using cf = std::complex<float>;
    redio_ovsave *h = nullptr;
    check(redio_ovsave_create(&h, taps.data(), taps.size(), nfft));
    struct G { redio_ovsave *h; ~G() { redio_ovsave_destroy(h); } } g{h};
    detail::run_block<cf, cf>(u, v, [&](const View<cf> &d) { return redio_ovsave_nout(h, d.len); },
                              [&](const View<cf> &d, const View<cf> &o, void *st) { return redio_ovsave_enqueue(h, d.data(), d.len, o.data(), st); });
}

// ---- the windowed blocks as STREAMS: history carried across messages on the device (redio_*_stream_*, include/redio.h) ----
// The blocks above keep the reference's per-message semantics (dsputils::convolve is stateless, dsputils.rs:30-32: a
// message seam costs ntaps - 1 outputs).  These carry the stream's unconsumed tail in HBM, so the concatenation of their
// output messages does not depend on how the input was cut into messages; a message that completes no
// output unit sends nothing.  The seam is staged by work on the block's own stream, inside the input's Reading scope, so the
// input buffer may be recycled as soon as that scope's `done` event has passed.
namespace detail {
template <typename NOut, typename Enqueue>
void run_stream_block(Receiver<View<std::complex<float>>> &u, Sender<View<std::complex<float>>> &v, NOut nout, Enqueue enqueue)
{
    using cf = std::complex<float>;
    BlockStream st;
    Ring ring;
    for (;;) {
        auto d = u.recv();
        const size_t n = nout(d.len);
        auto o = n ? ring.acquire<cf>(n, st) : View<cf>();
        size_t got = 0;
        {
            Reading<cf> in(d, st);
            check(enqueue(d, n ? o.data() : nullptr, &got, (void *)st));
        }
        if (n) {
            publish(o, st);
            v.send_unwrap(std::move(o));
        }
    }
}
} // namespace detail

inline void fir_stream(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps, size_t decim, bool fused)
{
    using cf = std::complex<float>;
    redio_fir *h = nullptr;
    check(redio_fir_create(&h, taps.data(), taps.size(), decim, REDIO_FIR_COMPLEX | (fused ? REDIO_FIR_FUSED : 0)));
    redio_fir_stream *s = nullptr;
    const int rc = redio_fir_stream_create(&s, h);
    struct G { redio_fir *h; redio_fir_stream *s; ~G() { redio_fir_stream_destroy(s); redio_fir_destroy(h); } } g{h, s};
    check(rc);
    detail::run_stream_block(u, v, [&](size_t len) { return redio_fir_stream_nout(s, len); },
                             [&](const View<cf> &d, cf *o, size_t *got, void *st) { return redio_fir_stream_enqueue(s, d.data(), d.len, o, got, st); });
}

inline void fir_fft_chain_stream(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps,
                                 size_t decim, int nfft, bool fused)
{
    using cf = std::complex<float>;
    redio_chain *h = nullptr;
    check(redio_chain_create(&h, taps.data(), taps.size(), decim, nfft, fused ? REDIO_FIR_FUSED : 0));
    redio_chain_stream *s = nullptr;
    const int rc = redio_chain_stream_create(&s, h);
    struct G { redio_chain *h; redio_chain_stream *s; ~G() { redio_chain_stream_destroy(s); redio_chain_destroy(h); } } g{h, s};
    check(rc);
    detail::run_stream_block(u, v, [&](size_t len) { return redio_chain_stream_nout(s, len); },
                             [&](const View<cf> &d, cf *o, size_t *got, void *st) { return redio_chain_stream_enqueue(s, d.data(), d.len, o, got, st); });
}

inline void channelizer_stream(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> proto, int nchan,
                               int taps_per_branch, bool fused)
{
    using cf = std::complex<float>;
    redio_pfb *h = nullptr;
    check(redio_pfb_create(&h, proto.data(), nchan, taps_per_branch, fused ? REDIO_FIR_FUSED : 0));
    redio_pfb_stream *s = nullptr;
    const int rc = redio_pfb_stream_create(&s, h);
    struct G { redio_pfb *h; redio_pfb_stream *s; ~G() { redio_pfb_stream_destroy(s); redio_pfb_destroy(h); } } g{h, s};
    check(rc);
    detail::run_stream_block(u, v, [&](size_t len) { return redio_pfb_stream_nout(s, len); },
                             [&](const View<cf> &d, cf *o, size_t *got, void *st) { return redio_pfb_stream_enqueue(s, d.data(), d.len, o, got, st); });
}

inline void overlap_save_stream(Receiver<View<std::complex<float>>> u, Sender<View<std::complex<float>>> v, std::vector<float> taps, int nfft)
{
    using cf = std::complex<float>;
    redio_ovsave *h = nullptr;
    check(redio_ovsave_create(&h, taps.data(), taps.size(), nfft));
    redio_ovsave_stream *s = nullptr;
    const int rc = redio_ovsave_stream_create(&s, h);
    struct G { redio_ovsave *h; redio_ovsave_stream *s; ~G() { redio_ovsave_stream_destroy(s); redio_ovsave_destroy(h); } } g{h, s};
    check(rc);
    detail::run_stream_block(u, v, [&](size_t len) { return redio_ovsave_stream_nout(s, len); },
                             [&](const View<cf> &d, cf *o, size_t *got, void *st) { return redio_ovsave_stream_enqueue(s, d.data(), d.len, o, got, st); });
}

} // namespace dev
} // namespace kpn
