// kpn_tests.cpp -- exercises the C++ kpn twin (include/kpn.hpp, include/wavio.hpp).
//   kpn_tests plumbing            CPU only: channel semantics and every kpn.rs block
//   kpn_tests c1 in.wav out.wav   BASELINE.json configs[0]: WAV -> shaper(1024) -> convolve(63 taps) -> WAV
//   kpn_tests fft in.bin out.bin N inv      kissfft::fft block over raw cf32 messages of N samples
//   kpn_tests resample in.bin out.bin ratio msg_len   samplerate::resample block over raw f32 messages
//   kpn_tests devring msg nmsg depth warm policy host_sync   three device blocks on bounded rings: no allocation after message `warm`, checksum printed
//   kpn_tests bench_c2 log2_msg nmsg depth resident|synth checksum|drop host_sync policy    one JSON line: graph against bare launches
//   kpn_tests bench_c2_list log2:nmsg:depth:host_sync:policy:source:sink ...              the same for a list of points, one process
//   kpn_tests bench_block_list channelizer|ovsave|fft|fir:log2_msg:nmsg ...                the other hot blocks, graph against bare launches
//   environment: KPN_DEV_STREAMS=per_block, KPN_DEV_RING=depth, KPN_DEV_RING_MIB=byte budget of a ring (dev::set_default_ring_bytes)
//   kpn_tests bench_c2_sweep seconds_per_point before lo hi                          message sizes 2^lo ... 2^hi (step 4x)
#include "../../include/kpn.hpp"
#include "../../include/wavio.hpp"
#include <sstream>
#include <iostream>
#include "../../include/kpn_dev.hpp"
#include <atomic>
#include <chrono>
#include <cassert>
#include <cmath>
#include <fstream>

using namespace kpn;

#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "CHECK failed: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

template <typename T>
static std::vector<T> drain(Receiver<T> &r)
{
    std::vector<T> out;
    while (auto v = r.try_recv_blocking()) out.push_back(*v);
    return out;
}
template <typename T>
static void feed(Sender<T> s, std::vector<T> v) { for (auto &x : v) s.send(x); }


// ---- stand-ins for the device calls of dev::DeviceApi: host memory, counting "events" and "streams", so that the rings and the ordering
// logic of include/kpn_dev.hpp (credits, recycling, who waits for whom, hand-over between threads, teardown) run here without a GPU and
// under the sanitizers (tests/san_check.sh) ----
namespace fake {
static std::atomic<long> mallocs{0}, frees{0}, ev_live{0}, records{0}, waits{0}, streams_live{0};
struct Ev { std::atomic<void *> on{nullptr}; };
struct St { int dummy; };
static int malloc_(void **p, size_t b) { *p = std::malloc(b ? b : 1); ++mallocs; return *p ? 0 : REDIO_ERR_NOMEM; }
static int free_(void *p) { std::free(p); ++frees; return 0; }
static int ev_create(void **e) { *e = new Ev; ++ev_live; return 0; }
static int ev_destroy(void *e) { delete (Ev *)e; --ev_live; return 0; }
static int ev_record(void *e, void *st) { ((Ev *)e)->on = st; ++records; return 0; }
static int ev_wait(void *, void *e) { (void)((Ev *)e)->on.load(); ++waits; return 0; }
static int st_create(void **s) { *s = new St; ++streams_live; return 0; }
static int st_destroy(void *s) { delete (St *)s; --streams_live; return 0; }
static int st_sync(void *) { return 0; }
static int get_device(int *d) { *d = 0; return 0; }
struct Install {
    dev::DeviceApi saved;
    Install() : saved(dev::api())
    {
        auto &a = dev::api();
        a.malloc_ = malloc_; a.free_ = free_; a.event_create = ev_create; a.event_destroy = ev_destroy; a.event_record = ev_record;
        a.stream_wait_event = ev_wait; a.stream_create = st_create; a.stream_destroy = st_destroy; a.stream_sync = st_sync; a.get_device = get_device;
        mallocs = frees = ev_live = records = waits = streams_live = 0;
    }
    ~Install() { dev::api() = saved; }
};
} // namespace fake

static int ring_logic()
{
    fake::Install inst;
    using BS = dev::BlockStream;
    { // credits: a ring of 2 parks the third acquire until a handle drops, and hands the SAME buffer back (no allocation).  Producer and
      // consumer on different streams: the consumer orders itself behind the writer (one record THERE + one wait HERE), the recycled
      // buffer's new writer behind the reader
        BS sA(BS::TRANSFER), sB(BS::TRANSFER);
        dev::Ring ring(2);
        auto a = ring.acquire<float>(1000, sA);
        auto b = ring.acquire<float>(1000, sA);
        CHECK(fake::mallocs.load() == 2 && a.data() != b.data() && fake::records.load() == 0);
        float *pa = a.data();
        std::atomic<int> got{0};
        float *pc = nullptr;
        std::thread t([&] { auto c = ring.acquire<float>(1000, sA); pc = c.data(); got = 1; });
        std::this_thread::sleep_for(std::chrono::milliseconds(30));
        CHECK(got.load() == 0);            // no credit: parked
        dev::publish(a, sA);
        { dev::Reading<float> rd(a, sB); }
        CHECK(fake::records.load() == 1 && fake::waits.load() == 1);
        { dev::Reading<float> rd(a, sA); } // the writer's own stream reads it: same queue, nothing to do
        CHECK(fake::records.load() == 1 && fake::waits.load() == 1);
        a = dev::View<float>();            // last handle drops -> buffer back in the ring -> the parked acquire wakes
        t.join();
        CHECK(got.load() == 1 && pc == pa && fake::mallocs.load() == 2);
        CHECK(fake::records.load() == 2 && fake::waits.load() == 2); // the new writer waited for reader sB only (sA is its own queue)
    }
    CHECK(fake::mallocs.load() == fake::frees.load() && fake::ev_live.load() == 0 && fake::streams_live.load() == 0);
    { // the default policy: compute blocks share the device's graph stream -> producer, consumer and recycling need no event at all;
      // a transfer block (own stream) reading the message does
        BS s1, s2, sx(BS::TRANSFER);
        CHECK((void *)s1 == (void *)s2 && (void *)sx != (void *)s1);
        const long r0 = fake::records.load(), w0 = fake::waits.load();
        dev::Ring ring(1);
        for (int i = 0; i < 10; ++i) {
            auto a = ring.acquire<int>(64, s1);
            dev::publish(a, s1);
            { dev::Reading<int> rd(a, s2); }
        }
        CHECK(fake::records.load() == r0 && fake::waits.load() == w0);
        auto a = ring.acquire<int>(64, s1);
        dev::publish(a, s1);
        { dev::Reading<int> rd(a, sx); rd.complete(); }       // e.g. to_host: synchronised with the CPU, nothing left to wait for
        CHECK(fake::records.load() == r0 + 1 && fake::waits.load() == w0 + 1);
        a = dev::View<int>();
        auto b2 = ring.acquire<int>(64, s1);
        CHECK(fake::records.load() == r0 + 1);                 // no reader registered: the recycle is free
        dev::set_stream_policy(dev::PER_BLOCK);
        BS s3, s4;
        CHECK((void *)s3 != (void *)s4);
        dev::set_stream_policy(dev::SHARED);
    }
    { // size classes: a message a little longer reuses the buffer; a much longer one regrows it (one free + one malloc)
        BS sA;
        dev::Ring ring(1);
        { auto a = ring.acquire<uint8_t>(1000, sA); }
        const long m0 = fake::mallocs.load();
        { auto a = ring.acquire<uint8_t>(1010, sA); }
        CHECK(fake::mallocs.load() == m0);
        { auto a = ring.acquire<uint8_t>(5000, sA); }
        CHECK(fake::mallocs.load() == m0 + 1 && ring.grows() == 2);
    }
    { // the byte budget: a ring of 4 with 1000 bytes to have out lets one 600-byte message go at a time (the first always goes, whatever
      // its size) and hands the buffer released last back; 200-byte messages use the whole depth.  Pre-allocation follows the budget
        BS sA;
        const long m0 = fake::mallocs.load();
        dev::set_ring_patience_ms(20000);
        dev::Ring ring(4, 1000);
        auto a = ring.acquire<uint8_t>(600, sA);
        CHECK(fake::mallocs.load() == m0 + 2); // one out at this size, one to spare
        uint8_t *pa = a.data();
        std::atomic<int> got{0};
        uint8_t *pb = nullptr;
        std::thread t([&] { auto b = ring.acquire<uint8_t>(600, sA); pb = b.data(); got = 1; });
        std::this_thread::sleep_for(std::chrono::milliseconds(30));
        CHECK(got.load() == 0);            // 600 + 600 > 1000: parked although three buffers may still be had
        a = dev::View<uint8_t>();
        t.join();
        CHECK(got.load() == 1 && pb == pa && fake::mallocs.load() == m0 + 2);
        { // larger than the cache (twice the budget): it cannot be kept there anyway, so only the depth bounds it
            auto big = ring.acquire<uint8_t>(5000, sA);
            auto big2 = ring.acquire<uint8_t>(5000, sA);
            CHECK(big.len == 5000 && big2.data() != big.data());
        }
        { // between the budget and the cache: goes out alone
            auto mid = ring.acquire<uint8_t>(1500, sA);
            std::atomic<int> got2{0};
            std::thread t2([&] { auto m2 = ring.acquire<uint8_t>(1500, sA); got2 = 1; });
            std::this_thread::sleep_for(std::chrono::milliseconds(30));
            CHECK(got2.load() == 0);
            mid = dev::View<uint8_t>();
            t2.join();
            CHECK(got2.load() == 1);
        }
        std::vector<dev::View<uint8_t>> held;
        for (int i = 0; i < 4; ++i) held.push_back(ring.acquire<uint8_t>(200, sA)); // 800 bytes out: within the budget, the full depth
        CHECK(held.size() == 4);
        std::atomic<int> got5{0};
        std::thread t5([&] { auto e = ring.acquire<uint8_t>(100, sA); got5 = 1; });
        std::this_thread::sleep_for(std::chrono::milliseconds(30));
        CHECK(got5.load() == 0);           // within the budget now, but all four buffers are out
        held.pop_back();
        t5.join();
        CHECK(got5.load() == 1 && ring.budget_yields() == 0);
    }
    { // the budget is a preference about order, not a bound.  A consumer that sits on a message for long (a host synchronisation) costs
      // one wait of the ring's patience and the next message goes out anyway; the budget stays in force
        BS sA;
        dev::set_ring_patience_ms(20);
        dev::Ring ring(3, 1000);
        auto kept = ring.acquire<uint8_t>(600, sA);
        const auto t0 = std::chrono::steady_clock::now();
        auto next = ring.acquire<uint8_t>(600, sA);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CHECK(next.data() != kept.data() && ms >= 15.0 && ring.budget_yields() == 1);
        kept = dev::View<uint8_t>();
        next = dev::View<uint8_t>();
        dev::set_ring_patience_ms(20000);
        auto a = ring.acquire<uint8_t>(600, sA);
        std::atomic<int> got{0};
        std::thread t([&] { auto b = ring.acquire<uint8_t>(600, sA); got = 1; });
        std::this_thread::sleep_for(std::chrono::milliseconds(40));
        CHECK(got.load() == 0);            // still held back by the budget
        a = dev::View<uint8_t>();
        t.join();
        CHECK(got.load() == 1 && ring.budget_yields() == 1);
    }
    { // a consumer that KEEPS every message while it waits for the next (600 + 600 > 1000): three waits in a row run out of patience,
      // then the ring stops using its budget; the depth still binds
        BS sA;
        dev::set_ring_patience_ms(20);
        dev::Ring ring(5, 1000);
        std::vector<dev::View<uint8_t>> kept;
        kept.push_back(ring.acquire<uint8_t>(600, sA));
        for (int i = 0; i < 3; ++i) {
            const auto t0 = std::chrono::steady_clock::now();
            kept.push_back(ring.acquire<uint8_t>(600, sA));
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            CHECK(ms >= 15.0);
        }
        CHECK(ring.budget_yields() == 3);
        const auto t1 = std::chrono::steady_clock::now();
        kept.push_back(ring.acquire<uint8_t>(600, sA));
        const double ms5 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
        CHECK(ms5 < 15.0 && ring.budget_yields() == 3);
        std::atomic<int> got{0};
        std::thread t([&] { auto d = ring.acquire<uint8_t>(600, sA); got = 1; });
        std::this_thread::sleep_for(std::chrono::milliseconds(60));
        CHECK(got.load() == 0);            // five out of five: the depth is a bound
        kept.pop_back();
        t.join();
        CHECK(got.load() == 1);
        dev::set_ring_patience_ms(50);
    }
    { // the ring and the stream go (their block ended) while a message is still held downstream: the last handle frees the buffer,
      // and the stream it names lives until then
        dev::View<int> kept;
        {
            BS sA(BS::TRANSFER);
            dev::Ring ring(3);
            kept = ring.acquire<int>(10, sA);
            dev::publish(kept, sA);
            auto other = ring.acquire<int>(10, sA);
        }
        kept.data()[9] = 7; // still valid memory (ASan would object)
        BS sB(BS::TRANSFER);
        { dev::Reading<int> rd(kept, sB); } // records on the ended block's stream: still alive
    }
    { // depth 0: unpooled, one allocation per message (the before side of `bench_c2`)
        BS sA;
        const long m0 = fake::mallocs.load();
        dev::Ring ring(0);
        for (int i = 0; i < 5; ++i) { auto a = ring.acquire<int>(10, sA); }
        CHECK(fake::mallocs.load() == m0 + 5);
    }
    for (int policy = 0; policy < 2; ++policy) {
        // three block threads, rings of 2, 2000 messages, fork in the middle: every message arrives intact and in order at both sinks,
        // and nothing is allocated after the first few messages (memory written and read by the host threads stands for the kernels)
        dev::set_stream_policy(policy ? dev::PER_BLOCK : dev::SHARED);
        auto [s1, r1] = channel<dev::View<uint32_t>>();
        auto [s2a, r2a] = channel<dev::View<uint32_t>>();
        auto [s2b, r2b] = channel<dev::View<uint32_t>>();
        auto [s3, r3] = channel<dev::View<uint32_t>>();
        const size_t N = 2000, L = 257;
        std::atomic<long> mallocs_at_50{-1};
        std::atomic<int> bad{0};
        std::vector<std::thread> th;
        th.push_back(spawn([s = std::move(s1), N, L]() mutable {
            BS st;
            dev::Ring ring(2);
            for (size_t i = 0; i < N; ++i) {
                auto d = ring.acquire<uint32_t>(L + i % 3, st);
                for (size_t k = 0; k < d.len; ++k) d.data()[k] = (uint32_t)(i * 1000003u + k);
                dev::publish(d, st);
                s.send_unwrap(std::move(d));
            }
        }));
        std::vector<Sender<dev::View<uint32_t>>> outs;
        outs.push_back(std::move(s2a)); outs.push_back(std::move(s2b));
        th.push_back(spawn([r = std::move(r1), o = std::move(outs)]() mutable { fork<dev::View<uint32_t>>(std::move(r), std::move(o)); }));
        th.push_back(spawn([r = std::move(r2a), s = std::move(s3)]() mutable { // a map block: out = in + 1
            BS st;
            dev::Ring ring(2);
            for (;;) {
                auto d = r.recv();
                auto o = ring.acquire<uint32_t>(d.len, st);
                {
                    dev::Reading<uint32_t> in(d, st);
                    for (size_t k = 0; k < d.len; ++k) o.data()[k] = d.data()[k] + 1;
                }
                dev::publish(o, st);
                s.send_unwrap(std::move(o));
            }
        }));
        auto checker = [&bad, &mallocs_at_50, L](Receiver<dev::View<uint32_t>> r, uint32_t add) {
            BS st(BS::TRANSFER);
            size_t i = 0;
            try {
                for (;; ++i) {
                    auto d = r.recv();
                    dev::Reading<uint32_t> in(d, st);
                    if (d.len != L + i % 3) ++bad;
                    for (size_t k = 0; k < d.len; ++k) if (d.data()[k] != (uint32_t)(i * 1000003u + k) + add) { ++bad; break; }
                    if (i == 50) mallocs_at_50 = fake::mallocs.load();
                }
            } catch (const hangup &) {}
            if (i != 2000) ++bad;
        };
        th.push_back(spawn([&, r = std::move(r3)]() mutable { checker(std::move(r), 1); }));
        th.push_back(spawn([&, r = std::move(r2b)]() mutable { checker(std::move(r), 0); }));
        for (auto &t : th) t.join();
        CHECK(bad.load() == 0);
        CHECK(mallocs_at_50.load() > 0 && fake::mallocs.load() == mallocs_at_50.load()); // steady state: no allocation
    }
    dev::set_stream_policy(dev::SHARED);
    CHECK(fake::mallocs.load() == fake::frees.load() && fake::ev_live.load() == 0);
    CHECK(fake::streams_live.load() == 1); // the graph stream of device 0 lives for the process
    std::puts("ring ok");
    return 0;
}

static int plumbing()
{
    if (int rc = ring_logic()) return rc;
    { // channel: FIFO, blocking recv, hang-up when the last sender drops
        auto [tx, rx] = channel<int>();
        auto tx2 = tx;
        tx.send(1); tx2.send(2);
        CHECK(rx.recv() == 1 && rx.recv() == 2);
        { Sender<int> a = std::move(tx); Sender<int> b = std::move(tx2); }
        bool threw = false;
        try { rx.recv(); } catch (const hangup &) { threw = true; }
        CHECK(threw);
    }
    { // bounded channel: send waits for a slot, a dropped receiver releases a blocked sender
        auto [tx, rx] = bounded_channel<int>(2);
        std::atomic<int> sent{0};
        std::thread prod([t = tx, &sent]() mutable { for (int i = 0; i < 6; ++i) { if (!t.send(i)) break; ++sent; } });
        for (int spin = 0; spin < 200 && sent.load() < 2; ++spin) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        CHECK(sent.load() == 2);                       // the third send is parked: no credit
        CHECK(rx.recv() == 0);
        for (int spin = 0; spin < 200 && sent.load() < 3; ++spin) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        CHECK(sent.load() == 3);
        for (int i = 1; i < 6; ++i) CHECK(rx.recv() == i);
        prod.join();
        auto [tx2, rx2] = bounded_channel<int>(1);
        tx2.send(7);
        std::atomic<int> refused{-1};
        std::thread blocked([t = tx2, &refused]() mutable { refused = t.send(8) ? 0 : 1; }); // parks, then sees the hang-up
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        rx2.close();
        blocked.join();
        CHECK(refused.load() == 1);
    }
    { // send to a dropped receiver fails (Err), does not throw
        auto [tx, rx] = channel<int>();
        rx.close();
        CHECK(!tx.send(1));
    }
    { // rle (kpn.rs:17): last run never flushed
        auto [a, ar] = channel<int>(); auto [b, br] = channel<std::pair<int, size_t>>();
        auto t = spawn([&, r = std::move(ar), s = std::move(b)]() mutable { rle<int>(std::move(r), std::move(s)); });
        feed<int>(std::move(a), {1, 1, 1, 0, 0, 1, 1});
        t.join();
        auto o = drain(br);
        CHECK(o.size() == 2 && o[0] == std::make_pair(1, (size_t)3) && o[1] == std::make_pair(0, (size_t)2));
    }
    { // rld(rle) round trip on complete runs + dle/dld
        auto [a, ar] = channel<std::pair<int, size_t>>(); auto [b, br] = channel<int>();
        auto t = spawn([&, r = std::move(ar), s = std::move(b)]() mutable { rld<int>(std::move(r), std::move(s)); });
        feed<std::pair<int, size_t>>(std::move(a), {{7, 2}, {9, 3}});
        t.join();
        auto o = drain(br);
        CHECK((o == std::vector<int>{7, 7, 9, 9, 9}));
        auto [c, cr] = channel<std::pair<int, size_t>>(); auto [d, dr] = channel<std::pair<int, float>>();
        auto t2 = spawn([&, r = std::move(cr), s = std::move(d)]() mutable { dle<int>(std::move(r), std::move(s), 1000); });
        feed<std::pair<int, size_t>>(std::move(c), {{1, 500}});
        t2.join();
        auto o2 = drain(dr);
        CHECK(o2.size() == 1 && o2[0].second == 0.5f);
        auto [e, er] = channel<std::pair<int, float>>(); auto [g, gr] = channel<int>();
        auto t3 = spawn([&, r = std::move(er), s = std::move(g)]() mutable { dld<int>(std::move(r), std::move(s), 8.0f); });
        feed<std::pair<int, float>>(std::move(e), {{3, 0.5f}});
        t3.join();
        CHECK(drain(gr).size() == 4);
    }
    { // differentiator (first value never emitted), dxdt (keeps the difference)
        auto [a, ar] = channel<int>(); auto [b, br] = channel<int>();
        auto t = spawn([&, r = std::move(ar), s = std::move(b)]() mutable { differentiator<int>(std::move(r), std::move(s)); });
        feed<int>(std::move(a), {5, 5, 6, 6, 5});
        t.join();
        CHECK((drain(br) == std::vector<int>{6, 5}));
        auto [c, cr] = channel<float>(); auto [d, dr] = channel<float>();
        auto t2 = spawn([&, r = std::move(cr), s = std::move(d)]() mutable { dxdt<float>(std::move(r), std::move(s)); });
        feed<float>(std::move(c), {1.f, 4.f, 10.f});
        t2.join();
        CHECK((drain(dr) == std::vector<float>{3.f, 7.f})); // 4-1, then 10-3 (as written)
    }
    { // shaper drops a trailing partial block; shaper_vecs / unpacketizer flatten
        auto [a, ar] = channel<int>(); auto [b, br] = channel<std::vector<int>>();
        auto t = spawn([&, r = std::move(ar), s = std::move(b)]() mutable { shaper<int>(std::move(r), std::move(s), 3); });
        feed<int>(std::move(a), {1, 2, 3, 4, 5, 6, 7});
        t.join();
        auto blocks = drain(br);
        CHECK(blocks.size() == 2 && blocks[1] == (std::vector<int>{4, 5, 6}));
        auto [c, cr] = channel<std::vector<int>>(); auto [d, dr] = channel<int>();
        auto t2 = spawn([&, r = std::move(cr), s = std::move(d)]() mutable { shaper_vecs<int>(std::move(r), std::move(s)); });
        feed<std::vector<int>>(std::move(c), blocks);
        t2.join();
        CHECK((drain(dr) == std::vector<int>{1, 2, 3, 4, 5, 6}));
        auto [e, er] = channel<std::vector<int>>(); auto [g, gr] = channel<int>();
        auto t3 = spawn([&, r = std::move(er), s = std::move(g)]() mutable { unpacketizer<int>(std::move(r), std::move(s)); });
        feed<std::vector<int>>(std::move(e), blocks);
        t3.join();
        CHECK(drain(gr).size() == 6);
    }
    { // shaper_optional: emit only exact-length groups
        auto [a, ar] = channel<std::optional<int>>(); auto [b, br] = channel<std::vector<int>>();
        auto t = spawn([&, r = std::move(ar), s = std::move(b)]() mutable { shaper_optional<int>(std::move(r), std::move(s), 2); });
        feed<std::optional<int>>(std::move(a), {1, 2, std::nullopt, 3, std::nullopt, 4, 5, std::nullopt});
        t.join();
        auto o = drain(br);
        CHECK(o.size() == 2 && o[1] == (std::vector<int>{4, 5}));
    }
    { // fork, mul, sum, mul_vecs (zip truncation), sum_vecs, delay, applicator, cross_applicator(_vecs), looper_optional
        auto [a, ar] = channel<float>(); auto [b, br] = channel<float>(); auto [c, cr] = channel<float>();
        std::vector<Sender<float>> outs; outs.push_back(std::move(b)); outs.push_back(std::move(c));
        auto t = spawn([&, r = std::move(ar), o = std::move(outs)]() mutable { fork<float>(std::move(r), std::move(o)); });
        feed<float>(std::move(a), {1.f, 2.f});
        t.join();
        CHECK(drain(br).size() == 2 && drain(cr).size() == 2);
        auto [d, dr] = channel<float>(); auto [e, er] = channel<float>();
        auto t2 = spawn([&, r = std::move(dr), s = std::move(e)]() mutable { mul<float>(std::move(r), std::move(s), 3.f); });
        feed<float>(std::move(d), {2.f});
        t2.join();
        CHECK(drain(er)[0] == 6.f);
        auto [g, gr] = channel<std::vector<float>>(); auto [h, hr] = channel<std::vector<float>>();
        auto t3 = spawn([&, r = std::move(gr), s = std::move(h)]() mutable { mul_vecs<float>(std::move(r), std::move(s), {2.f, 3.f}); });
        feed<std::vector<float>>(std::move(g), {{1.f, 1.f, 1.f}});
        t3.join();
        auto mv = drain(hr);
        CHECK(mv[0] == (std::vector<float>{2.f, 3.f}));
        auto [i, ir] = channel<int>(); auto [j, jr] = channel<int>();
        auto t4 = spawn([&, r = std::move(ir), s = std::move(j)]() mutable { delay<int>(std::move(r), std::move(s), -1); });
        feed<int>(std::move(i), {1, 2});
        t4.join();
        CHECK((drain(jr) == std::vector<int>{-1, 1, 2}));
        auto [k, kr] = channel<int>(); auto [l, lr] = channel<double>();
        auto t5 = spawn([&, r = std::move(kr), s = std::move(l)]() mutable {
            cross_applicator<int, double>(std::move(r), std::move(s), [](int x) { return x * 0.5; }); });
        feed<int>(std::move(k), {3});
        t5.join();
        CHECK(drain(lr)[0] == 1.5);
        auto [m, mr] = channel<std::optional<int>>(); auto [n, nr] = channel<int>();
        auto t6 = spawn([&, r = std::move(mr), s = std::move(n)]() mutable { looper_optional<int>(std::move(r), std::move(s)); });
        feed<std::optional<int>>(std::move(m), {1, std::nullopt, 2});
        t6.join();
        CHECK((drain(nr) == std::vector<int>{1, 2}));
    }
    { // the remaining map / plumbing blocks: applicator_vecs, cross_applicator_vecs, delay_vecs, looper, soft_source, print_sink
        auto [a, ar] = channel<std::vector<int>>(); auto [b, br] = channel<std::vector<int>>();
        auto t = spawn([&, r = std::move(ar), s = std::move(b)]() mutable { applicator_vecs<int>(std::move(r), std::move(s), [](const int &x) { return x * x; }); });
        feed<std::vector<int>>(std::move(a), {{1, 2, 3}, {}, {4}});
        t.join();
        auto o = drain(br);
        CHECK(o.size() == 3 && o[0] == (std::vector<int>{1, 4, 9}) && o[1].empty() && o[2] == (std::vector<int>{16}));   // kpn.rs:134-138
        auto [c, cr] = channel<std::vector<int>>(); auto [d, dr] = channel<std::vector<double>>();
        auto t2 = spawn([&, r = std::move(cr), s = std::move(d)]() mutable { cross_applicator_vecs<int, double>(std::move(r), std::move(s), [](const int &x) { return 0.5 * x; }); });
        feed<std::vector<int>>(std::move(c), {{1, 2}, {3}});
        t2.join();
        auto o2 = drain(dr);
        CHECK(o2.size() == 2 && o2[0] == (std::vector<double>{0.5, 1.0}) && o2[1] == (std::vector<double>{1.5}));          // kpn.rs:170-174
        auto [e, er] = channel<std::vector<int>>(); auto [g, gr] = channel<std::vector<int>>();
        auto t3 = spawn([&, r = std::move(er), s = std::move(g)]() mutable { delay_vecs<std::vector<int>>(std::move(r), std::move(s), std::vector<int>{7, 7}); });
        feed<std::vector<int>>(std::move(e), {{1}, {2, 3}});
        t3.join();
        auto o3 = drain(gr);
        CHECK(o3.size() == 3 && o3[0] == (std::vector<int>{7, 7}) && o3[2] == (std::vector<int>{2, 3}));                 // kpn.rs:261-263: the constant first
        auto [h, hr] = channel<int>(); auto [k, kr] = channel<long>();
        auto t4 = spawn([&, r = std::move(hr), s = std::move(k)]() mutable {
            looper<int, long>(std::move(r), std::move(s), [](Receiver<int> &in, Sender<long> &out) { long acc = 0; for (;;) { acc += in.recv(); out.send_unwrap(acc); } });
        });
        feed<int>(std::move(h), {1, 2, 3, 4});
        t4.join();
        CHECK((drain(kr) == std::vector<long>{1, 3, 6, 10}));                                                             // kpn.rs:148-150: the closure owns the stream
        auto [m, mr] = channel<int>();
        // kpn.rs:141-145: the closure runs, then the block parks FOREVER so that its sender never hangs up -- the thread is detached, not joined
        std::thread([s = std::move(m)]() mutable { soft_source<int>(std::move(s), [](Sender<int> &v) { for (int i = 0; i < 3; ++i) v.send(i); }); }).detach();
        CHECK(mr.recv() == 0 && mr.recv() == 1 && mr.recv() == 2);
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
        CHECK(!mr.try_recv());   // nothing more, and no hang-up either: a recv() here would block, as downstream of the reference's source
        auto [p, pr] = channel<int>();
        std::ostringstream cap; auto *old = std::cout.rdbuf(cap.rdbuf());
        auto t6 = spawn([&, r = std::move(pr)]() mutable { print_sink<int>(std::move(r)); });
        feed<int>(std::move(p), {5, 6});
        t6.join();
        std::cout.rdbuf(old);
        CHECK(cap.str() == "5\n6\n");                                                                                    // kpn.rs:104-108: println per item
    }
    { // sum_across / mul_across / sum_across_vecs
        auto [a, ar] = channel<float>(); auto [b, br] = channel<float>(); auto [o, orx] = channel<float>();
        std::vector<Receiver<float>> ins; ins.push_back(std::move(ar)); ins.push_back(std::move(br));
        auto t = spawn([&, r = std::move(ins), s = std::move(o)]() mutable { sum_across<float>(std::move(r), std::move(s), 10.f); });
        feed<float>(std::move(a), {1.f, 2.f}); feed<float>(std::move(b), {3.f, 4.f});
        t.join();
        CHECK((drain(orx) == std::vector<float>{14.f, 16.f}));
    }
    { // grapes panics on an empty input (kpn.rs:245)
        auto [a, ar] = channel<int>(); auto [o, orx] = channel<int>();
        std::vector<Receiver<int>> ins; ins.push_back(std::move(ar));
        bool threw = false;
        try { grapes<int>(std::move(ins), std::move(o)); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw);
        (void)a; (void)orx;
    }
    // b2d / eat / binconv (kpn.rs:111-124,295-299) with the width lists of ratpak.rs:115,119
    CHECK(b2d({1, 0, 1}) == 5);
    {
        std::vector<size_t> bits(36, 0);
        bits[3] = 1; bits[11] = 1; bits[35] = 1; // 0001 00000001 0000 000000000000 00000001
        auto f = eat(bits, {4, 8, 4, 12, 8});
        CHECK((f == std::vector<size_t>{1, 1, 0, 0, 1}));
        auto g = eat(bits, {4, 8, 2, 10, 12});
        CHECK(g.size() == 5 && g[4] == 1);
        bool threw = false;
        try { eat(bits, {30, 10}); } catch (const std::out_of_range &) { threw = true; }
        CHECK(threw);
    }
    { // WAV round trip (float32 mono + stereo-as-IQ, PCM16 read)
        std::vector<float> x(4096);
        for (size_t i = 0; i < x.size(); ++i) x[i] = std::sin(0.01f * (float)i);
        wavio::write_wav_f32("/tmp/kpn_rt.wav", x, 48000, 1);
        auto [a, ar] = channel<float>();
        auto t = spawn([&, s = std::move(a)]() mutable { wavio::wav_source_f32(std::move(s), "/tmp/kpn_rt.wav", 48000); });
        t.join();
        CHECK(drain(ar) == x);
        auto [b, br] = channel<float>();
        auto t2 = spawn([&, s = std::move(b)]() mutable { wavio::wav_source_f32(std::move(s), "/tmp/kpn_rt.wav", 48000, true); });
        t2.join();
        CHECK(drain(br).size() == 2048); // the reference's (frames/2)/1024 chunk count
        bool threw = false;
        try { auto [c, cr] = channel<float>(); wavio::wav_source_f32(std::move(c), "/tmp/kpn_rt.wav", 44100); } catch (const std::runtime_error &) { threw = true; }
        CHECK(threw); // assert_eq!(samplerate)
        wavio::write_wav_f32("/tmp/kpn_iq.wav", x, 48000, 2);
        auto [d, dr] = channel<std::complex<float>>();
        auto t3 = spawn([&, s = std::move(d)]() mutable { wavio::wav_source_complex_f32(std::move(s), "/tmp/kpn_iq.wav", 48000); });
        t3.join();
        auto iq = drain(dr);
        CHECK(iq.size() == 2048 && iq[1] == std::complex<float>(x[2], x[3]));
    }
    { // malformed WAV files: a clean std::runtime_error or the frames that are there -- never a read or an allocation from an unchecked field
        auto put = [](const std::string &name, const std::vector<uint8_t> &b) { FILE *f = std::fopen(name.c_str(), "wb"); if (!b.empty()) std::fwrite(b.data(), 1, b.size(), f); std::fclose(f); };
        auto u32 = [](std::vector<uint8_t> &b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); };
        auto u16 = [](std::vector<uint8_t> &b, uint16_t v) { b.push_back((uint8_t)v); b.push_back((uint8_t)(v >> 8)); };
        auto tag = [](std::vector<uint8_t> &b, const char *t) { b.insert(b.end(), t, t + 4); };
        auto header = [&](uint32_t fmt_sz, uint16_t format, uint16_t ch, uint16_t bits, uint32_t data_sz, size_t data_present) {
            std::vector<uint8_t> b; tag(b, "RIFF"); u32(b, 36 + data_sz); tag(b, "WAVE"); tag(b, "fmt "); u32(b, fmt_sz);
            std::vector<uint8_t> fm; u16(fm, format); u16(fm, ch); u32(fm, 48000); u32(fm, 48000u * ch * (bits / 8)); u16(fm, (uint16_t)(ch * (bits / 8))); u16(fm, bits);
            fm.resize(fmt_sz < 64 ? fmt_sz : 64, 0);
            b.insert(b.end(), fm.begin(), fm.end());
            if (fmt_sz & 1) b.push_back(0);
            tag(b, "data"); u32(b, data_sz);
            for (size_t i = 0; i < data_present; ++i) b.push_back((uint8_t)(i * 37 + 1));
            return b;
        };
        auto outcome = [&](const std::vector<uint8_t> &b, size_t *frames = nullptr) { // 0: read, 1: runtime_error
            put("/tmp/kpn_bad.wav", b);
            wavio::WavInfo info;
            try { auto v = wavio::read_wav("/tmp/kpn_bad.wav", info); if (frames) *frames = info.frames; CHECK(v.size() == info.frames * info.channels); return 0; }
            catch (const std::runtime_error &) { return 1; }
        };
        size_t fr = 0;
        CHECK(outcome(header(16, 3, 1, 32, 400, 400), &fr) == 0 && fr == 100);          // well formed
        CHECK(outcome(header(16, 1, 2, 16, 400, 400), &fr) == 0 && fr == 100);          // PCM16 stereo
        CHECK(outcome(header(8, 3, 1, 32, 400, 400)) == 1);                             // format chunk shorter than its fixed part
        CHECK(outcome(header(0xFFFFFFF0u, 3, 1, 32, 400, 400)) == 1);                   // ... or absurdly long
        CHECK(outcome(header(16, 3, 0, 32, 400, 400)) == 1);                            // no channels
        CHECK(outcome(header(16, 3, 1, 0, 400, 400)) == 1);                             // zero bits per sample
        CHECK(outcome(header(16, 1, 1, 24, 400, 400)) == 1);                            // a width it does not convert
        CHECK(outcome(header(16, 3, 2, 32, 0xFFFFFFFFu, 404), &fr) == 0 && fr == 50);   // streaming writer's size: the whole frames present (404 bytes = 50 frames + 4)
        CHECK(outcome(header(16, 3, 1, 32, 400, 123), &fr) == 0 && fr == 30);           // truncated data
        CHECK(outcome(header(17, 3, 1, 32, 40, 40), &fr) == 0 && fr == 10);             // odd-sized format chunk, pad byte skipped
        CHECK(outcome({'R', 'I', 'F', 'F', 1, 0, 0, 0, 'W', 'A', 'V'}) == 1);           // cut inside the magic
        CHECK(outcome({}) == 1);
        { std::vector<uint8_t> b; tag(b, "RIFF"); u32(b, 4); tag(b, "WAVE"); tag(b, "data"); u32(b, 8); b.resize(b.size() + 8, 1); CHECK(outcome(b) == 1); } // data before fmt
        // a seeded mutation run over a good file: every outcome is "read" or "runtime_error" (under ASan in tests/san_check.sh: no bad access)
        uint32_t st = 12345; auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
        const auto good = header(18, 3, 2, 32, 800, 800);
        int reads = 0, errors = 0;
        for (int it = 0; it < 3000; ++it) {
            auto b = good;
            const int nmut = 1 + (int)(rnd() % 4);
            for (int m = 0; m < nmut; ++m) {
                const size_t pos = rnd() % 64; // the header region
                switch (rnd() % 4) {
                case 0: b[pos] = (uint8_t)rnd(); break;
                case 1: b[pos] = 0; break;
                case 2: b[pos] = 0xFF; break;
                default: b.resize(rnd() % (b.size() + 1)); break;
                }
                if (b.size() <= pos) break;
            }
            (outcome(b) == 0 ? reads : errors)++;
        }
        CHECK(reads > 100 && errors > 100);
    }
    { // a block that throws mid-stream is that task's panic only: the process lives and the hang-up cascades
        auto [a, ar] = channel<int>();
        auto [b, br] = channel<int>();
        auto [c, cr] = channel<int>();
        auto t1 = spawn([s = std::move(a)]() mutable { for (int i = 0; i < 10; ++i) s.send(i); });
        auto t2 = spawn([r = std::move(ar), s = std::move(b)]() mutable {
            for (;;) {
                int v = r.recv();
                if (v == 3) throw std::out_of_range("block failure injected by the test");
                s.send_unwrap(v);
            }
        });
        auto t3 = spawn([r = std::move(br), s = std::move(c)]() mutable { for (;;) s.send_unwrap(r.recv() * 2); });
        t1.join(); t2.join(); t3.join();
        CHECK((drain(cr) == std::vector<int>{0, 2, 4}));
    }
    std::puts("plumbing ok");
    return 0;
}

// BASELINE.json configs[0]: wav in -> shaper(1024) -> cross_applicator(convolve 63 taps) -> shaper_vecs -> wav out
static int c1(const char *in, const char *out)
{
    const std::vector<float> taps = dsputils::lpf_corrected(63, 0.1f);
    auto [s0, r0] = channel<float>();
    auto [s1, r1] = channel<std::vector<float>>();
    auto [s2, r2] = channel<std::vector<float>>();
    auto [s3, r3] = channel<float>();
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s0), in]() mutable { wavio::wav_source_f32(std::move(s), in, 48000); }));
    th.push_back(spawn([r = std::move(r0), s = std::move(s1)]() mutable { shaper<float>(std::move(r), std::move(s), 1024); }));
    th.push_back(spawn([r = std::move(r1), s = std::move(s2), &taps]() mutable {
        cross_applicator<std::vector<float>, std::vector<float>>(std::move(r), std::move(s),
                                                                 [&taps](std::vector<float> x) { return dsputils::convolve(x, taps); });
    }));
    th.push_back(spawn([r = std::move(r2), s = std::move(s3)]() mutable { shaper_vecs<float>(std::move(r), std::move(s)); }));
    th.push_back(spawn([r = std::move(r3), out]() mutable { wavio::wav_sink_f32(std::move(r), out, 48000); }));
    for (auto &t : th) t.join();
    return 0;
}

template <typename T>
static std::vector<T> read_bin(const char *fn)
{
    std::ifstream f(fn, std::ios::binary | std::ios::ate);
    size_t n = (size_t)f.tellg() / sizeof(T);
    std::vector<T> v(n);
    f.seekg(0);
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(n * sizeof(T)));
    return v;
}
template <typename T>
static void write_bin(const char *fn, const std::vector<T> &v)
{
    std::ofstream f(fn, std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}

static int fft_graph(const char *in, const char *out, uint32_t n, uint32_t inv)
{
    auto x = read_bin<std::complex<float>>(in);
    auto [s0, r0] = channel<std::complex<float>>();
    auto [s1, r1] = channel<std::vector<std::complex<float>>>();
    auto [s2, r2] = channel<std::vector<std::complex<float>>>();
    auto [s3, r3] = channel<std::complex<float>>();
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s0), &x]() mutable { for (auto &v : x) s.send(v); }));
    th.push_back(spawn([r = std::move(r0), s = std::move(s1), n]() mutable { shaper<std::complex<float>>(std::move(r), std::move(s), n); }));
    th.push_back(spawn([r = std::move(r1), s = std::move(s2), n, inv]() mutable { kissfft::fft(std::move(r), std::move(s), n, inv); }));
    th.push_back(spawn([r = std::move(r2), s = std::move(s3)]() mutable { shaper_vecs<std::complex<float>>(std::move(r), std::move(s)); }));
    std::vector<std::complex<float>> y;
    while (auto v = r3.try_recv_blocking()) y.push_back(*v);
    for (auto &t : th) t.join();
    write_bin(out, y);
    return 0;
}

static int resample_graph(const char *in, const char *out, double ratio, size_t msg)
{
    auto x = read_bin<float>(in);
    auto [s0, r0] = channel<float>();
    auto [s1, r1] = channel<std::vector<float>>();
    auto [s2, r2] = channel<std::vector<float>>();
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s0), &x]() mutable { for (auto &v : x) s.send(v); }));
    th.push_back(spawn([r = std::move(r0), s = std::move(s1), msg]() mutable { shaper<float>(std::move(r), std::move(s), msg); }));
    th.push_back(spawn([r = std::move(r1), s = std::move(s2), ratio]() mutable { samplerate::resample(std::move(r), std::move(s), ratio); }));
    std::vector<float> y;
    while (auto v = r2.try_recv_blocking()) y.insert(y.end(), v->begin(), v->end());
    for (auto &t : th) t.join();
    write_bin(out, y);
    return 0;
}

// device-resident graph: host messages -> to_device -> dev::shaper(5120*nb+126 views) is not needed here;
// the chain consumes each message whole: bytes/IQ stay in HBM between blocks, fork shares one allocation.
static int dev_chain_graph(const char *in, const char *out_spec, const char *out_fir, size_t msg)
{
    using cf = std::complex<float>;
    auto x = read_bin<cf>(in);
    const std::vector<float> taps = dsputils::lpf_corrected(127, 0.08f);
    auto [s0, r0] = channel<std::vector<cf>>();
    auto [s1, r1] = channel<dev::View<cf>>();
    auto [s2a, r2a] = channel<dev::View<cf>>();
    auto [s2b, r2b] = channel<dev::View<cf>>();
    auto [s3a, r3a] = channel<dev::View<cf>>();
    auto [s3b, r3b] = channel<dev::View<cf>>();
    auto [s4a, r4a] = channel<std::vector<cf>>();
    auto [s4b, r4b] = channel<std::vector<cf>>();
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s0), &x, msg]() mutable {
        for (size_t o = 0; o + msg <= x.size(); o += msg) s.send(std::vector<cf>(x.begin() + (long)o, x.begin() + (long)(o + msg)));
    }));
    th.push_back(spawn([r = std::move(r0), s = std::move(s1)]() mutable { dev::to_device<cf>(std::move(r), std::move(s)); }));
    std::vector<Sender<dev::View<cf>>> outs;
    outs.push_back(std::move(s2a)); outs.push_back(std::move(s2b));
    th.push_back(spawn([r = std::move(r1), o = std::move(outs)]() mutable { fork<dev::View<cf>>(std::move(r), std::move(o)); })); // zero-copy
    th.push_back(spawn([r = std::move(r2a), s = std::move(s3a), taps]() mutable { dev::fir_fft_chain(std::move(r), std::move(s), taps, 5, 1024, true); }));
    th.push_back(spawn([r = std::move(r2b), s = std::move(s3b), taps]() mutable { dev::fir(std::move(r), std::move(s), taps, 5, true); }));
    th.push_back(spawn([r = std::move(r3a), s = std::move(s4a)]() mutable { dev::to_host<cf>(std::move(r), std::move(s)); }));
    th.push_back(spawn([r = std::move(r3b), s = std::move(s4b)]() mutable { dev::to_host<cf>(std::move(r), std::move(s)); }));
    std::vector<cf> spec, fir;
    while (auto v = r4a.try_recv_blocking()) spec.insert(spec.end(), v->begin(), v->end());
    while (auto v = r4b.try_recv_blocking()) fir.insert(fir.end(), v->begin(), v->end());
    for (auto &t : th) t.join();
    write_bin(out_spec, spec);
    write_bin(out_fir, fir);
    return 0;
}

// dev::shaper: re-chunk a View stream (zero-copy inside a message, assembled across a seam)
static int dev_shaper_graph(const char *in, const char *out, size_t msg, size_t l)
{
    auto x = read_bin<float>(in);
    auto [s0, r0] = channel<std::vector<float>>();
    auto [s1, r1] = channel<dev::View<float>>();
    auto [s2, r2] = channel<dev::View<float>>();
    auto [s3, r3] = channel<std::vector<float>>();
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s0), &x, msg]() mutable {
        for (size_t o = 0; o < x.size(); o += msg) s.send(std::vector<float>(x.begin() + (long)o, x.begin() + (long)std::min(o + msg, x.size())));
    }));
    th.push_back(spawn([r = std::move(r0), s = std::move(s1)]() mutable { dev::to_device<float>(std::move(r), std::move(s)); }));
    th.push_back(spawn([r = std::move(r1), s = std::move(s2), l]() mutable { dev::shaper<float>(std::move(r), std::move(s), l); }));
    th.push_back(spawn([r = std::move(r2), s = std::move(s3)]() mutable { dev::to_host<float>(std::move(r), std::move(s)); }));
    std::vector<float> y;
    size_t nmsg = 0;
    while (auto v = r3.try_recv_blocking()) { if (v->size() != l) return 4; y.insert(y.end(), v->begin(), v->end()); ++nmsg; }
    for (auto &t : th) t.join();
    write_bin(out, y);
    return 0;
}

// device-resident elementwise + resampler + channelizer blocks: f32 stream -> sum_vecs -> mul_vecs -> resample
static int dev_mix_graph(const char *in, const char *out, size_t msg, double ratio)
{
    auto x = read_bin<float>(in);
    std::vector<float> c(msg), c2(msg);
    for (size_t i = 0; i < msg; ++i) { c[i] = (float)(i % 7) * 0.25f - 0.5f; c2[i] = 1.0f + (float)(i % 5) * 0.125f; }
    auto [s0, r0] = channel<std::vector<float>>();
    auto [s1, r1] = channel<dev::View<float>>();
    auto [s2, r2] = channel<dev::View<float>>();
    auto [s3, r3] = channel<dev::View<float>>();
    auto [s4, r4] = channel<dev::View<float>>();
    auto [s5, r5] = channel<std::vector<float>>();
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s0), &x, msg]() mutable {
        for (size_t o = 0; o < x.size(); o += msg) s.send(std::vector<float>(x.begin() + (long)o, x.begin() + (long)std::min(o + msg, x.size())));
    }));
    th.push_back(spawn([r = std::move(r0), s = std::move(s1)]() mutable { dev::to_device<float>(std::move(r), std::move(s)); }));
    th.push_back(spawn([r = std::move(r1), s = std::move(s2), c]() mutable { dev::sum_vecs<float>(std::move(r), std::move(s), c); }));
    th.push_back(spawn([r = std::move(r2), s = std::move(s3), c2]() mutable { dev::mul_vecs<float>(std::move(r), std::move(s), c2); }));
    th.push_back(spawn([r = std::move(r3), s = std::move(s4), ratio]() mutable { dev::resample(std::move(r), std::move(s), ratio); }));
    th.push_back(spawn([r = std::move(r4), s = std::move(s5)]() mutable { dev::to_host<float>(std::move(r), std::move(s)); }));
    std::vector<float> y;
    while (auto v = r5.try_recv_blocking()) y.insert(y.end(), v->begin(), v->end());
    for (auto &t : th) t.join();
    write_bin(out, y);
    return 0;
}

// cf32 stream -> 64-channel channelizer and -> overlap-save, both device-resident, from one forked View
static int dev_bank_graph(const char *in, const char *out_pfb, const char *out_ovs, size_t msg)
{
    using cf = std::complex<float>;
    auto x = read_bin<cf>(in);
    const std::vector<float> proto = dsputils::lpf_corrected(64 * 16, 0.45f / 64.0f), taps = dsputils::lpf_corrected(127, 0.08f);
    auto [s0, r0] = channel<std::vector<cf>>();
    auto [s1, r1] = channel<dev::View<cf>>();
    auto [s2a, r2a] = channel<dev::View<cf>>();
    auto [s2b, r2b] = channel<dev::View<cf>>();
    auto [s3a, r3a] = channel<dev::View<cf>>();
    auto [s3b, r3b] = channel<dev::View<cf>>();
    auto [s4a, r4a] = channel<std::vector<cf>>();
    auto [s4b, r4b] = channel<std::vector<cf>>();
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s0), &x, msg]() mutable {
        for (size_t o = 0; o + msg <= x.size(); o += msg) s.send(std::vector<cf>(x.begin() + (long)o, x.begin() + (long)(o + msg)));
    }));
    th.push_back(spawn([r = std::move(r0), s = std::move(s1)]() mutable { dev::to_device<cf>(std::move(r), std::move(s)); }));
    std::vector<Sender<dev::View<cf>>> outs;
    outs.push_back(std::move(s2a)); outs.push_back(std::move(s2b));
    th.push_back(spawn([r = std::move(r1), o = std::move(outs)]() mutable { fork<dev::View<cf>>(std::move(r), std::move(o)); }));
    th.push_back(spawn([r = std::move(r2a), s = std::move(s3a), proto]() mutable { dev::channelizer(std::move(r), std::move(s), proto, 64, 16, true); }));
    th.push_back(spawn([r = std::move(r2b), s = std::move(s3b), taps]() mutable { dev::overlap_save(std::move(r), std::move(s), taps, 4096); }));
    th.push_back(spawn([r = std::move(r3a), s = std::move(s4a)]() mutable { dev::to_host<cf>(std::move(r), std::move(s)); }));
    th.push_back(spawn([r = std::move(r3b), s = std::move(s4b)]() mutable { dev::to_host<cf>(std::move(r), std::move(s)); }));
    std::vector<cf> a, b;
    while (auto v = r4a.try_recv_blocking()) a.insert(a.end(), v->begin(), v->end());
    while (auto v = r4b.try_recv_blocking()) b.insert(b.end(), v->begin(), v->end());
    for (auto &t : th) t.join();
    write_bin(out_pfb, a);
    write_bin(out_ovs, b);
    return 0;
}

// BASELINE.json configs[3] driven the way the reference's host works (src/ratpak.rs:60-185: one OS thread per block, all in ONE
// process): the stream is time-sharded over every visible GPU, one channelizer thread per GPU writes the per-destination layout,
// and the main thread regroups with redio_pfb_exchange_all (RCCL send/recv in one group) -- no Python, no torch.
// Output file: for every device g, [all rows in time order][channels of g].
static int dev_c4_sharded(const char *in, const char *out, int want_dev)
{
    using cf = std::complex<float>;
    auto x = read_bin<cf>(in);
    const int M = 64, P = 16;
    int ndev = 0;
    dev::check(redio_device_count(&ndev));
    if (want_dev > 0 && want_dev < ndev) ndev = want_dev;
    while (M % ndev) --ndev;
    const size_t total_rows = x.size() / M, nout = total_rows - P + 1, cpg = (size_t)M / ndev;
    const std::vector<float> proto = dsputils::lpf_corrected((size_t)M * P, 0.45f / M);
    std::vector<redio_comm *> comms((size_t)ndev, nullptr);
    dev::check(redio_comm_init_all(comms.data(), ndev, nullptr));
    std::vector<size_t> first((size_t)ndev), rows((size_t)ndev);
    for (int g = 0; g < ndev; ++g) { // contiguous output rows, the remainder to the lowest ranks (sharding.channelizer_time_shard)
        const size_t base = nout / ndev, extra = nout % ndev;
        rows[(size_t)g] = base + ((size_t)g < extra ? 1 : 0);
        first[(size_t)g] = (size_t)g * base + ((size_t)g < extra ? (size_t)g : extra);
    }
    std::vector<void *> d_grouped((size_t)ndev, nullptr), d_out((size_t)ndev, nullptr), streams((size_t)ndev, nullptr);
    std::vector<std::thread> th;
    std::atomic<int> failed{0};
    for (int g = 0; g < ndev; ++g)
        th.push_back(spawn([&, g]() { // the channelizer block of GPU g
            try {
                dev::check(redio_set_device(g));
                redio_pfb *h = nullptr;
                dev::check(redio_pfb_create(&h, proto.data(), M, P, REDIO_FIR_FUSED));
                const size_t nin = (rows[(size_t)g] + P - 1) * M;
                void *d_in = nullptr;
                dev::check(redio_malloc(&d_in, nin * sizeof(cf)));
                dev::check(redio_malloc(&d_grouped[(size_t)g], rows[(size_t)g] * M * sizeof(cf)));
                dev::check(redio_malloc(&d_out[(size_t)g], nout * cpg * sizeof(cf)));
                dev::check(redio_stream_create(&streams[(size_t)g]));
                dev::check(redio_upload(d_in, x.data() + first[(size_t)g] * M, nin * sizeof(cf), streams[(size_t)g]));
                dev::check(redio_pfb_enqueue(h, d_in, nin, d_grouped[(size_t)g], ndev, streams[(size_t)g]));
                dev::check(redio_stream_sync(streams[(size_t)g]));
                redio_free(d_in);
                redio_pfb_destroy(h);
            } catch (...) { failed = 1; throw; }
        }));
    for (auto &t : th) t.join();
    if (failed) return 4;
    dev::check(redio_pfb_exchange_all(comms.data(), ndev, d_grouped.data(), d_out.data(), rows.data(), cpg, streams.data()));
    std::vector<cf> all;
    for (int g = 0; g < ndev; ++g) {
        dev::check(redio_set_device(g));
        dev::check(redio_stream_sync(streams[(size_t)g]));
        std::vector<cf> mine(nout * cpg);
        dev::check(redio_download(mine.data(), d_out[(size_t)g], mine.size() * sizeof(cf), nullptr));
        dev::check(redio_stream_sync(nullptr));
        all.insert(all.end(), mine.begin(), mine.end());
        redio_free(d_grouped[(size_t)g]); redio_free(d_out[(size_t)g]); redio_stream_destroy(streams[(size_t)g]);
        redio_comm_destroy(comms[(size_t)g]);
    }
    write_bin(out, all);
    std::printf("devices %d\n", ndev);
    return 0;
}

// a stream cut into messages of awkward, varying lengths through the carried-history blocks: the concatenated outputs are those
// of ONE stateless call on the whole stream
static int dev_stream_graph(const char *in, const char *out_chain, const char *out_fir, const char *out_ovs, size_t seed)
{
    using cf = std::complex<float>;
    auto x = read_bin<cf>(in);
    const std::vector<float> taps = dsputils::lpf_corrected(127, 0.08f);
    auto [s0, r0] = channel<std::vector<cf>>();
    auto [s1, r1] = channel<dev::View<cf>>();
    std::vector<Sender<dev::View<cf>>> outs;
    std::vector<Receiver<dev::View<cf>>> ins;
    for (int i = 0; i < 3; ++i) { auto [a, b] = channel<dev::View<cf>>(); outs.push_back(std::move(a)); ins.push_back(std::move(b)); }
    auto [c0, d0] = channel<dev::View<cf>>();
    auto [c1, d1] = channel<dev::View<cf>>();
    auto [c2, d2] = channel<dev::View<cf>>();
    auto [e0, f0] = channel<std::vector<cf>>();
    auto [e1, f1] = channel<std::vector<cf>>();
    auto [e2, f2] = channel<std::vector<cf>>();
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s0), &x, seed]() mutable {
        uint64_t r = seed * 6364136223846793005ull + 1442695040888963407ull;
        for (size_t o = 0; o < x.size();) {
            r = r * 6364136223846793005ull + 1442695040888963407ull;
            const size_t pick[6] = {1, 125, 126, 5119, 20001, 70000};
            size_t m = pick[(r >> 33) % 6];
            if (o + m > x.size()) m = x.size() - o;
            s.send(std::vector<cf>(x.begin() + (long)o, x.begin() + (long)(o + m)));
            o += m;
        }
    }));
    th.push_back(spawn([r = std::move(r0), s = std::move(s1)]() mutable { dev::to_device<cf>(std::move(r), std::move(s)); }));
    th.push_back(spawn([r = std::move(r1), o = std::move(outs)]() mutable { fork<dev::View<cf>>(std::move(r), std::move(o)); }));
    th.push_back(spawn([r = std::move(ins[0]), s = std::move(c0), taps]() mutable { dev::fir_fft_chain_stream(std::move(r), std::move(s), taps, 5, 1024, true); }));
    th.push_back(spawn([r = std::move(ins[1]), s = std::move(c1), taps]() mutable { dev::fir_stream(std::move(r), std::move(s), taps, 5, false); }));
    th.push_back(spawn([r = std::move(ins[2]), s = std::move(c2), taps]() mutable { dev::overlap_save_stream(std::move(r), std::move(s), taps, 4096); }));
    th.push_back(spawn([r = std::move(d0), s = std::move(e0)]() mutable { dev::to_host<cf>(std::move(r), std::move(s)); }));
    th.push_back(spawn([r = std::move(d1), s = std::move(e1)]() mutable { dev::to_host<cf>(std::move(r), std::move(s)); }));
    th.push_back(spawn([r = std::move(d2), s = std::move(e2)]() mutable { dev::to_host<cf>(std::move(r), std::move(s)); }));
    std::vector<cf> a, b, c;
    while (auto v = f0.try_recv_blocking()) a.insert(a.end(), v->begin(), v->end());
    while (auto v = f1.try_recv_blocking()) b.insert(b.end(), v->begin(), v->end());
    while (auto v = f2.try_recv_blocking()) c.insert(c.end(), v->begin(), v->end());
    for (auto &t : th) t.join();
    write_bin(out_chain, a);
    write_bin(out_fir, b);
    write_bin(out_ovs, c);
    return 0;
}

// the receiver's byte messages (rtlsdr::rtlSource sends Vec<u8>, rtlsdr.rs:127-152) through the one-kernel bytes -> spectra block
static int dev_bytes_chain(const char *in, const char *out, size_t msg_bytes)
{
    using cf = std::complex<float>;
    auto raw = read_bin<uint8_t>(in);
    const std::vector<float> taps = dsputils::lpf_corrected(127, 0.08f);
    auto [s0, r0] = channel<std::vector<uint8_t>>();
    auto [s1, r1] = channel<dev::View<uint8_t>>();
    auto [s2, r2] = channel<dev::View<cf>>();
    auto [s3, r3] = channel<std::vector<cf>>();
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s0), &raw, msg_bytes]() mutable {
        for (size_t o = 0; o < raw.size(); o += msg_bytes)
            s.send(std::vector<uint8_t>(raw.begin() + (long)o, raw.begin() + (long)std::min(raw.size(), o + msg_bytes)));
    }));
    th.push_back(spawn([r = std::move(r0), s = std::move(s1)]() mutable { dev::to_device<uint8_t>(std::move(r), std::move(s)); }));
    th.push_back(spawn([r = std::move(r1), s = std::move(s2), taps]() mutable { dev::bytes_fir_fft_chain(std::move(r), std::move(s), taps, 5, 1024, true); }));
    th.push_back(spawn([r = std::move(r2), s = std::move(s3)]() mutable { dev::to_host<cf>(std::move(r), std::move(s)); }));
    std::vector<cf> a;
    while (auto v = r3.try_recv_blocking()) a.insert(a.end(), v->begin(), v->end());
    for (auto &t : th) t.join();
    write_bin(out, a);
    return 0;
}


// A three-block device graph on bounded rings: synth source -> fused chain -> checksum sink, nmsg messages of msg samples, ring depth
// `depth`.  A pass-through tap between chain and sink samples redio_malloc_count() after message `warm`: the graph must not allocate
// after that (SURVEY.md 8b "credit/ring").  Prints the checksum of every spectrum word (checked against the oracle by the test).
// policy 0: the blocks share the graph stream; 1: a stream per block (order made by events on demand); host_sync 1: the debugging mode.
static int dev_ring_graph(size_t msg, size_t nmsg, size_t depth, size_t warm, int policy, int host_sync)
{
    using cf = std::complex<float>;
    dev::set_default_ring_depth(depth);
    dev::set_stream_policy(policy ? dev::PER_BLOCK : dev::SHARED);
    dev::set_host_sync(host_sync != 0);
    const std::vector<float> taps = dsputils::lpf_corrected(127, 0.08f);
    auto [s1, r1] = channel<dev::View<cf>>();
    auto [s2, r2] = channel<dev::View<cf>>();
    auto [s3, r3] = channel<dev::View<cf>>();
    std::atomic<unsigned long long> at_warm{0};
    unsigned long long sum = 0;
    size_t seen = 0;
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s1), msg, nmsg]() mutable { dev::synth_iq_source(std::move(s), 0x5EED0002u, msg, nmsg); }));
    th.push_back(spawn([r = std::move(r1), s = std::move(s2), taps]() mutable { dev::fir_fft_chain(std::move(r), std::move(s), taps, 5, 1024, true); }));
    th.push_back(spawn([r = std::move(r2), s = std::move(s3), &at_warm, warm]() mutable {
        for (size_t i = 0;; ++i) {
            auto d = r.recv();
            if (i == warm) at_warm = redio_malloc_count();
            s.send_unwrap(std::move(d));
        }
    }));
    th.push_back(spawn([r = std::move(r3), &sum, &seen]() mutable { dev::checksum_sink<cf>(std::move(r), &sum, &seen); }));
    for (auto &t : th) t.join();
    std::printf("messages %zu checksum %llu mallocs_at_warm %llu mallocs_at_end %llu\n", seen, sum, at_warm.load(), redio_malloc_count());
    return seen == nmsg ? 0 : 4;
}

// ---- bench_c2: what a LibRedio graph would actually run (one thread per block, messages through channels: kpn.rs:278-291,
// kissfft.rs:18-31, ratpak.rs:60-185) against the bare plan launches the headline times ----
// source -> dev::fir_fft_chain (127 taps / 5 -> 1024-point transform) -> sink, messages of 2^log2_msg cf32 samples.
//   source "resident": views of R pre-generated messages, cycled (the bench contract: inputs resident in HBM when the timed region starts)
//          "synth":    every message generated afresh by redio_synth_iq in the source block (8 more bytes per sample through HBM)
//   sink   "checksum": redio_checksum_u32 over every spectrum word;  "drop": orders itself behind the message and drops it
//   depth: ring buffers per block (0 = one hipMalloc + hipFree per message);  host_sync 1 = hipStreamSynchronize before every send;
//   policy 0 = compute blocks share the graph stream, 1 = a stream per block (events on demand)
//          (depth 0 + host_sync 1 + policy 1 is the round-5 behaviour: the "before" line)
// bare = the same work without the graph: the same plan, the same R input and output buffers, redio_chain_enqueue (+ redio_checksum_u32
// for the checksum sink) back to back from one thread on one stream.
// Both are timed on the host clock between two completed synchronisations after >= 150 ms of warm-up work.
struct BenchC2 { double bare_us, bare_chain_only_us, graph_us; size_t used; unsigned long long mallocs; unsigned long long checksum; size_t nmsg; };
static int bench_c2_one(int log2_msg, size_t nmsg, size_t depth, bool resident, bool checksum, bool host_sync, int policy, BenchC2 *res, bool carried = false)
{
    using cf = std::complex<float>;
    using clk = std::chrono::steady_clock;
    const size_t msg = (size_t)1 << log2_msg, R = 4;
    nmsg = (nmsg + R - 1) / R * R; // whole cycles of the R resident messages: the checksum does not depend on where the warm-up ended
    const std::vector<float> taps = dsputils::lpf_corrected(127, 0.08f);
    auto big = dev::make<cf>(R * msg); // R distinct resident messages
    dev::check(redio_synth_iq(big.data(), 0x5EED0002u, 0, R * msg, nullptr));
    dev::check(redio_stream_sync(nullptr));
    // ---- bare launches ----
    redio_chain *h = nullptr;
    dev::check(redio_chain_create(&h, taps.data(), taps.size(), 5, 1024, REDIO_FIR_FUSED));
    // carried: the chain as a STREAM (redio_chain_stream_*: the unconsumed tail of every message kept on the device, SURVEY.md 8d C2 "history
    // carried"): every message of 2^k samples then yields 2^k / 5120 spectra on average instead of dropping its last 126 + samples
    redio_chain_stream *hs = nullptr;
    if (carried) dev::check(redio_chain_stream_create(&hs, h));
    const size_t nblk = carried ? (msg / 5120 + 1) : redio_chain_nblocks(h, msg), used = carried ? msg : nblk * 5120, nout = nblk * 1024;
    double bare_us = 0, bare_chain_us = 0;
    size_t warm = 8;
    {
        auto outs = dev::make<cf>(R * nout);
        auto acc = dev::make<unsigned long long>(1);
        dev::BlockStream st(dev::BlockStream::TRANSFER);
        auto burst = [&](size_t n, bool with_sink) {
            for (size_t i = 0; i < n; ++i) {
                size_t got = nout;
                if (carried) dev::check(redio_chain_stream_enqueue(hs, big.data() + (i % R) * msg, msg, outs.data() + (i % R) * nout, &got, st));
                else dev::check(redio_chain_enqueue(h, big.data() + (i % R) * msg, msg, outs.data() + (i % R) * nout, st));
                if (with_sink && got) dev::check(redio_checksum_u32(outs.data() + (i % R) * nout, got * 2, acc.data(), st));
            }
            dev::check(redio_stream_sync(st));
        };
        burst(8, checksum);
        auto t0 = clk::now();
        size_t done = 0;
        while (std::chrono::duration<double>(clk::now() - t0).count() < 0.15) { burst(32, checksum); done += 32; }
        const double per = std::chrono::duration<double>(clk::now() - t0).count() / (double)done;
        warm = std::max<size_t>(8, (size_t)(0.15 / per));
        auto t1 = clk::now();
        burst(nmsg, checksum);
        bare_us = std::chrono::duration<double>(clk::now() - t1).count() / (double)nmsg * 1e6;
        bare_chain_us = bare_us;
        if (checksum) {
            burst(std::min<size_t>(nmsg, 64), false);
            auto t2 = clk::now();
            burst(nmsg, false);
            bare_chain_us = std::chrono::duration<double>(clk::now() - t2).count() / (double)nmsg * 1e6;
        }
    }
    if (hs) redio_chain_stream_destroy(hs);
    redio_chain_destroy(h);
    // ---- the graph ----
    dev::set_default_ring_depth(depth);
    dev::set_host_sync(host_sync);
    dev::set_stream_policy(policy ? dev::PER_BLOCK : dev::SHARED);
    auto [s1, r1] = bounded_channel<dev::View<cf>>(8);
    auto [s2, r2] = channel<dev::View<cf>>();
    const size_t total = warm + nmsg;
    clk::time_point t0, t1;
    size_t carried_spectra = 0; // carried: spectra inside the timed region
    unsigned long long m0 = 0, m1 = 0, sum = 0;
    std::vector<std::thread> th;
    if (resident)
        th.push_back(spawn([s = std::move(s1), big, msg, total, R]() mutable { for (size_t i = 0; i < total; ++i) s.send_unwrap(big.sub((i % R) * msg, msg)); }));
    else
        th.push_back(spawn([s = std::move(s1), msg, total]() mutable { dev::synth_iq_source(std::move(s), 0x5EED0002u, msg, total); }));
    if (carried) th.push_back(spawn([r = std::move(r1), s = std::move(s2), taps]() mutable { dev::fir_fft_chain_stream(std::move(r), std::move(s), taps, 5, 1024, true); }));
    else th.push_back(spawn([r = std::move(r1), s = std::move(s2), taps]() mutable { dev::fir_fft_chain(std::move(r), std::move(s), taps, 5, 1024, true); }));
    th.push_back(spawn([&, r = std::move(r2)]() mutable {
        dev::BlockStream st;
        auto acc = dev::make<unsigned long long>(1);
        const unsigned long long zero = 0;
        dev::check(redio_upload(acc.data(), &zero, 8, st));
        dev::check(redio_stream_sync(st));
        if (carried) { // the stream block sends only when spectra complete, so the sink counts SPECTRA: the timed region starts when those of the first
                       // `warm` messages have arrived and been waited for (the host threads run far ahead of the GPU: a host-side start time would not do)
            const auto spectra_of = [](size_t n) { return n < 5246 ? (size_t)0 : ((n - 127) / 5 + 1) / 1024; };
            const size_t s_warm = spectra_of(warm * msg), s_total = spectra_of(total * msg);
            size_t seen = 0;
            bool started = false;
            try {
                for (;;) {
                    {
                        auto d = r.recv();
                        dev::Reading<cf> in(d, st);
                        if (checksum && started) dev::check(redio_checksum_u32(d.data(), d.len * 2, acc.data(), st));
                        seen += d.len / 1024;
                    }
                    if (!started && seen >= s_warm) { dev::check(redio_stream_sync(st)); m0 = redio_malloc_count(); t0 = clk::now(); started = true; carried_spectra = s_total - seen; }
                }
            } catch (const hangup &) {
            }
        } else
        for (size_t i = 0; i < total; ++i) {
            auto d = r.recv();
            {
                dev::Reading<cf> in(d, st);
                if (checksum && i >= warm) dev::check(redio_checksum_u32(d.data(), d.len * 2, acc.data(), st));
            }
            d = dev::View<cf>(); // the clock's synchronisation below is the bench's, not the graph's: nothing of the graph is held across it
            if (i + 1 == warm) { dev::check(redio_stream_sync(st)); m0 = redio_malloc_count(); t0 = clk::now(); }
        }
        dev::check(redio_stream_sync(st));
        t1 = clk::now();
        m1 = redio_malloc_count();
        dev::check(redio_download(&sum, acc.data(), 8, st));
        dev::check(redio_stream_sync(st));
    }));
    for (auto &t : th) t.join();
    dev::set_host_sync(false);
    dev::set_default_ring_depth(4);
    dev::set_stream_policy(dev::SHARED);
    res->bare_us = bare_us;
    res->bare_chain_only_us = bare_chain_us;
    res->graph_us = std::chrono::duration<double>(t1 - t0).count() / (double)nmsg * 1e6;
    if (carried) res->graph_us *= (double)nmsg * (double)msg / ((double)carried_spectra * 5120.0); // per message's worth of samples actually inside the timed region
    res->used = used; res->mallocs = m1 - m0; res->checksum = sum; res->nmsg = nmsg;
    return 0;
}

static void bench_c2_print(int log2_msg, size_t depth, bool resident, bool checksum, bool host_sync, int policy, const BenchC2 &r, bool carried = false)
{
    std::printf("{\"mode\": \"bench_c2\", \"log2_msg\": %d, \"used_samples_per_msg\": %zu, \"messages\": %zu, \"ring\": %zu, \"source\": \"%s\", "
                "\"sink\": \"%s\", \"host_sync\": %d, \"streams\": \"%s\", \"history\": \"%s\", \"bare_us_per_msg\": %.3f, \"bare_gsps\": %.3f, \"bare_chain_only_us_per_msg\": %.3f, "
                "\"graph_us_per_msg\": %.3f, \"graph_gsps\": %.3f, \"frac_of_bare\": %.4f, \"frac_of_bare_chain_only\": %.4f, \"mallocs_in_timed_region\": %llu, "
                "\"checksum\": %llu}\n",
                log2_msg, r.used, r.nmsg, depth, resident ? "resident" : "synth", checksum ? "checksum" : "drop", (int)host_sync,
                policy ? "per_block" : "shared", carried ? "carried" : "per_message", r.bare_us, (double)r.used / r.bare_us * 1e-3, r.bare_chain_only_us, r.graph_us,
                (double)r.used / r.graph_us * 1e-3, r.bare_us / r.graph_us, r.bare_chain_only_us / r.graph_us, r.mallocs, r.checksum);
    std::fflush(stdout);
}

// source "carried" = resident messages through dev::fir_fft_chain_stream (the chain as a stream: history carried across messages)
static int bench_c2(int log2_msg, size_t nmsg, size_t depth, const std::string &source, const std::string &sink, int host_sync, int policy)
{
    BenchC2 r{};
    const bool carried = source == "carried";
    if (int rc = bench_c2_one(log2_msg, nmsg, depth, source != "synth", sink == "checksum", host_sync != 0, policy, &r, carried)) return rc;
    bench_c2_print(log2_msg, depth, source != "synth", sink == "checksum", host_sync != 0, policy, r, carried);
    return 0;
}

// message sizes 2^lo ... 2^hi (step 4x), each about `seconds_per_point` of timed work: the graph as shipped (rings of 4, shared graph
// stream), with before & 1 the round-5 behaviour beside it (no pool, host sync, a stream per block), with before & 2 the per-block-stream
// policy (rings, events on demand)
static int bench_c2_sweep(double seconds_per_point, int before, int lo, int hi)
{
    for (int k = lo; k <= hi; k += (k < 16 ? 16 - k : 2)) {
        // messages per point from the kernel's rate (~0.5 ms per 2^28 samples) with a floor for the launch-bound sizes
        const double est_us = std::max(12.0, 512.0 * std::ldexp(1.0, k - 28));
        const size_t nmsg = std::max<size_t>(20, (size_t)(seconds_per_point * 1e6 / est_us));
        BenchC2 r{};
        if (int rc = bench_c2_one(k, nmsg, 4, true, true, false, 0, &r)) return rc;
        bench_c2_print(k, 4, true, true, false, 0, r);
        if (before & 2) {
            if (int rc = bench_c2_one(k, nmsg, 4, true, true, false, 1, &r)) return rc;
            bench_c2_print(k, 4, true, true, false, 1, r);
        }
        if (before & 1) {
            if (int rc = bench_c2_one(k, std::max<size_t>(20, nmsg / 4), 0, true, true, true, 1, &r)) return rc;
            bench_c2_print(k, 0, true, true, true, 1, r);
        }
    }
    return 0;
}

// one process, many points: each spec is log2_msg:nmsg:depth:host_sync:policy:source:sink (bench.py's kpn_graph_c2 leg)
static int bench_c2_list(int nspec, char **specs)
{
    for (int i = 0; i < nspec; ++i) {
        int k = 0, sync = 0, policy = 0;
        size_t nmsg = 0, depth = 0;
        char src[16] = {0}, snk[16] = {0};
        if (std::sscanf(specs[i], "%d:%zu:%zu:%d:%d:%15[a-z]:%15[a-z]", &k, &nmsg, &depth, &sync, &policy, src, snk) != 7) return 2;
        if (int rc = bench_c2(k, nmsg, depth, src, snk, sync, policy)) return rc;
    }
    return 0;
}

// ---- bench_block: the other hot blocks behind the operator API, same method as bench_c2 (source of 4 resident messages -> block -> checksum sink, one thread
// per block, against the same launches made bare from one thread on one stream):
//   channelizer  dev::channelizer, 64 channels x 16 taps per branch (BASELINE.json configs[3])        16 B per sample
//   ovsave       dev::overlap_save, 65536-point blocks, 8193 taps (configs[4])                        message = 65536 + k * 57344 samples
//   fft          dev::fft, 1024-point forward transforms (kissfft::fft, kissfft.rs:18-31)
//   fir          dev::fir, 127 taps, decimate by 5 (dsputils::convolve's decimating form)
// One JSON line; GS/s of input samples.
static int bench_block(const std::string &kind, int log2_msg, size_t nmsg)
{
    using cf = std::complex<float>;
    using clk = std::chrono::steady_clock;
    const size_t R = 4;
    nmsg = (nmsg + R - 1) / R * R;
    size_t msg = log2_msg > 40 ? (size_t)log2_msg / 1024 * 1024 : (size_t)1 << log2_msg; // above 40: a sample count (whole 1024-point blocks)
    if (kind == "ovsave") msg = 65536 + ((msg - 65536) / 57344) * 57344; // whole blocks: nothing of a message is dropped
    auto big = dev::make<cf>(R * msg);
    dev::check(redio_synth_iq(big.data(), 0x5EED0004u, 0, R * msg, nullptr));
    dev::check(redio_stream_sync(nullptr));
    const std::vector<float> taps127 = dsputils::lpf_corrected(127, 0.08f), proto = dsputils::lpf_corrected(64 * 16, 0.45f / 64.0f), taps8k = dsputils::lpf_corrected(8193, 0.08f);
    redio_pfb *hp = nullptr; redio_ovsave *ho = nullptr; redio_fft *hf = nullptr; redio_fir *hr = nullptr;
    size_t nout = 0;
    if (kind == "channelizer") { dev::check(redio_pfb_create(&hp, proto.data(), 64, 16, REDIO_FIR_FUSED)); nout = redio_pfb_nrows(hp, msg) * 64; }
    else if (kind == "ovsave") { dev::check(redio_ovsave_create(&ho, taps8k.data(), taps8k.size(), 65536)); nout = redio_ovsave_nout(ho, msg); }
    else if (kind == "fft") { dev::check(redio_fft_create(&hf, 1024, 0)); nout = msg; }
    else if (kind == "fir") { dev::check(redio_fir_create(&hr, taps127.data(), taps127.size(), 5, REDIO_FIR_COMPLEX | REDIO_FIR_FUSED)); nout = redio_fir_nout(hr, msg); }
    else return 2;
    auto enqueue = [&](const cf *in, cf *out, void *st) {
        if (hp) return redio_pfb_enqueue(hp, in, msg, out, 1, st);
        if (ho) return redio_ovsave_enqueue(ho, in, msg, out, st);
        if (hf) return redio_fft_enqueue(hf, in, out, msg / 1024, st);
        return redio_fir_enqueue(hr, in, msg, out, st);
    };
    double bare_us = 0;
    size_t warm = 8;
    {
        // KPN_BENCH_SEPARATE_OUTS=1: the bare leg's outputs as R allocations instead of one (what the device's page mapping of a buffer is worth)
        const bool separate = std::getenv("KPN_BENCH_SEPARATE_OUTS") != nullptr;
        auto outs = dev::make<cf>(separate ? 1 : R * nout);
        std::vector<dev::View<cf>> each;
        for (size_t i = 0; i < R; ++i) each.push_back(separate ? dev::make<cf>(nout + nout / 8) : outs.sub(i * nout, nout));
        auto acc = dev::make<unsigned long long>(1);
        dev::BlockStream st(dev::BlockStream::TRANSFER);
        auto burst = [&](size_t n) {
            for (size_t i = 0; i < n; ++i) {
                dev::check(enqueue(big.data() + (i % R) * msg, each[i % R].data(), st));
                dev::check(redio_checksum_u32(each[i % R].data(), nout * 2, acc.data(), st));
            }
            dev::check(redio_stream_sync(st));
        };
        burst(4);
        auto t0 = clk::now();
        size_t done = 0;
        while (std::chrono::duration<double>(clk::now() - t0).count() < 0.15) { burst(8); done += 8; }
        warm = std::max<size_t>(8, (size_t)(0.15 / (std::chrono::duration<double>(clk::now() - t0).count() / (double)done)));
        auto t1 = clk::now();
        burst(nmsg);
        bare_us = std::chrono::duration<double>(clk::now() - t1).count() / (double)nmsg * 1e6;
    }
    if (hp) redio_pfb_destroy(hp);
    if (ho) redio_ovsave_destroy(ho);
    if (hf) redio_fft_destroy(hf);
    if (hr) redio_fir_destroy(hr);
    auto [s1, r1] = bounded_channel<dev::View<cf>>(8);
    auto [s2, r2] = channel<dev::View<cf>>();
    const size_t total = warm + nmsg;
    clk::time_point t0, t1;
    unsigned long long m0 = 0, m1 = 0, sum = 0;
    std::vector<std::thread> th;
    th.push_back(spawn([s = std::move(s1), big, msg, total, R]() mutable { for (size_t i = 0; i < total; ++i) s.send_unwrap(big.sub((i % R) * msg, msg)); }));
    if (kind == "channelizer") th.push_back(spawn([r = std::move(r1), s = std::move(s2), proto]() mutable { dev::channelizer(std::move(r), std::move(s), proto, 64, 16, true); }));
    else if (kind == "ovsave") th.push_back(spawn([r = std::move(r1), s = std::move(s2), taps8k]() mutable { dev::overlap_save(std::move(r), std::move(s), taps8k, 65536); }));
    else if (kind == "fft") th.push_back(spawn([r = std::move(r1), s = std::move(s2)]() mutable { dev::fft(std::move(r), std::move(s), 1024, 0); }));
    else th.push_back(spawn([r = std::move(r1), s = std::move(s2), taps127]() mutable { dev::fir(std::move(r), std::move(s), taps127, 5, true); }));
    th.push_back(spawn([&, r = std::move(r2)]() mutable {
        dev::BlockStream st;
        auto acc = dev::make<unsigned long long>(1);
        const unsigned long long zero = 0;
        dev::check(redio_upload(acc.data(), &zero, 8, st));
        dev::check(redio_stream_sync(st));
        for (size_t i = 0; i < total; ++i) {
            auto d = r.recv();
            {
                dev::Reading<cf> in(d, st);
                if (i >= warm) dev::check(redio_checksum_u32(d.data(), d.len * 2, acc.data(), st));
            }
            d = dev::View<cf>(); // the clock's synchronisation below is the bench's, not the graph's: nothing of the graph is held across it
            if (i + 1 == warm) { dev::check(redio_stream_sync(st)); m0 = redio_malloc_count(); t0 = clk::now(); }
        }
        dev::check(redio_stream_sync(st));
        t1 = clk::now();
        m1 = redio_malloc_count();
        dev::check(redio_download(&sum, acc.data(), 8, st));
        dev::check(redio_stream_sync(st));
    }));
    for (auto &t : th) t.join();
    const double graph_us = std::chrono::duration<double>(t1 - t0).count() / (double)nmsg * 1e6;
    std::printf("{\"mode\": \"bench_block\", \"block\": \"%s\", \"msg_samples\": %zu, \"messages\": %zu, \"bare_us_per_msg\": %.3f, \"bare_gsps\": %.3f, \"graph_us_per_msg\": %.3f, "
                "\"graph_gsps\": %.3f, \"frac_of_bare\": %.4f, \"mallocs_in_timed_region\": %llu, \"budget_yields\": %llu, \"checksum\": %llu}\n",
                kind.c_str(), msg, nmsg, bare_us, (double)msg / bare_us * 1e-3, graph_us, (double)msg / graph_us * 1e-3, bare_us / graph_us, m1 - m0,
                (unsigned long long)dev::budget_yields_ref().load(), sum);
    std::fflush(stdout);
    return 0;
}
// kind:log2_msg:nmsg ...
static int bench_block_list(int nspec, char **specs)
{
    for (int i = 0; i < nspec; ++i) {
        char kind[24] = {0};
        int k = 0;
        size_t n = 0;
        if (std::sscanf(specs[i], "%23[a-z]:%d:%zu", kind, &k, &n) != 3) return 2;
        if (int rc = bench_block(kind, k, n)) return rc;
    }
    return 0;
}

int main(int argc, char **argv)
{
    try {
        std::string mode = argc > 1 ? argv[1] : "plumbing";
        // the device graphs of this driver under the other stream policy / ring depth (tests/test_kpn_cpp.py runs them both ways)
        if (const char *e = std::getenv("KPN_DEV_STREAMS")) dev::set_stream_policy(std::string(e) == "per_block" ? dev::PER_BLOCK : dev::SHARED);
        if (const char *e = std::getenv("KPN_DEV_RING")) dev::set_default_ring_depth((size_t)std::atol(e));
        if (const char *e = std::getenv("KPN_DEV_RING_PATIENCE_MS")) dev::set_ring_patience_ms(std::atoi(e));
        if (const char *e = std::getenv("KPN_DEV_RING_MIB")) dev::set_default_ring_bytes((size_t)std::atol(e) << 20); // 0: one message out at a time
        if (mode == "plumbing") return plumbing();
        if (mode == "c1" && argc == 4) return c1(argv[2], argv[3]);
        if (mode == "fft" && argc == 6) return fft_graph(argv[2], argv[3], (uint32_t)std::atoi(argv[4]), (uint32_t)std::atoi(argv[5]));
        if (mode == "devchain" && argc == 6) return dev_chain_graph(argv[2], argv[3], argv[4], (size_t)std::atol(argv[5]));
        if (mode == "devstream" && argc == 7) return dev_stream_graph(argv[2], argv[3], argv[4], argv[5], (size_t)std::atol(argv[6]));
        if (mode == "devbytes" && argc == 5) return dev_bytes_chain(argv[2], argv[3], (size_t)std::atol(argv[4]));
        if (mode == "devc4" && argc == 5) return dev_c4_sharded(argv[2], argv[3], std::atoi(argv[4]));
        if (mode == "devshaper" && argc == 6) return dev_shaper_graph(argv[2], argv[3], (size_t)std::atol(argv[4]), (size_t)std::atol(argv[5]));
        if (mode == "devmix" && argc == 6) return dev_mix_graph(argv[2], argv[3], (size_t)std::atol(argv[4]), std::atof(argv[5]));
        if (mode == "devbank" && argc == 6) return dev_bank_graph(argv[2], argv[3], argv[4], (size_t)std::atol(argv[5]));
        if (mode == "devring" && argc == 8)
            return dev_ring_graph((size_t)std::atol(argv[2]), (size_t)std::atol(argv[3]), (size_t)std::atol(argv[4]), (size_t)std::atol(argv[5]), std::atoi(argv[6]), std::atoi(argv[7]));
        if (mode == "bench_c2" && argc == 9)
            return bench_c2(std::atoi(argv[2]), (size_t)std::atol(argv[3]), (size_t)std::atol(argv[4]), argv[5], argv[6], std::atoi(argv[7]), std::atoi(argv[8]));
        if (mode == "bench_block_list" && argc > 2) return bench_block_list(argc - 2, argv + 2);
        if (mode == "bench_c2_list" && argc > 2) return bench_c2_list(argc - 2, argv + 2);
        if (mode == "bench_c2_sweep" && argc == 6) return bench_c2_sweep(std::atof(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), std::atoi(argv[5]));
        if (mode == "resample" && argc == 6) return resample_graph(argv[2], argv[3], std::atof(argv[4]), (size_t)std::atol(argv[5]));
        std::fprintf(stderr, "usage: see the header of kpn_tests.cpp\n");
        return 2;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "kpn_tests: %s\n", e.what());
        return 3;
    }
}
