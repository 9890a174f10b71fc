// kpn_tests.cpp -- exercises the C++ kpn twin (include/kpn.hpp, include/wavio.hpp).
//   kpn_tests plumbing            CPU only: channel semantics and every kpn.rs block
//   kpn_tests c1 in.wav out.wav   BASELINE.json configs[0]: WAV -> shaper(1024) -> convolve(63 taps) -> WAV
//   kpn_tests fft in.bin out.bin N inv      kissfft::fft block over raw cf32 messages of N samples
//   kpn_tests resample in.bin out.bin ratio msg_len   samplerate::resample block over raw f32 messages
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

static int plumbing()
{
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

int main(int argc, char **argv)
{
    try {
        std::string mode = argc > 1 ? argv[1] : "plumbing";
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
        if (mode == "resample" && argc == 6) return resample_graph(argv[2], argv[3], std::atof(argv[4]), (size_t)std::atol(argv[5]));
        std::fprintf(stderr, "usage: see the header of kpn_tests.cpp\n");
        return 2;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "kpn_tests: %s\n", e.what());
        return 3;
    }
}
