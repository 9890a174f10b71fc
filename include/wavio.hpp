// wavio.hpp -- C++ twin of LibRedio's wavio sources (src/wavio/src/wavio.rs) plus the sink the
// reference lacks.  The reference reads through libsndfile; this is a minimal RIFF/WAVE reader and
// writer (PCM16 and IEEE float32, any channel count) with the same block contract:
//   wav_source_f32(u, fname, s_rate)          wavio.rs:12-28  mono, one f32 per message
//   wav_source_complex_f32(u, fname, s_rate)  wavio.rs:30-46  stereo-as-IQ, one Complex<f32> per message
// Both assert the sample rate and channel count (:15-16, :33-34) and read 1024-item chunks (:20, :38).
// Two behaviours of the reference are NOT reproduced by default and are documented instead:
//   * it reads only (frames/2)/1024 chunks (:19, :37), i.e. about half (mono) or a quarter (IQ) of
//     the file; pass reference_chunk_count = true to get exactly that;
//   * it then parks forever to keep its Sender alive (:25-27, :43-45); here the source returns, its
//     Sender drops and the hang-up flows downstream as a clean end of stream.
#pragma once
#include "kpn.hpp"
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace wavio {

struct WavInfo {
    uint32_t samplerate = 0;
    uint16_t channels = 0;
    uint16_t format = 0; // 1 = PCM, 3 = IEEE float
    uint16_t bits = 0;
    uint64_t frames = 0;
};

// whole file -> interleaved f32 (PCM16 is scaled by 1/32768 as libsndfile's read_f32 does).  A header this reader cannot serve --
// not RIFF/WAVE, a format chunk shorter than its 16 fixed bytes or absurdly long, no channels, a sample size other than the two it
// converts, data before the format -- throws std::runtime_error; a data chunk that claims more bytes than the file holds (streaming
// writers leave 0xFFFFFFFF there) yields the whole frames that are present, as libsndfile does.  Nothing is read or allocated from an
// unchecked header field (tests/cpp/kpn_tests.cpp "wavbad", under AddressSanitizer in tests/san_check.sh).
inline std::vector<float> read_wav(const std::string &fname, WavInfo &info)
{
    struct Closer { FILE *f; ~Closer() { if (f) std::fclose(f); } } file{std::fopen(fname.c_str(), "rb")};
    FILE *f = file.f;
    if (!f) throw std::runtime_error("wavio: cannot open " + fname);
    auto rd = [&](void *p, size_t n) { if (std::fread(p, 1, n, f) != n) throw std::runtime_error("wavio: short read"); };
    char id[4]; uint32_t sz;
    rd(id, 4); rd(&sz, 4);
    if (std::memcmp(id, "RIFF", 4)) throw std::runtime_error("wavio: not RIFF");
    rd(id, 4);
    if (std::memcmp(id, "WAVE", 4)) throw std::runtime_error("wavio: not WAVE");
    std::vector<float> out;
    bool have_fmt = false;
    for (;;) {
        if (std::fread(id, 1, 4, f) != 4) break;
        rd(&sz, 4);
        if (!std::memcmp(id, "fmt ", 4)) {
            if (sz < 16 || sz > 4096) throw std::runtime_error("wavio: format chunk of " + std::to_string(sz) + " bytes");
            std::vector<uint8_t> b(sz);
            rd(b.data(), sz);
            if (sz & 1) std::fseek(f, 1, SEEK_CUR); // chunks are word aligned
            std::memcpy(&info.format, &b[0], 2); std::memcpy(&info.channels, &b[2], 2);
            std::memcpy(&info.samplerate, &b[4], 4); std::memcpy(&info.bits, &b[14], 2);
            if (info.format == 0xFFFE && sz >= 26) std::memcpy(&info.format, &b[24], 2); // WAVE_FORMAT_EXTENSIBLE
            if (info.channels == 0) throw std::runtime_error("wavio: no channels");
            have_fmt = true;
        } else if (!std::memcmp(id, "data", 4)) {
            if (!have_fmt) throw std::runtime_error("wavio: data before fmt");
            const bool f32 = info.format == 3 && info.bits == 32, pcm16 = info.format == 1 && info.bits == 16;
            if (!f32 && !pcm16) throw std::runtime_error("wavio: only PCM16 and float32 are supported");
            const size_t bps = f32 ? 4 : 2;
            // no more than the file holds, whole frames only
            const long here = std::ftell(f);
            std::fseek(f, 0, SEEK_END);
            const long fend = std::ftell(f);
            std::fseek(f, here, SEEK_SET);
            size_t bytes = sz;
            if (here >= 0 && fend >= here && (size_t)(fend - here) < bytes) bytes = (size_t)(fend - here);
            const size_t frame = bps * info.channels;
            const size_t items = (bytes / frame) * info.channels;
            out.resize(items);
            if (f32) {
                rd(out.data(), items * 4);
            } else {
                std::vector<int16_t> t(items);
                rd(t.data(), items * 2);
                for (size_t i = 0; i < items; ++i) out[i] = (float)t[i] / 32768.0f;
            }
            info.frames = items / info.channels;
            break;
        } else {
            if (std::fseek(f, (long)sz + (long)(sz & 1), SEEK_CUR) != 0) break;
        }
    }
    return out;
}

inline void write_wav_f32(const std::string &fname, const std::vector<float> &interleaved, uint32_t samplerate, uint16_t channels)
{
    FILE *f = std::fopen(fname.c_str(), "wb");
    if (!f) throw std::runtime_error("wavio: cannot create " + fname);
    const uint32_t data_bytes = (uint32_t)(interleaved.size() * 4), riff = 36 + data_bytes, fmt_sz = 16;
    const uint16_t format = 3, bits = 32, align = (uint16_t)(channels * 4);
    const uint32_t byte_rate = samplerate * align;
    std::fwrite("RIFF", 1, 4, f); std::fwrite(&riff, 4, 1, f); std::fwrite("WAVE", 1, 4, f);
    std::fwrite("fmt ", 1, 4, f); std::fwrite(&fmt_sz, 4, 1, f); std::fwrite(&format, 2, 1, f); std::fwrite(&channels, 2, 1, f);
    std::fwrite(&samplerate, 4, 1, f); std::fwrite(&byte_rate, 4, 1, f); std::fwrite(&align, 2, 1, f); std::fwrite(&bits, 2, 1, f);
    std::fwrite("data", 1, 4, f); std::fwrite(&data_bytes, 4, 1, f);
    std::fwrite(interleaved.data(), 4, interleaved.size(), f);
    std::fclose(f);
}

// wavio.rs:12-28
inline void wav_source_f32(kpn::Sender<float> u, const std::string &fname, uint32_t s_rate, bool reference_chunk_count = false)
{
    WavInfo info;
    std::vector<float> all = read_wav(fname, info);
    if (info.samplerate != s_rate) throw std::runtime_error("assert_eq!(info.samplerate, s_rate) (wavio.rs:15)");
    if (info.channels != 1) throw std::runtime_error("assert_eq!(info.channels, 1) (wavio.rs:16)");
    const uint64_t chunks = reference_chunk_count ? (info.frames / 2) / 1024 : all.size() / 1024; // :19
    for (uint64_t c = 0; c < chunks; ++c)
        for (size_t i = 0; i < 1024; ++i) u.send_unwrap(all[c * 1024 + i]); // :20-23
}

// wavio.rs:30-46
inline void wav_source_complex_f32(kpn::Sender<std::complex<float>> u, const std::string &fname, uint32_t s_rate,
                                   bool reference_chunk_count = false)
{
    WavInfo info;
    std::vector<float> all = read_wav(fname, info);
    if (info.samplerate != s_rate) throw std::runtime_error("assert_eq!(info.samplerate, s_rate) (wavio.rs:33)");
    if (info.channels != 2) throw std::runtime_error("assert_eq!(info.channels, 2) (wavio.rs:34)");
    const uint64_t chunks = reference_chunk_count ? (info.frames / 2) / 1024 : all.size() / 1024; // :37
    for (uint64_t c = 0; c < chunks; ++c)
        for (size_t i = 0; i < 1024; i += 2) u.send_unwrap({all[c * 1024 + i], all[c * 1024 + i + 1]}); // :38-41
}

// the sink the reference does not have: drain a stream of f32 into a mono float32 WAV
inline void wav_sink_f32(kpn::Receiver<float> u, const std::string &fname, uint32_t s_rate)
{
    std::vector<float> all;
    while (auto x = u.try_recv_blocking()) all.push_back(*x);
    write_wav_f32(fname, all, s_rate, 1);
}

inline void wav_sink_complex_f32(kpn::Receiver<std::complex<float>> u, const std::string &fname, uint32_t s_rate)
{
    std::vector<float> all;
    while (auto x = u.try_recv_blocking()) { all.push_back(x->real()); all.push_back(x->imag()); }
    write_wav_f32(fname, all, s_rate, 2);
}

} // namespace wavio
