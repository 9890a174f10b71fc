/*
 * redio_oracle.h -- CPU restatement of the LibRedio hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker.  Nothing under libredio_amd/ links, imports or calls it.
 *
 * PARITY STATUS: "parity unpinned".  The reference (ade-ma/LibRedio) ships no tests, golden
 * vectors or fixtures for this path, cannot be compiled here (2015 nightly Rust, no rustc), and two
 * of its three hot blocks do their arithmetic in libraries that are absent from the reference tree:
 *   - kissfft   (git submodule https://github.com/itdaniher/kissfft/, .gitmodules:1-3, empty dir,
 *                commit unrecoverable).  Restated here from the published kissfft 1.3.0
 *                kiss_fft.c algorithm (kf_factor / kf_work / kf_bfly{2,3,4,5,generic}).
 *   - libsamplerate (system -lsamplerate, src/samplerate/src/samplerate.rs:32, version unpinned).
 *                Restated from the published libsamplerate 0.1.8 src_sinc.c mono algorithm; its
 *                coefficient tables cannot be reproduced, so the table is a stated
 *                Kaiser-windowed sinc of the same shape (see orc_src_*).  Converters 3 / 4 follow src_zoh.c / src_linear.c
 *                (interleaved channels, the library's own sample-unit arithmetic); the one place where the published code reads
 *                before its input array (a single-frame message, in_used == 0 in the main loop) is DEFINED here as the value
 *                carried from the previous message.  Multi-channel sinc conversion is defined as one mono conversion per channel.
 * The oracle is pinned instead by (i) float64 numpy/scipy cross-checks and known-answer tests in
 * tests/test_oracle_*.py and (ii) the committed fixtures in tests/golden/.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef REDIO_ORACLE_H
#define REDIO_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float r, i; } orc_cpx; /* num::complex::Complex<f32>, src/kissfft/src/kissfft.rs:14 */

/* ---- A1: dsputils::convolve, src/dsputils/src/dsputils.rs:30-32 ---- */
/* valid-mode cross-correlation, strict left fold from 0.0, separate rounded mul and add.
 * returns number of outputs (nu-nv+1, or 0 when nu<nv); nv==0 returns (size_t)-1 (Rust panics). */
size_t orc_convolve_f32(const float *u, size_t nu, const float *v, size_t nv, float *out);
size_t orc_convolve_f64(const double *u, size_t nu, const double *v, size_t nv, double *out);
/* The same fold applied per component to interleaved cf32 with real taps, keeping only out[decim*i]
 * (decimation and complex input are new compositions).  fused!=0 replaces
 * (a + x*h) by fmaf(x, h, a) -- the documented fast-mode deviation. */
size_t orc_fir_c32(const orc_cpx *x, size_t n, const float *taps, size_t k, size_t decim,
                   int fused, orc_cpx *out);
size_t orc_fir_f32(const float *x, size_t n, const float *taps, size_t k, size_t decim,
                   int fused, float *out);

/* ---- A2-A4: tap generators, src/dsputils/src/dsputils.rs:38-94 (quirk-faithful) ---- */
void orc_window(size_t m, float *out /* m+1 values */);          /* :38-51 */
int  orc_sinc(size_t m, float fc, float *out /* m values */);     /* :53-63, -1 if fc>=0.5 */
int  orc_lpf(size_t m, float fc, float *out);                     /* :66-71 */
int  orc_hpf(size_t m, float fc, float *out);                     /* :74-79, -1 if m<2 */
int  orc_bsf(size_t m, float fc1, float fc2, float *out);         /* :82-88 */
int  orc_bpf(size_t m, float fc1, float fc2, float *out);         /* :91-94 */
/* corrected Blackman-Nuttall windowed sinc (documented deviation) */
int  orc_lpf_corrected(size_t m, float fc, float *out);

/* ---- A5/A5x: kissfft (published kissfft 1.3.0 kiss_fft.c restated) ---- */
typedef struct orc_kiss_state orc_kiss_state;
orc_kiss_state *orc_kiss_fft_alloc(int nfft, int inverse);
void orc_kiss_fft(const orc_kiss_state *st, const orc_cpx *fin, orc_cpx *fout);
void orc_kiss_fft_free(orc_kiss_state *st);
int  orc_kiss_factors(int nfft, int *facbuf /* 2*32 ints */);     /* kf_factor; returns #stages */
/* block contract of kissfft::fft, src/kissfft/src/kissfft.rs:18-31: nblocks transforms of exactly nfft */
void orc_fft_blocks(int nfft, int inverse, const orc_cpx *in, orc_cpx *out, size_t nblocks);

/* ---- synthetic IQ, SURVEY.md 8d: murmur3 fmix32 hash -> [-1,1) ---- */
uint32_t orc_hash32(uint32_t seed, uint64_t index);
void orc_synth_iq(uint32_t seed, uint64_t first_sample, size_t n, orc_cpx *out);
void orc_synth_f32(uint32_t seed, uint64_t first_sample, size_t n, float *out);

/* ---- C2 chain: FIR(K) decimate-by-D (valid, whole buffer) -> nfft-point forward FFT blocks ---- */
/* returns number of spectra written; scratch is managed internally. */
size_t orc_chain_fir_fft(const orc_cpx *x, size_t n, const float *taps, size_t k, size_t decim,
                         int nfft, int fused, orc_cpx *out);

/* ---- C4: M-channel polyphase channelizer = per-branch convolve + per-row kiss_fft (new composition) ---- */
size_t orc_pfb_channelizer(const orc_cpx *x, size_t n, const float *h, int M, int P, int fused, orc_cpx *out);

/* ---- C5: overlap-save FFT convolution = convolve semantics through kiss_fft blocks (new composition) ---- */
size_t orc_overlap_save(const orc_cpx *x, size_t n, const float *h, size_t k, int nfft, orc_cpx *out);

/* ---- A6/A6x: samplerate::resample, src/samplerate/src/samplerate.rs:59-87 + src_sinc.c ---- */
typedef struct {
    const float *data_in; float *data_out;
    long input_frames, output_frames, input_frames_used, output_frames_gen;
    int end_of_input; double src_ratio;
} orc_src_data; /* SRC_DATA, src/samplerate/src/samplerate.rs:15-24 (C layout) */
typedef struct orc_src_state orc_src_state;
orc_src_state *orc_src_new(int converter_type, int channels, int *error); /* :61 */
void orc_src_delete(orc_src_state *s);
int  orc_src_process(orc_src_state *s, orc_src_data *d);                  /* :76 */
int  orc_src_reset(orc_src_state *s);
int  orc_src_set_ratio(orc_src_state *s, double new_ratio);              /* :40 */
/* table access so the GPU library and the oracle can be compared coefficient by coefficient */
int  orc_src_table(int converter_type, const float **coeffs, int *half_len, int *increment);
/* one message of the resample block: lout = (ratio*len + 1) as usize (:64); returns frames generated */
long orc_resample_block(orc_src_state *s, const float *in, long len, double ratio, float *out, long cap);

/* ---- A9: bit-exact paths ---- */
size_t orc_b2d(const size_t *bits, size_t n);                              /* src/kpn/src/kpn.rs:111-113 */
int    orc_eat(const size_t *bits, size_t nbits, const size_t *widths, size_t nw, size_t *out); /* :116-124 */
void   orc_discretize(const float *x, size_t n, size_t *out);              /* src/bitfount/src/bitfount.rs:87-96 */
int    orc_data_to_samples(const uint8_t *d, size_t n, orc_cpx *out);      /* src/rtlsdr/src/rtlsdr.rs:159-162 */
/* trigger state machine, src/bitfount/src/bitfount.rs:36-85: feed nblocks blocks of `block` floats;
 * emitted buffers are concatenated into out with their lengths in out_lens; returns #emitted */
typedef struct orc_trigger_state orc_trigger_state;
orc_trigger_state *orc_trigger_new(void);
void   orc_trigger_free(orc_trigger_state *t);
size_t orc_trigger_feed(orc_trigger_state *t, const float *blocks, size_t nblocks, size_t block,
                        float *out, size_t out_cap, size_t *out_lens, size_t lens_cap, size_t *out_total);
float  orc_block_sum(const float *x, size_t n); /* sequential f32 sum, bitfount.rs:48 */
void   orc_norm_c32(const orc_cpx *x, size_t n, float *out); /* Complex::norm = hypotf, src/ratpak.rs:64-68 */
void   orc_zip_f32(const float *a, const float *b, size_t n, int add, float *out);       /* mul_vecs / sum_vecs, src/kpn/src/kpn.rs:198-203, 227-231 */
void   orc_zip_c32(const orc_cpx *a, const orc_cpx *b, size_t n, int add, orc_cpx *out);

#ifdef __cplusplus
}
#endif
#endif
