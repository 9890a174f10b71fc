/*
 * redio.h -- C ABI of libredio.so, the MI355X (gfx950) engine for LibRedio's per-sample DSP blocks.
 *
 * This is the drop-in boundary of SURVEY.md 8(b): plain pointers and sizes, no C++/torch types.
 * Paths below are relative to the reference tree (ade-ma/LibRedio).
 *
 * Two families of entry points share the same kernels:
 *   (1) host-buffer, synchronous calls that replace what the reference computes per message
 *       (redio_convolve_f32 for dsputils::convolve; kiss_fft.h / samplerate.h for the C symbols the
 *       reference's Rust already binds);
 *   (2) device-resident plans (redio_fir_*, redio_fft_*, redio_chain_*, ...) that take device
 *       pointers and a HIP stream and only launch kernels inside *_enqueue, so a graph of blocks can keep
 *       its streams in HBM.  Throughput numbers use (2).  Scratch: the few paths that need a plan-owned
 *       intermediate (the two-kernel chain, staged FFT sizes) size it with the plan's *_reserve(); an
 *       un-reserved plan grows it on first use (an allocation) and refuses to do so while the stream is
 *       being captured (REDIO_ERR_NOT_RESERVED).  Everything else never allocates or synchronises.
 *
 * Every function returns REDIO_OK (0) or a negative redio error / positive HIP error code mapped by
 * redio_strerror(); nothing aborts or panics (the reference's unwrap()/assert!/panic! sites become
 * error returns -- SURVEY.md 5 "failure detection").
 * Thread safety: distinct handles may be used concurrently from distinct threads; each call binds
 * the handle's device first (the reference runs one OS thread per block, src/ratpak.rs:60-185).
 */
#ifndef REDIO_H
#define REDIO_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- errors ---- */
enum {
    REDIO_OK = 0,
    REDIO_ERR_ARG = -1,         /* NULL pointer / zero taps / zero decimation / bad size */
    REDIO_ERR_NOMEM = -2,
    REDIO_ERR_UNSUPPORTED = -3, /* shape has no kernel (documented per function) */
    REDIO_ERR_NO_DEVICE = -4,   /* no HIP device: the product path never falls back to the CPU */
    REDIO_ERR_ASSERT = -5,      /* an assert!() of the reference would have fired (e.g. fc >= 0.5) */
    REDIO_ERR_NOT_RESERVED = -6, /* a plan would have to allocate scratch while its stream is being captured */
    REDIO_ERR_COMM = -7,        /* RCCL: librccl.so not loadable, or a communicator call failed */
    REDIO_ERR_HIP_BASE = -1000  /* -(1000 + hipError_t) */
};
const char *redio_strerror(int code);
const char *redio_version(void);

/* ---- device / memory / stream / event helpers (thin, so non-torch hosts can drive plans) ---- */
int redio_device_count(int *count);
int redio_set_device(int device);
int redio_get_device(int *device); /* the calling thread's current device */
int redio_malloc(void **dptr, size_t bytes);
int redio_free(void *dptr);
int redio_upload(void *dst_dev, const void *src_host, size_t bytes, void *stream);
int redio_download(void *dst_host, const void *src_dev, size_t bytes, void *stream);
int redio_copy(void *dst_dev, const void *src_dev, size_t bytes, void *stream);
/* pinned host memory that kernels can address directly (zero-copy over PCIe): *host is the CPU pointer, *dev the
 * pointer to pass as a device buffer.  For small host-resident messages this saves the two staged copies of
 * redio_upload / redio_download (the kiss_fft drop-in uses it up to 8192 points). */
int redio_host_alloc(void **host, void **dev, size_t bytes);
int redio_host_free(void *host);
int redio_stream_create(void **stream);
int redio_stream_destroy(void *stream);
int redio_stream_sync(void *stream); /* NULL = default stream */
/* writes `value` to the 32-bit word at d_word (device-addressable memory, e.g. the device alias of a redio_host_alloc buffer) when the stream
 * reaches this point: a host thread polling the word sees the work enqueued before it finished, without hipStreamSynchronize */
int redio_stream_signal(void *stream, void *d_word, uint32_t value);
/* Launch graphs for launch-bound pipelines (many small messages): every *_enqueue below only launches
 * kernels on the given stream -- no allocation, no synchronisation -- so a sequence of them can be
 * recorded once between redio_graph_begin/end on a stream created by redio_stream_create and replayed
 * with one submission.  Pointers and sizes are baked in: replay on the same buffers.  Run the sequence
 * once un-captured first: plans size their internal scratch on first use.
 * (redio_src_process synchronises and is not capturable; redio_*_create / *_destroy never are.) */
typedef struct redio_graph redio_graph;
int redio_graph_begin(void *stream);
int redio_graph_end(void *stream, redio_graph **g);
int redio_graph_launch(redio_graph *g, void *stream);
int redio_graph_destroy(redio_graph *g);
int redio_event_create(void **event);
int redio_event_destroy(void *event);
int redio_event_record(void *event, void *stream);
int redio_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on stop */
/* Ordering between the streams of different blocks without the host: an event for synchronisation only (no time stamps: cheaper to
 * record than redio_event_create's), the wait a consumer's stream performs on the event its producer recorded (work enqueued on `stream`
 * after this call runs after everything the producer's stream held when it recorded `event`; returns at once, nothing blocks the
 * host), and the host-side wait for the places that really hand data to the CPU.  include/kpn_dev.hpp carries such an event in every
 * message (the reference's channels order host Vecs, src/kpn/src/kpn.rs:11; on the device the event is that order). */
int redio_event_create_sync(void **event);
int redio_stream_wait_event(void *stream, void *event);
int redio_event_sync(void *event);
/* device allocations made through redio_malloc since the library was loaded (all threads): lets a host check that a steady-state
 * graph no longer allocates (tests/test_kpn_cpp.py: a bounded ring never allocates after warm-up) */
unsigned long long redio_malloc_count(void);

/* ---- tap generators: src/dsputils/src/dsputils.rs:38-94 (host side, run once per filter) ----
 * Quirk-faithful to the reference as written (window() returns m+1 values, lpf()[1] is NaN, ...);
 * redio_lpf_corrected is the documented-deviation designer used for benchmark taps. */
int redio_window(size_t m, float *out /* m+1 */);                 /* :38-51 */
int redio_sinc(size_t m, float fc, float *out /* m */);           /* :53-63; fc>=0.5 -> REDIO_ERR_ASSERT */
int redio_lpf(size_t m, float fc, float *out);                    /* :66-71 */
int redio_hpf(size_t m, float fc, float *out);                    /* :74-79; m<2 -> REDIO_ERR_ASSERT */
int redio_bsf(size_t m, float fc1, float fc2, float *out);        /* :82-88 */
int redio_bpf(size_t m, float fc1, float fc2, float *out);        /* :91-94 */
int redio_lpf_corrected(size_t m, float fc, float *out);

/* ---- A1: dsputils::convolve, src/dsputils/src/dsputils.rs:30-32 ----
 * Host buffers in/out, synchronous: out[i] = sum_j u[i+j]*v[j] folded left to right from 0.0 with a
 * separately rounded multiply and add (bit-exact with the reference fold).  *nout = nu-nv+1, or 0
 * when nu < nv (windows() yields nothing); nv == 0 -> REDIO_ERR_ASSERT (windows(0) panics). */
int redio_convolve_f32(const float *u, size_t nu, const float *v, size_t nv, float *out, size_t *nout);

/* ---- device-resident FIR plan (A1 extended: complex input x real taps, decimation) ---- */
enum {
    REDIO_FIR_COMPLEX = 1u, /* input/output are interleaved cf32 {re, im}; taps stay real */
    REDIO_FIR_FUSED = 2u    /* acc = fmaf(x, h, acc) instead of acc + x*h: faster, still strict order */
};
typedef struct redio_fir redio_fir;
int redio_fir_create(redio_fir **h, const float *taps_host, size_t ntaps, size_t decim, unsigned flags);
int redio_fir_destroy(redio_fir *h);
/* number of outputs for n_in inputs: 0 if n_in < ntaps, else (n_in - ntaps)/decim + 1 */
size_t redio_fir_nout(const redio_fir *h, size_t n_in);
/* valid-mode over one device buffer; d_out must hold redio_fir_nout() samples; in != out */
int redio_fir_enqueue(redio_fir *h, const void *d_in, size_t n_in, void *d_out, void *stream);

/* ---- A5: kissfft::fft block, src/kissfft/src/kissfft.rs:18-31 (device-resident, batched) ----
 * nbatch consecutive messages of exactly nfft cf32 samples; unnormalised; inverse!=0 flips the sign
 * of the exponent.  Arithmetic order of the published kissfft butterflies (bit-exact with
 * oracle/oracle_kiss.c).  d_in == d_out is allowed. */
typedef struct redio_fft redio_fft;
int redio_fft_create(redio_fft **h, int nfft, int inverse);
int redio_fft_destroy(redio_fft *h);
int redio_fft_enqueue(redio_fft *h, const void *d_in, void *d_out, size_t nbatch, void *stream);
/* staging for the sizes that need it (in-place calls of the multi-launch path, prime factors above 5 beyond LDS) */
int redio_fft_reserve(redio_fft *h, size_t nbatch);
/* messages that start every in_stride samples (overlapping blocks when in_stride < nfft); no aliasing */
int redio_fft_enqueue_strided(redio_fft *h, const void *d_in, void *d_out, size_t nbatch, long in_stride, void *stream);

/* ---- C2 chain: FIR (ntaps, decimate decim) -> nfft-point forward FFT of consecutive blocks ----
 * Fused single kernel for nfft = 1024 with (ntaps, decim) in {(127, 5), (127, 3), (127, 1), (63, 5), (63, 1)} on a
 * 16-byte aligned stream; other shapes run the FIR and FFT kernels back to back through a plan-owned
 * intermediate buffer (same results).  Trailing decimated
 * samples that do not fill a block are dropped, as kpn::shaper would (src/kpn/src/kpn.rs:278-282). */
typedef struct redio_chain redio_chain;
int redio_chain_create(redio_chain **h, const float *taps_host, size_t ntaps, size_t decim, int nfft, unsigned flags);
int redio_chain_destroy(redio_chain *h);
size_t redio_chain_nblocks(const redio_chain *h, size_t n_in);
int redio_chain_is_fused(const redio_chain *h);
/* force the two-kernel path (for measurement): 0 = fused when available, 1 = never fused */
int redio_chain_set_unfused(redio_chain *h, int unfused);
/* sizes the two-kernel path's intermediate for inputs of up to n_in samples, so that enqueue never allocates */
int redio_chain_reserve(redio_chain *h, size_t n_in);
int redio_chain_enqueue(redio_chain *h, const void *d_in, size_t n_in, void *d_out, void *stream);
/* The receiver's own format in: interleaved u8 I/Q bytes (rtlsdr::data_to_samples, src/rtlsdr/src/rtlsdr.rs:159-162: i as f32 / 127.0 - 1.0)
 * straight into the chain -- nbytes / 2 samples, the spectra redio_data_to_samples + redio_chain_enqueue would give, bit for bit.
 * The shapes with a fused cf32 kernel (above) on 4-byte aligned bytes are ONE kernel ((127, 5): 3.6 bytes per sample through HBM instead
 * of 25.6); other shapes convert
 * into a plan-owned buffer first (grown on first use: REDIO_ERR_NOT_RESERVED inside a stream capture). */
int redio_chain_enqueue_u8(redio_chain *h, const void *d_bytes, size_t nbytes, void *d_out, void *stream);
/* sizes that buffer (and the two-kernel path's intermediate) for messages of up to nbytes bytes, so that enqueue_u8 never allocates */
int redio_chain_reserve_u8(redio_chain *h, size_t nbytes);
/* launch geometry of the fused kernel for a call that yields nblocks blocks (0 for a plan that runs as two kernels): a wavefront
 * owns redio_chain_blocks_per_wave() consecutive blocks (the last one of the launch possibly fewer), the launch has
 * redio_chain_launch_waves() = ceil(nblocks / blocks_per_wave) wavefronts.  Tests pick run boundaries from it, the probes size
 * their stamp buffers from it. */
size_t redio_chain_blocks_per_wave(const redio_chain *h, size_t nblocks);
size_t redio_chain_launch_waves(const redio_chain *h, size_t nblocks);
/* the fused kernel's name as rocprofv3 reports it, spaces removed (e.g. "chain_v4_kernel<127,5,true,2,8,false,true,false,false>"); NULL for a
 * two-kernel plan.  bench.py refuses a counter file (profiles/rNN_traffic.json) recorded for another kernel.  Owned by the plan. */
const char *redio_chain_kernel_name(redio_chain *h);
/* diagnostic, per plan: while d_buf (device memory, capacity_waves records of 4 x u64) is set, wavefront w < capacity_waves of the
 * fused kernel of THIS plan writes {shader cycles, 100 MHz ticks, start tick, XCC/HW id} into record w (tools/clock_probe.py reads
 * the clock the chip holds from it); size it with redio_chain_launch_waves().  NULL (default) turns it off.  Timings of stamped
 * launches are never quoted. */
int redio_chain_set_debug_stamps(redio_chain *h, void *d_buf, size_t capacity_waves);

/* ---- A9: the bit-exact ingest / slicing path of the shipped graph (src/ratpak.rs:60-76) ----
 * All device-resident; results are bit-identical to the reference arithmetic (oracle_bits.c). */
/* rtlsdr::data_to_samples, src/rtlsdr/src/rtlsdr.rs:159-162: byte pairs -> cf32 (i as f32/127.0 - 1.0);
 * an odd byte count is the reference's index panic -> REDIO_ERR_ASSERT */
int redio_data_to_samples(const void *d_bytes, size_t nbytes, void *d_out_c32, void *stream);
/* the |x| map of src/ratpak.rs:64-68: Complex::norm = hypotf(re, im) */
int redio_norm_c32(const void *d_in_c32, size_t n, void *d_out_f32, void *stream);
/* both of the above fused: u8 IQ -> magnitude (2 B read + 4 B written per sample); d_bytes 8-byte and
 * d_mag 16-byte aligned */
int redio_ingest_u8_mag(const void *d_bytes, size_t nbytes, void *d_mag_f32, void *stream);
/* per-block sums in sample order, the `s` of bitfount::trigger (src/bitfount/src/bitfount.rs:48) */
int redio_block_sums(const void *d_in_f32, size_t nblocks, size_t block, void *d_sums_f32, void *stream);
/* bitfount::discretize, src/bitfount/src/bitfount.rs:87-96: max = fold(0.0, f32::max); out = (x > max/2)
 * as one byte per sample (the reference sends usize); d_scratch_u32 holds the max's bit pattern */
int redio_discretize(const void *d_in_f32, size_t n, void *d_out_u8, void *d_scratch_u32, void *stream);
/* bitfount::trigger, src/bitfount/src/bitfount.rs:36-85: state persists across calls; feeds nblocks
 * blocks of `block` f32 magnitudes; emitted buffers are appended to d_out with their lengths in
 * lens[] (host).  Synchronous (the adaptive threshold is a scalar recurrence run on the host).
 * If the buffers this call would emit do not fit (more than lens_cap of them, or more than out_cap
 * floats) nothing is consumed: REDIO_ERR_ARG with the needed counts in *nemit / *total, retry larger. */
typedef struct redio_trigger redio_trigger;
int redio_trigger_create(redio_trigger **h);
int redio_trigger_destroy(redio_trigger *h);
int redio_trigger_feed(redio_trigger *h, const void *d_blocks_f32, size_t nblocks, size_t block, void *d_out_f32, size_t out_cap,
                       size_t *lens, size_t lens_cap, size_t *nemit, size_t *total, void *stream);

/* ---- the run-length / bit-field stage downstream of discretize (src/ratpak.rs:77-119) ----
 * One-byte values (discretize emits 0/1), u64 run lengths, f32 seconds; all device-resident. */
/* kpn::rle, src/kpn/src/kpn.rs:17-29: a run is emitted when the value changes, so the open run is
 * carried in the handle across calls and never flushed.  *nruns runs written (<= cap).  Synchronous. */
typedef struct redio_rle redio_rle;
int redio_rle_create(redio_rle **h);
int redio_rle_destroy(redio_rle *h);
int redio_rle_feed(redio_rle *h, const void *d_in_u8, size_t n, void *d_vals_u8, void *d_counts_u64, size_t cap, size_t *nruns,
                   void *stream);
/* kpn::dle, kpn.rs:32-38: seconds = ct as f32 / s_rate as f32 */
int redio_dle(const void *d_counts_u64, size_t n, size_t s_rate, void *d_seconds_f32, void *stream);
/* kpn::rld, kpn.rs:50-56: (value, count) -> repeated values; d_scratch: (nruns + 1) u64 */
int redio_rld(const void *d_vals_u8, const void *d_counts_u64, size_t nruns, void *d_out_u8, size_t cap, void *d_scratch, size_t *nout,
              void *stream);
/* kpn::dld, kpn.rs:41-47: n = (dur * s_rate) as usize per run; d_scratch: (2 * nruns + 1) u64 */
int redio_dld(const void *d_vals_u8, const void *d_seconds_f32, size_t nruns, float s_rate, void *d_out_u8, size_t cap, void *d_scratch,
              size_t *nout, void *stream);
/* kpn::binconv = eat (kpn.rs:116-124) per message: nmsg messages of nbits one-byte binary digits ->
 * nfields u64 each, MSB first (b2d, kpn.rs:111-113); widths overrunning a message -> REDIO_ERR_ASSERT */
int redio_binconv(const void *d_bits_u8, size_t nmsg, size_t nbits, const size_t *widths, size_t nfields, void *d_out_u64, void *stream);

/* ---- C5: overlap-save FFT convolution (BASELINE.json configs[4]; a new composition) ----
 * The valid-mode correlation of dsputils::convolve (dsputils.rs:30-32) on cf32 with real taps, computed
 * per block of nfft samples: out[b*hop + i] = IFFT(FFT(x[b*hop ..]) .* conj(FFT(taps)))[i] / nfft, i < hop,
 * hop = nfft - ntaps + 1, kissfft-order transforms.  Only whole blocks are produced:
 * nout = ((n_in - nfft)/hop + 1) * hop, 0 when n_in < nfft.  Bit-identical to oracle orc_overlap_save;
 * within K*eps*sum|x*h| of the direct fold. */
typedef struct redio_ovsave redio_ovsave;
int redio_ovsave_create(redio_ovsave **h, const float *taps_host, size_t ntaps, int nfft);
int redio_ovsave_destroy(redio_ovsave *h);
size_t redio_ovsave_nout(const redio_ovsave *h, size_t n_in);
int redio_ovsave_enqueue(redio_ovsave *h, const void *d_in, size_t n_in, void *d_out, void *stream);

/* ---- C4: M-channel polyphase channelizer (BASELINE.json configs[3]; a new composition) ----
 * Prototype of nchan*taps_per_branch taps; branch m filters rows x_t[m] = x[nchan*t + m] with
 * g_m[p] = proto[nchan*p + m] using the fold of dsputils::convolve (dsputils.rs:31), then every row
 * goes through kissfft's nchan-point forward transform (kissfft.rs:26).  Output rows: T-P+1 with
 * T = floor(n_in / nchan).  nchan = 64 with taps_per_branch in {4, 8, 16} runs one fused kernel; any other
 * shape runs the branch filters and the nchan-point transform as two passes (same results).
 * flags: REDIO_FIR_FUSED as for redio_fir_create.
 * Output layout: ngroups = 1 -> [row][nchan]; ngroups = G -> [group][row][nchan/G], the send layout
 * of the multi-GPU regrouping (one contiguous chunk per destination rank). */
typedef struct redio_pfb redio_pfb;
int redio_pfb_create(redio_pfb **h, const float *proto_taps_host, int nchan, int taps_per_branch, unsigned flags);
int redio_pfb_destroy(redio_pfb *h);
size_t redio_pfb_nrows(const redio_pfb *h, size_t n_in);
int redio_pfb_enqueue(redio_pfb *h, const void *d_in, size_t n_in, void *d_out, int ngroups, void *stream);
/* the same from the receiver's interleaved u8 I/Q bytes (rtlsdr::data_to_samples, src/rtlsdr/src/rtlsdr.rs:159-162), nbytes / 2 samples,
 * bit for bit the rows of redio_data_to_samples + redio_pfb_enqueue: one kernel for 64 channels x 16 taps per branch (10 instead of
 * 26 bytes per sample through HBM), other shapes convert into a plan-owned buffer first (grown on first use). */
int redio_pfb_enqueue_u8(redio_pfb *h, const void *d_bytes, size_t nbytes, void *d_out, int ngroups, void *stream);
int redio_pfb_reserve_u8(redio_pfb *h, size_t nbytes, int ngroups); /* as redio_pfb_reserve, for messages of up to nbytes bytes */
/* scratch of the two-pass shapes for inputs of up to n_in samples.  The fused 64-channel kernel needs none; the one-kernel shapes (32 ...
 * 1024 channels x 4 / 8 / 16 taps per branch) need it only for an output that is not 16-byte aligned and reserve NOTHING unless
 * REDIO_PFB_RESERVE_TWO_PASS is or-ed into ngroups (un-reserved, that fall-back sizes its scratch at first use and returns
 * REDIO_ERR_NOT_RESERVED inside a capture). */
#define REDIO_PFB_RESERVE_TWO_PASS 0x40000000
int redio_pfb_reserve(redio_pfb *h, size_t n_in, int ngroups);
/* the same as redio_pfb_reserve(h, n_in, ngroups | REDIO_PFB_RESERVE_TWO_PASS) under a name of its own: sizes the two-pass scratch of ANY
 * shape without a fused 64-channel kernel, so that an output that is not 16-byte aligned never allocates (e.g. inside a capture) */
int redio_pfb_reserve_two_pass(redio_pfb *h, size_t n_in, int ngroups);

/* ---- the channelizer's one exchange step over RCCL / xGMI (the only collective on the path) ----
 * Time-sharded channelizer: every rank runs redio_pfb_enqueue(..., ngroups = G) on its own slice of the stream and
 * holds d_grouped = [G groups][its rows][chans_per_rank cf32]; the exchange sends group q to rank q
 * (ncclGroupStart; ncclSend/ncclRecv per peer; ncclGroupEnd), so that afterwards rank g holds
 * d_out = [sum(rows_per_rank) rows, in rank (= time) order][chans_per_rank cf32] for ITS channels.
 * rows_per_rank[G] (host) may be ragged.  Enqueued on `stream`; nothing synchronises.
 * librccl.so is loaded on first use; REDIO_ERR_COMM (text: redio_comm_last_error) if it is missing or a call fails.
 * One rank per process: rank 0 calls redio_comm_unique_id, the launcher hands the 128 bytes to every rank, every rank
 * calls redio_comm_init_rank on its own device.  Several ranks in one process (the reference's thread-per-block host):
 * redio_comm_init_all(comms, ndev, devices) then redio_pfb_exchange_all, which wraps all ranks' transfers in one group. */
#define REDIO_COMM_ID_BYTES 128
typedef struct redio_comm redio_comm;
int redio_comm_unique_id(void *id128);
int redio_comm_init_rank(redio_comm **c, int nranks, int rank, const void *id128);
int redio_comm_init_all(redio_comm **comms, int ndev, const int *devices /* NULL = 0..ndev-1 */);
int redio_comm_destroy(redio_comm *c);
int redio_comm_rank(const redio_comm *c);
int redio_comm_size(const redio_comm *c);
const char *redio_comm_last_error(void); /* text of the calling thread's last REDIO_ERR_COMM */
int redio_pfb_exchange(redio_comm *c, const void *d_grouped, void *d_out, const size_t *rows_per_rank, size_t chans_per_rank, void *stream);
/* the same with rank q's rows placed at row out_row_offset[q] of d_out: lets a slice be analysed in pieces whose exchanges run beside
 * the analysis of the next piece on another HIP stream and still build the time-ordered result in place */
int redio_pfb_exchange_at(redio_comm *c, const void *d_grouped, void *d_out, const size_t *rows_per_rank, const size_t *out_row_offset,
                          size_t chans_per_rank, void *stream);
int redio_pfb_exchange_all(redio_comm *const *comms, int ndev, const void *const *d_grouped, void *const *d_out,
                           const size_t *rows_per_rank, size_t chans_per_rank, void *const *streams);

/* ---- carried history: the windowed plans above as STREAMS (BASELINE.json configs[1] "history carried") ----
 * redio_fir_enqueue & co. are stateless per call, like dsputils::convolve (dsputils.rs:30-32), which loses ntaps-1
 * outputs at every message seam.  A *_stream handle sits on a plan (not owned: destroy the stream first) and keeps the
 * stream's unconsumed tail (fewer samples than one window) on the device, so that feeding a stream in ANY pieces gives
 * exactly the bits of one stateless call on the whole stream:
 *     fir:    y[i] = fold_j x[decim*i + j]*taps[j] for every i whose window has arrived (decimation phase 0 at stream start)
 *     chain:  spectrum b from decimated samples [b*nfft, (b+1)*nfft) of that y
 *     pfb:    row t from input rows t .. t+taps_per_branch-1 (layout [row][nchan] only)
 *     ovsave: block b -> hop outputs, blocks every hop samples from the stream start
 * enqueue() writes *nout (= redio_*_stream_nout(h, n_new), known before the call) output samples to d_out; the new
 * samples are read in place (only a seam of fewer than one window is staged through a plan-owned buffer).  It only
 * launches kernels and small device copies -- with one exception: a plan shape that runs as two kernels (redio_chain_is_fused() == 0,
 * or a body that starts on an odd sample) keeps a plan-owned intermediate that *_stream_create sizes for the seam windows and
 * that grows on a longer body: call redio_chain_reserve(plan, largest message) once to keep enqueue allocation-free.
 * The host-side counters advance only when every kernel of the call has been launched (a failed call can be repeated); feed one
 * stream from one thread in order -- a call that finds another thread inside enqueue() / reset() of the same handle returns
 * REDIO_ERR_ARG.  Not graph-capturable (the split changes from call to call).  pending() = samples carried. */
typedef struct redio_fir_stream redio_fir_stream;
int redio_fir_stream_create(redio_fir_stream **h, redio_fir *plan);
int redio_fir_stream_destroy(redio_fir_stream *h);
int redio_fir_stream_reset(redio_fir_stream *h);
size_t redio_fir_stream_nout(const redio_fir_stream *h, size_t n_new);
size_t redio_fir_stream_pending(const redio_fir_stream *h);
int redio_fir_stream_enqueue(redio_fir_stream *h, const void *d_new, size_t n_new, void *d_out, size_t *nout, void *stream);
typedef struct redio_chain_stream redio_chain_stream;
int redio_chain_stream_create(redio_chain_stream **h, redio_chain *plan);
int redio_chain_stream_destroy(redio_chain_stream *h);
int redio_chain_stream_reset(redio_chain_stream *h);
size_t redio_chain_stream_nout(const redio_chain_stream *h, size_t n_new);
size_t redio_chain_stream_pending(const redio_chain_stream *h);
int redio_chain_stream_enqueue(redio_chain_stream *h, const void *d_new, size_t n_new, void *d_out, size_t *nout, void *stream);
typedef struct redio_pfb_stream redio_pfb_stream;
int redio_pfb_stream_create(redio_pfb_stream **h, redio_pfb *plan);
/* the chain / channelizer streams fed with the receiver's interleaved u8 I/Q bytes (rtlsdr.rs:127-162): d_new points to bytes,
 * n_new still counts SAMPLES (2 bytes each); the history is carried as bytes and every window runs redio_*_enqueue_u8.
 * All other redio_{chain,pfb}_stream_* calls apply unchanged. */
int redio_chain_stream_create_u8(redio_chain_stream **h, redio_chain *plan);
int redio_pfb_stream_create_u8(redio_pfb_stream **h, redio_pfb *plan);
int redio_pfb_stream_destroy(redio_pfb_stream *h);
int redio_pfb_stream_reset(redio_pfb_stream *h);
size_t redio_pfb_stream_nout(const redio_pfb_stream *h, size_t n_new);
size_t redio_pfb_stream_pending(const redio_pfb_stream *h);
int redio_pfb_stream_enqueue(redio_pfb_stream *h, const void *d_new, size_t n_new, void *d_out, size_t *nout, void *stream);
typedef struct redio_ovsave_stream redio_ovsave_stream;
int redio_ovsave_stream_create(redio_ovsave_stream **h, redio_ovsave *plan);
int redio_ovsave_stream_destroy(redio_ovsave_stream *h);
int redio_ovsave_stream_reset(redio_ovsave_stream *h);
size_t redio_ovsave_stream_nout(const redio_ovsave_stream *h, size_t n_new);
size_t redio_ovsave_stream_pending(const redio_ovsave_stream *h);
int redio_ovsave_stream_enqueue(redio_ovsave_stream *h, const void *d_new, size_t n_new, void *d_out, size_t *nout, void *stream);

/* ---- A6: samplerate::resample's native side, src/samplerate/src/samplerate.rs:59-87 ----
 * nchan independent mono streams that share ratio and block lengths (the reference creates one
 * src_new(SRC_SINC_MEDIUM_QUALITY, 1) state per block, :61).  Control flow, output count law and
 * arithmetic order are those of the published libsamplerate 0.1.8 sinc converter; the coefficient
 * table is a stated design.  Bit-identical to oracle/oracle_src.c.
 * Return values: REDIO_OK, a negative redio error, or one of the library's positive error codes. */
enum {
    REDIO_SRC_ERR_MALLOC_FAILED = 1, REDIO_SRC_ERR_BAD_STATE = 2, REDIO_SRC_ERR_BAD_DATA = 3,
    REDIO_SRC_ERR_BAD_DATA_PTR = 4, REDIO_SRC_ERR_NO_PRIVATE = 5, REDIO_SRC_ERR_BAD_SRC_RATIO = 6,
    REDIO_SRC_ERR_BAD_PROC_PTR = 7, REDIO_SRC_ERR_SHIFT_BITS = 8, REDIO_SRC_ERR_FILTER_LEN = 9,
    REDIO_SRC_ERR_BAD_CONVERTER = 10, REDIO_SRC_ERR_BAD_CHANNEL_COUNT = 11,
    REDIO_SRC_ERR_SINC_BAD_BUFFER_LEN = 12, REDIO_SRC_ERR_SIZE_INCOMPATIBILITY = 13,
    REDIO_SRC_ERR_BAD_PRIV_PTR = 14, REDIO_SRC_ERR_BAD_SINC_STATE = 15, REDIO_SRC_ERR_DATA_OVERLAP = 16,
    REDIO_SRC_ERR_BAD_CALLBACK = 17, REDIO_SRC_ERR_BAD_MODE = 18, REDIO_SRC_ERR_NULL_CALLBACK = 19,
    REDIO_SRC_ERR_NO_VARIABLE_RATIO = 20, REDIO_SRC_ERR_SINC_PREPARE_DATA_BAD_LEN = 21,
    REDIO_SRC_ERR_BAD_INTERNAL_STATE = 22
};
typedef struct redio_src redio_src;
/* converter (samplerate.rs:26-30): 0 best / 1 medium / 2 fastest sinc, 3 zero-order hold, 4 linear; anything else
 * REDIO_SRC_ERR_BAD_CONVERTER (the reference only ever asks for 1).  nchan independent mono streams, stored as rows. */
int redio_src_create(redio_src **h, int converter, int nchan);
int redio_src_destroy(redio_src *h);
int redio_src_reset(redio_src *h);
int redio_src_set_ratio(redio_src *h, double ratio);
/* arithmetic of the device-resident form.  EXACT (default): double accumulation in the library's
 * order, bit-identical to the oracle; calls whose phase is uniform (constant ratio, 1/ratio an
 * integer, no end_of_input) run as one launch.  FAST: those uniform calls use an f32 polyphase
 * filter bank (taps rounded to f32, f32 accumulation) - NOT bit-identical, error bounded by
 * ntaps * 2^-23 * sum|h| * max|x|; every other call still runs EXACT.  EPOCHS: EXACT with one
 * launch per buffer refill, the literal schedule of the library (kept for cross-checks). */
enum { REDIO_SRC_EXACT = 0, REDIO_SRC_FAST = 1, REDIO_SRC_EPOCHS = 2 };
int redio_src_set_mode(redio_src *h, int mode);
/* device-resident: d_in[nchan][in_stride], d_out[nchan][out_stride] f32; synchronises the stream
 * before returning (per-output parameters are produced by the host state machine) */
int redio_src_process(redio_src *h, const void *d_in, long input_frames, long in_stride, void *d_out, long output_frames,
                      long out_stride, double src_ratio, int end_of_input, long *input_frames_used,
                      long *output_frames_gen, void *stream);
/* host buffers, interleaved frames of the handle's nchan channels (mono: plain samples), synchronous: the body of the
 * src_process drop-in (include/samplerate.h); every channel is converted exactly as a mono stream */
int redio_src_process_host(redio_src *h, const float *data_in, long input_frames, float *data_out, long output_frames,
                           double src_ratio, int end_of_input, long *input_frames_used, long *output_frames_gen);
/* diagnostics: buffer-refill epochs of this handle served by the periodic-phase kernel (constant rational ratios such as
 * 48000/44100 or 2.0: P sets of coefficients, LDS tiles) and by the general per-tap kernel; both bit-identical */
int redio_src_path_counts(const redio_src *h, long *periodic, long *general);
/* the coefficient table of a converter (coeffs_out may be NULL; it holds half_len + 2 floats) */
int redio_src_table(int converter, float *coeffs_out, int *half_len, int *increment);

/* ---- kpn vector maps on device: mul_vecs (src/kpn/src/kpn.rs:198-203) and sum_vecs (:227-231) ----
 * out[i] = a[i] * b[i] / a[i] + b[i] for i < n (the caller passes n = min of the two lengths, as zip does);
 * f32 and Complex<f32> ((ar*br - ai*bi, ar*bi + ai*br), every operation rounded on its own). */
int redio_mul_f32(const void *d_a, const void *d_b, void *d_out, size_t n, void *stream);
int redio_add_f32(const void *d_a, const void *d_b, void *d_out, size_t n, void *stream);
int redio_mul_c32(const void *d_a, const void *d_b, void *d_out, size_t n, void *stream);
int redio_add_c32(const void *d_a, const void *d_b, void *d_out, size_t n, void *stream);

/* order-free 64-bit sum of the n 32-bit words at d_words, ADDED to the u64 at d_sum_u64 (device memory the caller zeroed; one atomic
 * per workgroup): the checking sink of a device-resident graph -- reads every word a block produced, exact whatever the order */
int redio_checksum_u32(const void *d_words, size_t n, void *d_sum_u64, void *stream);

/* ---- synthetic input: hash-generated cf32 / f32 in [-1, 1), device side ---- */
int redio_synth_iq(void *d_out, uint32_t seed, uint64_t first_sample, size_t n, void *stream);
int redio_synth_f32(void *d_out, uint32_t seed, uint64_t first_sample, size_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* REDIO_H */
