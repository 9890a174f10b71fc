/*
 * kiss_fft.h -- drop-in for the three C symbols LibRedio's Rust binds from libkissfft
 * (src/kissfft/src/kissfft.rs:11-16; linked as dylib=kissfft, src/kissfft/build.rs:6):
 *     fn kiss_fft_alloc(nfft: u32, inverse_fft: u32, mem: *mut u8, lenmem: *mut u64) -> *const u8;
 *     fn kiss_fft(cfg: *const u8, fin: *const Complex<f32>, fout: *mut Complex<f32>);
 *     fn kiss_fft_cleanup();
 * Exported by libkissfft.so in this repo; the transform itself runs on the MI355X through
 * redio_fft_* (include/redio.h).  Host pageable buffers in and out, synchronous: the result is in
 * fout when kiss_fft returns (kissfft.rs:26-27).  fin == fout is allowed.
 *
 * The reference has no way to report a failure here (no return value, cfg unchecked at
 * kissfft.rs:19).  On a HIP failure kiss_fft_alloc returns NULL; kiss_fft fills fout with NaN and
 * writes one line to stderr rather than leave fout (uninitialised at kissfft.rs:21-22) untouched.
 */
#ifndef KISS_FFT_H
#define KISS_FFT_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float r, i; } kiss_fft_cpx; /* = num::complex::Complex<f32>, interleaved */
typedef struct kiss_fft_state *kiss_fft_cfg;

/* mem/lenmem placement protocol of the published API: lenmem == NULL -> heap; otherwise *lenmem is
 * set to the bytes needed (alignment slack included) and the cfg is placed inside mem when it is non-NULL and large enough (else NULL);
 * any alignment of mem is accepted, the cfg returned is the first suitably aligned address inside it. */
kiss_fft_cfg kiss_fft_alloc(int nfft, int inverse_fft, void *mem, size_t *lenmem);
void kiss_fft(kiss_fft_cfg cfg, const kiss_fft_cpx *fin, kiss_fft_cpx *fout);
void kiss_fft_stride(kiss_fft_cfg cfg, const kiss_fft_cpx *fin, kiss_fft_cpx *fout, int fin_stride);
void kiss_fft_cleanup(void);             /* no global state: no-op (kissfft.rs:30 is unreachable) */
int kiss_fft_next_fast_size(int n);      /* next n whose only prime factors are 2, 3, 5 */
void kiss_fft_free(kiss_fft_cfg cfg);    /* releases the device plan (the published macro is free()) */
/* Not in the published interface: the wall-time bound (nanoseconds, default 100 000) of the completion poll a call makes before it falls
 * back to an ordinary stream wait; 0 takes the fall-back on every call (a test hook). */
void redio_kiss_fft_set_spin_ns(long ns);

#ifdef __cplusplus
}
#endif
#endif
