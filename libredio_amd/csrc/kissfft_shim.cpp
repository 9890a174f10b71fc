// kissfft_shim.cpp -- libkissfft.so: the C symbols of src/kissfft/src/kissfft.rs:11-16 on top of
// the redio FFT plan.  See include/kiss_fft.h for the contract.
#include "../../include/kiss_fft.h"
#include "../../include/redio.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdint.h>

struct kiss_fft_state {
    int nfft;
    int inverse;
    int on_heap;
    redio_fft *plan;
    void *d_buf;   // nfft complex on the device
    void *stream;
    kiss_fft_cpx *gather; // host scratch for strided input
    // small transforms: pinned, device-addressable message buffers (zero-copy) instead of two staged copies
    void *pin_in, *pin_in_dev, *pin_out, *pin_out_dev;
    // completion word in pinned memory, written by the stream behind the kernel (redio_stream_signal): the call polls it instead of
    // entering hipStreamSynchronize
    void *flag, *flag_dev;
    uint32_t seq;
};
enum { KISS_ZERO_COPY_MAX = 8192 };
// wall-time bound of the completion poll (nanoseconds): 100 us by default; redio_kiss_fft_set_spin_ns is a test hook (0 forces the fall-back
// to redio_stream_sync on every call, tests/test_gpu_parity.py) and not part of the published kiss_fft interface
static long g_kiss_spin_ns = 100000;
static long kiss_spin_ns(void) { return __atomic_load_n(&g_kiss_spin_ns, __ATOMIC_RELAXED); }
extern "C" void redio_kiss_fft_set_spin_ns(long ns) { __atomic_store_n(&g_kiss_spin_ns, ns < 0 ? 0 : ns, __ATOMIC_RELAXED); }

extern "C" kiss_fft_cfg kiss_fft_alloc(int nfft, int inverse_fft, void *mem, size_t *lenmem)
{
    // the published contract takes ANY mem with *lenmem >= the reported size (a cfg carved out of a byte arena): the size reported
    // includes alignof - 1 bytes of slack and the state sits at the first suitably aligned address inside mem (advisor, round 5)
    const size_t need = sizeof(kiss_fft_state) + alignof(kiss_fft_state) - 1;
    kiss_fft_state *st = NULL;
    if (lenmem == NULL) {
        st = (kiss_fft_state *)malloc(sizeof(kiss_fft_state));
        if (st) st->on_heap = 1;
    } else {
        if (mem != NULL && *lenmem >= need) {
            const uintptr_t a = ((uintptr_t)mem + alignof(kiss_fft_state) - 1) & ~(uintptr_t)(alignof(kiss_fft_state) - 1);
            st = (kiss_fft_state *)a;
            st->on_heap = 0;
        }
        *lenmem = need;
    }
    if (!st) return NULL;
    st->nfft = nfft; st->inverse = inverse_fft; st->plan = NULL; st->d_buf = NULL; st->stream = NULL; st->gather = NULL;
    st->pin_in = st->pin_in_dev = st->pin_out = st->pin_out_dev = NULL; st->flag = st->flag_dev = NULL; st->seq = 0;
    if (nfft <= 0) { if (st->on_heap) free(st); return NULL; }
    int rc = redio_fft_create(&st->plan, nfft, inverse_fft);
    if (rc == REDIO_OK) rc = redio_malloc(&st->d_buf, (size_t)nfft * sizeof(kiss_fft_cpx));
    if (rc == REDIO_OK) rc = redio_stream_create(&st->stream);
    if (rc == REDIO_OK && nfft <= KISS_ZERO_COPY_MAX) {
        // best effort: without mapped pinned memory the copy path below serves every size
        if (redio_host_alloc(&st->pin_in, &st->pin_in_dev, (size_t)nfft * sizeof(kiss_fft_cpx)) != REDIO_OK ||
            redio_host_alloc(&st->pin_out, &st->pin_out_dev, (size_t)nfft * sizeof(kiss_fft_cpx)) != REDIO_OK) {
            redio_host_free(st->pin_in); redio_host_free(st->pin_out);
            st->pin_in = st->pin_in_dev = st->pin_out = st->pin_out_dev = NULL;
        } else if (redio_host_alloc(&st->flag, &st->flag_dev, 64) == REDIO_OK) {
            *(volatile uint32_t *)st->flag = 0;
        } else st->flag = st->flag_dev = NULL;
    }
    if (rc != REDIO_OK) {
        fprintf(stderr, "kiss_fft_alloc(%d): %s\n", nfft, redio_strerror(rc));
        redio_fft_destroy(st->plan);
        redio_free(st->d_buf);
        if (st->on_heap) free(st);
        return NULL;
    }
    return st;
}

extern "C" void kiss_fft_stride(kiss_fft_cfg st, const kiss_fft_cpx *fin, kiss_fft_cpx *fout, int in_stride)
{
    if (!st || !fin || !fout) return;
    const size_t bytes = (size_t)st->nfft * sizeof(kiss_fft_cpx);
    const kiss_fft_cpx *src = fin;
    if (in_stride != 1) {
        if (!st->gather) st->gather = (kiss_fft_cpx *)malloc(bytes);
        if (!st->gather) { // no error return in this interface (kissfft.rs:14): a poisoned block, as for a device failure
            fprintf(stderr, "kiss_fft_stride: out of memory (%zu bytes)\n", bytes);
            for (int i = 0; i < st->nfft; ++i) fout[i].r = fout[i].i = NAN;
            return;
        }
        for (int i = 0; i < st->nfft; ++i) st->gather[i] = fin[(size_t)i * in_stride];
        src = st->gather;
    }
    if (st->pin_out) { // the kernel reads and writes the pinned message buffers across PCIe itself
        memcpy(st->pin_in, src, bytes);
        int rc0 = redio_fft_enqueue(st->plan, st->pin_in_dev, st->pin_out_dev, 1, st->stream);
        if (rc0 == REDIO_OK) {
            const uint32_t seq = ++st->seq;
            volatile uint32_t *word = (volatile uint32_t *)st->flag;
            if (word && redio_stream_signal(st->stream, st->flag_dev, seq) == REDIO_OK) {
                // The word lands a few microseconds after the kernel.  Poll for at most KISS_SPIN_NS of wall time, then the ordinary wait.
                // Ordering: the stream writes the word behind the kernel (hipStreamWriteValue32 is ordered after all prior work of the
                // stream), the message buffers and the word are host-coherent pinned memory (redio_host_alloc: hipHostMallocMapped), so a
                // host that sees the word sees the kernel's stores to pin_out; the acquire fence keeps the memcpy below behind the poll.
                struct timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                for (;;) {
                    for (int i = 0; i < 64 && *word != seq; ++i) __builtin_ia32_pause();
                    if (*word == seq) break;
                    clock_gettime(CLOCK_MONOTONIC, &t1);
                    if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > kiss_spin_ns()) break;
                }
                if (*word != seq) rc0 = redio_stream_sync(st->stream);
                else __atomic_thread_fence(__ATOMIC_ACQUIRE);
            } else rc0 = redio_stream_sync(st->stream);
        }
        if (rc0 == REDIO_OK) { memcpy(fout, st->pin_out, bytes); return; }
        fprintf(stderr, "kiss_fft: %s\n", redio_strerror(rc0));
        for (int i = 0; i < st->nfft; ++i) fout[i].r = fout[i].i = NAN;
        return;
    }
    int rc = redio_upload(st->d_buf, src, bytes, st->stream);
    if (rc == REDIO_OK) rc = redio_fft_enqueue(st->plan, st->d_buf, st->d_buf, 1, st->stream);
    if (rc == REDIO_OK) rc = redio_download(fout, st->d_buf, bytes, st->stream);
    if (rc == REDIO_OK) rc = redio_stream_sync(st->stream);
    if (rc != REDIO_OK) {
        fprintf(stderr, "kiss_fft: %s\n", redio_strerror(rc));
        for (int i = 0; i < st->nfft; ++i) fout[i].r = fout[i].i = NAN;
    }
}

extern "C" void kiss_fft(kiss_fft_cfg cfg, const kiss_fft_cpx *fin, kiss_fft_cpx *fout) { kiss_fft_stride(cfg, fin, fout, 1); }
extern "C" void kiss_fft_cleanup(void) {}

extern "C" int kiss_fft_next_fast_size(int n)
{
    for (;; ++n) {
        int m = n;
        while ((m % 2) == 0) m /= 2;
        while ((m % 3) == 0) m /= 3;
        while ((m % 5) == 0) m /= 5;
        if (m <= 1) break;
    }
    return n;
}

extern "C" void kiss_fft_free(kiss_fft_cfg st)
{
    if (!st) return;
    redio_fft_destroy(st->plan);
    redio_free(st->d_buf);
    redio_host_free(st->pin_in);
    redio_host_free(st->pin_out);
    redio_host_free(st->flag);
    redio_stream_destroy(st->stream);
    free(st->gather);
    if (st->on_heap) free(st);
}
