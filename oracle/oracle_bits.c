/*
 * oracle_bits.c -- CPU restatement of the bit-exact slicing/indexing paths.
 * TEST INFRASTRUCTURE ONLY (see redio_oracle.h).  Parity unpinned by reference tests (there are
 * none); every function here is exact integer / IEEE compare arithmetic with a single possible
 * result, pinned by the known answers in tests/test_oracle_bits.py (e.g. b2d([1,0,1]) == 5 and the
 * width lists of src/ratpak.rs:115,119).
 */
#include "redio_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* src/kpn/src/kpn.rs:111-113: MSB-first list of binary digits -> unsigned */
size_t orc_b2d(const size_t *bits, size_t n)
{
    size_t acc = 0;
    for (size_t i = 0; i < n; ++i) acc += ((size_t)1 << (n - i - 1)) * bits[i];
    return acc;
}

/* src/kpn/src/kpn.rs:116-124: split by a list of widths; a width list that overruns the input
 * panics in the reference (slice out of bounds) -> -1 here */
int orc_eat(const size_t *bits, size_t nbits, const size_t *widths, size_t nw, size_t *out)
{
    size_t i = 0;
    for (size_t w = 0; w < nw; ++w) {
        if (i + widths[w] > nbits) return -1;
        out[w] = orc_b2d(bits + i, widths[w]);
        i += widths[w];
    }
    return 0;
}

/* src/bitfount/src/bitfount.rs:87-96: max = fold(0.0, f32::max); emit (x > max/2) as usize.
 * f32::max ignores a NaN operand (fmaxf semantics). */
void orc_discretize(const float *x, size_t n, size_t *out)
{
    float mx = 0.0f;
    for (size_t i = 0; i < n; ++i) mx = fmaxf(mx, x[i]);
    float thr = mx / 2.0f;
    for (size_t i = 0; i < n; ++i) out[i] = (x[i] > thr) ? 1 : 0;
}

/* src/rtlsdr/src/rtlsdr.rs:159-162: i2f(i) = i as f32/127.0 - 1.0; byte pairs -> cf32; an odd
 * byte count panics at i[1] -> -1 */
int orc_data_to_samples(const uint8_t *d, size_t n, orc_cpx *out)
{
    if (n & 1) return -1;
    for (size_t s = 0; s < n / 2; ++s) {
        out[s].r = (float)d[2 * s] / 127.0f - 1.0f;
        out[s].i = (float)d[2 * s + 1] / 127.0f - 1.0f;
    }
    return 0;
}

/* src/bitfount/src/bitfount.rs:48: samples.iter().map(|&x|x).sum() -- sequential f32 from 0.0 */
float orc_block_sum(const float *x, size_t n)
{
    float s = 0.0f;
    for (size_t i = 0; i < n; ++i) s = s + x[i];
    return s;
}

/* src/bitfount/src/bitfount.rs:36-85 */
struct orc_trigger_state {
    long trigger;       /* :42 */
    float threshold;    /* :44 */
    float *buf;         /* sample_buffer :43, starts as [0.0] */
    size_t len, cap;
};

static void trig_push(orc_trigger_state *t, const float *x, size_t n)
{
    if (t->len + n > t->cap) {
        size_t nc = t->cap ? t->cap : 1024;
        while (nc < t->len + n) nc *= 2;
        t->buf = (float *)realloc(t->buf, nc * sizeof(float));
        t->cap = nc;
    }
    memcpy(t->buf + t->len, x, n * sizeof(float));
    t->len += n;
}

orc_trigger_state *orc_trigger_new(void)
{
    orc_trigger_state *t = (orc_trigger_state *)calloc(1, sizeof(*t));
    float z = 0.0f;
    trig_push(t, &z, 1);
    return t;
}

void orc_trigger_free(orc_trigger_state *t)
{
    if (!t) return;
    free(t->buf);
    free(t);
}

size_t orc_trigger_feed(orc_trigger_state *t, const float *blocks, size_t nblocks, size_t block,
                        float *out, size_t out_cap, size_t *out_lens, size_t lens_cap, size_t *out_total)
{
    const long trigger_duration = 50;   /* :41 */
    const size_t block_size = 512;      /* :38 (only used in the OOM bound) */
    size_t nemit = 0, total = 0;
    for (size_t b = 0; b < nblocks; ++b) {
        const float *samples = blocks + b * block;
        t->trigger -= 1;                                              /* :46 */
        float s = orc_block_sum(samples, block);                      /* :48 */
        if (t->len > 1000 * (size_t)trigger_duration * block_size) {  /* :52-54 */
            t->len = 0;
            float z = 0.0f;
            trig_push(t, &z, 1);
        }
        if (t->threshold == 0.0f) t->threshold = s;                   /* :57-59 */
        if (t->trigger < 0) {                                         /* :62-65 */
            t->threshold += s / 1000.0f;
            t->threshold -= t->threshold * 0.002f;
        }
        if (s > t->threshold * 4.0f) t->trigger = trigger_duration;   /* :68-70 */
        if (t->trigger > 1) trig_push(t, samples, block);             /* :73-75 */
        if (t->trigger == 0) {                                        /* :78-81 */
            if (nemit < lens_cap && total + t->len <= out_cap) {
                memcpy(out + total, t->buf, t->len * sizeof(float));
                out_lens[nemit] = t->len;
            }
            total += t->len;
            ++nemit;
            t->len = 0;
        }
    }
    if (out_total) *out_total = total;
    return nemit;
}

/* |x| of the shipped graph's first map, src/ratpak.rs:64-68: cross_applicator_vecs(.., |x|{x.norm()})
 * with num 0.1.22 Complex::norm = re.hypot(im) (libm hypotf). */
void orc_norm_c32(const orc_cpx *x, size_t n, float *out)
{
    for (size_t i = 0; i < n; ++i) out[i] = hypotf(x[i].r, x[i].i);
}

/* kpn::mul_vecs (src/kpn/src/kpn.rs:198-203) and kpn::sum_vecs (:227-231): zip of the message with a
 * constant vector; f32 and Complex<f32> (num 0.1.22 Mul: (ar*br - ai*bi, ar*bi + ai*br)). */
void orc_zip_f32(const float *a, const float *b, size_t n, int add, float *out)
{
    for (size_t i = 0; i < n; ++i) out[i] = add ? a[i] + b[i] : a[i] * b[i];
}
void orc_zip_c32(const orc_cpx *a, const orc_cpx *b, size_t n, int add, orc_cpx *out)
{
    for (size_t i = 0; i < n; ++i) {
        if (add) { out[i].r = a[i].r + b[i].r; out[i].i = a[i].i + b[i].i; }
        else {
            const float re = a[i].r * b[i].r - a[i].i * b[i].i, im = a[i].r * b[i].i + a[i].i * b[i].r;
            out[i].r = re; out[i].i = im;
        }
    }
}
