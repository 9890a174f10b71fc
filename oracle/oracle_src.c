/*
 * oracle_src.c -- CPU restatement of the libsamplerate sinc converter that
 * src/samplerate/src/samplerate.rs:32-42,59-87 binds (src_new / src_process / ...), mono only
 * (the reference always passes channels = 1, samplerate.rs:61).  TEST INFRASTRUCTURE ONLY.
 *
 * libsamplerate is NOT in /root/reference (system library, version unpinned).  This restates the
 * published libsamplerate 0.1.8 algorithm (samplerate.c src_process + src_sinc.c
 * sinc_mono_vari_process / prepare_data / calc_output_single): band-limited interpolation over a
 * tabulated half-window with 12-bit fixed-point table indices, linear interpolation between
 * adjacent coefficients, double accumulation, the filter stretched by 1/ratio and scaled by ratio
 * when down-sampling, and the same internal buffer management -- so that output_frames_gen and
 * input_frames_used follow the same law call by call.
 *
 * PARITY UNPINNED, and sample-level parity with the real library is impossible here: its
 * coefficient tables (fastest 2464 / medium 22438 / best 340239 floats) cannot be reproduced.  The
 * tables below have the same sizes and increments (128 / 491 / 2381 entries per zero crossing of
 * the un-stretched sinc) but are a stated design: h[i] = fc * sinc(fc*i/inc) * kaiser(i/half, beta),
 * computed in double and rounded to float, with (fc, beta) = (0.80, 9.0) / (0.9425, 12.4) /
 * (0.9650, 16.0) chosen to land in the documented quality classes.
 */
#include "redio_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum {
    SRC_ERR_NO_ERROR = 0, SRC_ERR_MALLOC_FAILED, SRC_ERR_BAD_STATE, SRC_ERR_BAD_DATA,
    SRC_ERR_BAD_DATA_PTR, SRC_ERR_NO_PRIVATE, SRC_ERR_BAD_SRC_RATIO, SRC_ERR_BAD_PROC_PTR,
    SRC_ERR_SHIFT_BITS, SRC_ERR_FILTER_LEN, SRC_ERR_BAD_CONVERTER, SRC_ERR_BAD_CHANNEL_COUNT,
    SRC_ERR_SINC_BAD_BUFFER_LEN, SRC_ERR_SIZE_INCOMPATIBILITY, SRC_ERR_BAD_PRIV_PTR,
    SRC_ERR_BAD_SINC_STATE, SRC_ERR_DATA_OVERLAP, SRC_ERR_BAD_CALLBACK, SRC_ERR_BAD_MODE,
    SRC_ERR_NULL_CALLBACK, SRC_ERR_NO_VARIABLE_RATIO, SRC_ERR_SINC_PREPARE_DATA_BAD_LEN,
    SRC_ERR_BAD_INTERNAL_STATE
};
#define SRC_MAX_RATIO 256
#define SHIFT_BITS 12
#define FP_ONE ((double)(1 << SHIFT_BITS))
#define INV_FP_ONE (1.0 / FP_ONE)

/* ---- coefficient tables (stated design, see header) ---- */
typedef struct { int n; int increment; double fc; double beta; float *c; } src_table;
static src_table g_tab[3] = {
    {340239, 2381, 0.9650, 16.0, NULL}, /* 0 SRC_SINC_BEST_QUALITY   */
    {22438, 491, 0.9425, 12.4, NULL},   /* 1 SRC_SINC_MEDIUM_QUALITY */
    {2464, 128, 0.80, 9.0, NULL},       /* 2 SRC_SINC_FASTEST        */
};

static double bessel_i0(double x)
{
    double sum = 1.0, term = 1.0, q = x * x / 4.0;
    for (int k = 1; k < 500; ++k) {
        term *= q / ((double)k * (double)k);
        sum += term;
        if (term < 1e-18 * sum) break;
    }
    return sum;
}

static const src_table *get_table(int type)
{
    if (type < 0 || type > 2) return NULL;
    src_table *t = &g_tab[type];
    if (!t->c) {
        const double pi = 3.14159265358979323846;
        float *c = (float *)malloc((size_t)t->n * sizeof(float));
        int half = t->n - 2; /* coeff_half_len = ARRAY_LEN(coeffs) - 2 */
        double i0b = bessel_i0(t->beta);
        for (int i = 0; i < t->n; ++i) {
            double v = 0.0;
            if (i <= half) {
                double x = (double)i / (double)t->increment; /* in input samples at ratio >= 1 */
                double a = pi * t->fc * x;
                double s = (i == 0) ? 1.0 : sin(a) / a;
                double r = (double)i / (double)half;
                double w = bessel_i0(t->beta * sqrt(1.0 - r * r)) / i0b;
                v = t->fc * s * w;
            }
            c[i] = (float)v;
        }
        t->c = c;
    }
    return t;
}

int orc_src_table(int type, const float **coeffs, int *half_len, int *increment)
{
    const src_table *t = get_table(type);
    if (!t) return SRC_ERR_BAD_CONVERTER;
    *coeffs = t->c;
    *half_len = t->n - 2;
    *increment = t->increment;
    return 0;
}

/* ---- converter state (SRC_PRIVATE + SINC_FILTER of the published code, mono) ---- */
struct orc_src_state {
    double last_ratio, last_position;
    int type;
    const float *coeffs;
    int coeff_half_len, index_inc;
    long in_count, in_used, out_count, out_gen;
    int b_current, b_end, b_real_end, b_len;
    int front;           /* zero floats in front of buffer[0] (never written): see orc_src_new */
    float *buffer_base;  /* the allocation; buffer = buffer_base + front */
    float *buffer;
    /* channels > 1, sinc: one mono state per channel (the library's multi-channel loops do the same arithmetic per
     * channel on interleaved data; the end-of-input rule follows the mono loop).  Converters 3 / 4: src_zoh.c / src_linear.c. */
    int channels;
    struct orc_src_state **sub;
    float *last_value;
    int zl_reset;
};

/* src_set_ratio (samplerate.rs:40; libsamplerate 0.1.8 samplerate.c): the next src_process starts AT this ratio instead of gliding to its
 * src_ratio from the previous one -- it overwrites last_ratio, nothing else */
int orc_src_set_ratio(orc_src_state *s, double new_ratio)
{
    if (!s) return SRC_ERR_BAD_STATE;
    if (new_ratio < (1.0 / SRC_MAX_RATIO) || new_ratio > (1.0 * SRC_MAX_RATIO)) return SRC_ERR_BAD_SRC_RATIO;
    s->last_ratio = new_ratio;
    if (s->sub) for (int c = 0; c < s->channels; ++c) s->sub[c]->last_ratio = new_ratio;
    return SRC_ERR_NO_ERROR;
}

int orc_src_reset(orc_src_state *s)
{
    if (!s) return SRC_ERR_BAD_STATE;
    s->last_ratio = 0.0;
    s->last_position = 0.0;
    if (s->sub) { for (int c = 0; c < s->channels; ++c) orc_src_reset(s->sub[c]); return 0; }
    if (s->type >= 3) { s->zl_reset = 1; memset(s->last_value, 0, (size_t)s->channels * sizeof(float)); return 0; } /* zoh_reset / linear_reset */
    s->b_current = s->b_end = 0;
    s->b_real_end = -1;
    memset(s->buffer, 0, (size_t)(s->b_len + 1) * sizeof(float));
    return 0;
}

orc_src_state *orc_src_new(int converter_type, int channels, int *error)
{
    if (error) *error = 0;
    if (channels < 1) { if (error) *error = SRC_ERR_BAD_CHANNEL_COUNT; return NULL; }
    if (converter_type == 3 || converter_type == 4) { /* SRC_ZERO_ORDER_HOLD, SRC_LINEAR (samplerate.rs:29-30) */
        orc_src_state *z = (orc_src_state *)calloc(1, sizeof(*z));
        z->type = converter_type;
        z->channels = channels;
        z->last_value = (float *)calloc((size_t)channels, sizeof(float));
        orc_src_reset(z);
        return z;
    }
    const src_table *t = get_table(converter_type);
    if (!t) { if (error) *error = SRC_ERR_BAD_CONVERTER; return NULL; }
    if (channels > 1) {
        orc_src_state *m = (orc_src_state *)calloc(1, sizeof(*m));
        m->type = converter_type;
        m->channels = channels;
        m->sub = (orc_src_state **)calloc((size_t)channels, sizeof(*m->sub));
        for (int c = 0; c < channels; ++c) m->sub[c] = orc_src_new(converter_type, 1, NULL);
        return m;
    }
    orc_src_state *s = (orc_src_state *)calloc(1, sizeof(*s));
    s->channels = 1;
    s->type = converter_type;
    s->coeffs = t->c;
    s->coeff_half_len = t->n - 2;
    s->index_inc = t->increment;
    long bl = lrint(2.5 * s->coeff_half_len / (s->index_inc * 1.0) * SRC_MAX_RATIO);
    if (bl < 4096) bl = 4096;
    s->b_len = (int)bl;
    /* DEFINED where the published code is not: when the ratio DECREASES between two calls the filter widens (half_filter_chan_len follows
     * min(last_ratio, src_ratio)) while b_current still sits where the narrower filter left it, so the left wing's data_index =
     * b_current - coeff_count -- and prepare_data's memmove source b_current - half -- can be NEGATIVE: libsamplerate 0.1.8 then reads the
     * words in front of buffer[] (the filter struct's own fields).  Here the samples in front of the buffer are +0.0f (silence before the
     * stream, which is what those positions hold until the first move): the buffer is allocated with half_max zero floats in front of
     * index 0 that nothing ever writes.  The device images carry the same pad (src_host.hip). */
    s->front = (int)lrint((s->coeff_half_len + 2.0) / s->index_inc * SRC_MAX_RATIO) + 64;
    s->buffer_base = (float *)calloc((size_t)s->front + (size_t)s->b_len + 1, sizeof(float));
    s->buffer = s->buffer_base + s->front;
    orc_src_reset(s);
    return s;
}

void orc_src_delete(orc_src_state *s)
{
    if (!s) return;
    if (s->sub) { for (int c = 0; c < s->channels; ++c) orc_src_delete(s->sub[c]); free(s->sub); }
    free(s->last_value);
    free(s->buffer_base);
    free(s);
}

static double fmod_one(double x)
{
    double res = x - (double)lrint(x);
    if (res < 0.0) return res + 1.0;
    return res;
}

static int is_bad_src_ratio(double ratio)
{
    return (ratio < (1.0 / SRC_MAX_RATIO) || ratio > (1.0 * SRC_MAX_RATIO));
}

/* prepare_data: refill the linear buffer, keeping half_len samples of history before b_current */
static int prepare_data(orc_src_state *f, const orc_src_data *d, int half)
{
    int len;
    if (f->b_real_end >= 0) return 0;
    if (f->b_current == 0) {
        len = f->b_len - 2 * half;
        f->b_current = f->b_end = half;
    } else if (f->b_end + half + 1 < f->b_len) {
        len = f->b_len - f->b_current - half;
        if (len < 0) len = 0;
    } else {
        len = f->b_end - f->b_current;
        memmove(f->buffer, f->buffer + f->b_current - half, (size_t)(half + len) * sizeof(float));
        f->b_current = half;
        f->b_end = f->b_current + len;
        len = f->b_len - f->b_current - half;
        if (len < 0) len = 0;
    }
    long avail = f->in_count - f->in_used;
    if (avail < len) len = (int)avail;
    if (len < 0 || f->b_end + len > f->b_len) return SRC_ERR_SINC_PREPARE_DATA_BAD_LEN;
    memcpy(f->buffer + f->b_end, d->data_in + f->in_used, (size_t)len * sizeof(float));
    f->b_end += len;
    f->in_used += len;
    if (f->in_used == f->in_count && f->b_end - f->b_current < 2 * half && d->end_of_input) {
        /* last buffer: pad with zeros so the tail can be flushed */
        if (f->b_len - f->b_end < half + 5) {
            len = f->b_end - f->b_current;
            /* DEFINED where the published code is not (include/samplerate.h (iv)): below a ratio of about 1 / 213 this move can be longer
             * than the buffer (len < 2 half, b_len = 2.5 half at the smallest ratio); 0.1.8 writes past its allocation and then
             * zero-fills a negative length.  Here, and in the device path, the call fails with the library's bad-length code. */
            if (half + len > f->b_len) return SRC_ERR_SINC_PREPARE_DATA_BAD_LEN;
            memmove(f->buffer, f->buffer + f->b_current - half, (size_t)(half + len) * sizeof(float));
            f->b_current = half;
            f->b_end = f->b_current + len;
        }
        f->b_real_end = f->b_end;
        len = half + 5;
        if (len < 0 || f->b_end + len > f->b_len) len = f->b_len - f->b_end;
        memset(f->buffer + f->b_end, 0, (size_t)len * sizeof(float));
        f->b_end += len;
    }
    return 0;
}

/* calc_output_single: left wing walks the table down from its far end towards index start, the
 * data forwards up to b_current; right wing likewise from the other side, excluding index 0. */
static double calc_output(const orc_src_state *f, int32_t increment, int32_t start_filter_index)
{
    const int32_t max_filter_index = (int32_t)f->coeff_half_len << SHIFT_BITS;
    int32_t filter_index = start_filter_index;
    int coeff_count = (max_filter_index - filter_index) / increment;
    filter_index = filter_index + coeff_count * increment;
    int data_index = f->b_current - coeff_count;
    double left = 0.0;
    do {
        double fraction = (double)(filter_index & ((1 << SHIFT_BITS) - 1)) * INV_FP_ONE;
        int indx = filter_index >> SHIFT_BITS;
        double icoeff = f->coeffs[indx] + fraction * (f->coeffs[indx + 1] - f->coeffs[indx]);
        left += icoeff * f->buffer[data_index];
        filter_index -= increment;
        data_index = data_index + 1;
    } while (filter_index >= 0);

    filter_index = increment - start_filter_index;
    coeff_count = (max_filter_index - filter_index) / increment;
    filter_index = filter_index + coeff_count * increment;
    data_index = f->b_current + 1 + coeff_count;
    double right = 0.0;
    do {
        double fraction = (double)(filter_index & ((1 << SHIFT_BITS) - 1)) * INV_FP_ONE;
        int indx = filter_index >> SHIFT_BITS;
        double icoeff = f->coeffs[indx] + fraction * (f->coeffs[indx + 1] - f->coeffs[indx]);
        right += icoeff * f->buffer[data_index];
        filter_index -= increment;
        data_index = data_index - 1;
    } while (filter_index > 0);
    return left + right;
}

/* zoh_vari_process (src_zoh.c) and linear_vari_process (src_linear.c) of the published libsamplerate 0.1.8, interleaved
 * channels.  ZOH repeats the sample before the output instant; linear interpolates between the two samples around it:
 * (float)(a + input_index * (b - a)) with a, b float (their difference is a float) and input_index double. */
static int zoh_linear_process(orc_src_state *p, orc_src_data *d)
{
    const int ch_n = p->channels, lin = p->type == 4;
    if (d->input_frames <= 0) return SRC_ERR_NO_ERROR;
    if (p->zl_reset) { /* just reset: the value "before" the stream is its first frame */
        for (int ch = 0; ch < ch_n; ++ch) p->last_value[ch] = d->data_in[ch];
        p->zl_reset = 0;
    }
    const long in_count = d->input_frames * ch_n, out_count = d->output_frames * ch_n;
    long in_used = 0, out_gen = 0;
    double src_ratio = p->last_ratio, input_index = p->last_position, rem;
    /* samples before the first sample of this input array */
    while (input_index < 1.0 && out_gen < out_count) {
        if (lin ? (in_used + ch_n * (1.0 + input_index) >= in_count) : (in_used + ch_n * input_index >= in_count)) break;
        if (out_count > 0 && fabs(p->last_ratio - d->src_ratio) > 1e-20)
            src_ratio = p->last_ratio + out_gen * (d->src_ratio - p->last_ratio) / out_count;
        for (int ch = 0; ch < ch_n; ++ch) {
            d->data_out[out_gen] = lin ? (float)(p->last_value[ch] + input_index * (d->data_in[ch] - p->last_value[ch])) : p->last_value[ch];
            out_gen++;
        }
        input_index += 1.0 / src_ratio;
    }
    rem = fmod_one(input_index);
    in_used += ch_n * lrint(input_index - rem);
    input_index = rem;
    /* main loop */
    while (out_gen < out_count && (lin ? (in_used + ch_n * input_index < in_count) : (in_used + ch_n * input_index <= in_count))) {
        if (out_count > 0 && fabs(p->last_ratio - d->src_ratio) > 1e-20)
            src_ratio = p->last_ratio + out_gen * (d->src_ratio - p->last_ratio) / out_count;
        for (int ch = 0; ch < ch_n; ++ch) {
            /* in_used == 0 can reach this loop when the message holds a single frame (the loop above breaks at once): the
             * published code then reads data_in[-channels + ch], before the array.  Defined here -- and in the device path --
             * as the value carried from the previous message, which is what that address would hold in a contiguous stream. */
            const float a = in_used >= ch_n ? d->data_in[in_used - ch_n + ch] : p->last_value[ch];
            d->data_out[out_gen] = lin ? (float)(a + input_index * (d->data_in[in_used + ch] - a)) : a;
            out_gen++;
        }
        input_index += 1.0 / src_ratio;
        rem = fmod_one(input_index);
        in_used += ch_n * lrint(input_index - rem);
        input_index = rem;
    }
    if (in_used > in_count) {
        input_index += (in_used - in_count) / ch_n;
        in_used = in_count;
    }
    p->last_position = input_index;
    if (in_used > 0)
        for (int ch = 0; ch < ch_n; ++ch) p->last_value[ch] = d->data_in[in_used - ch_n + ch];
    p->last_ratio = src_ratio;
    d->input_frames_used = in_used / ch_n;
    d->output_frames_gen = out_gen / ch_n;
    return SRC_ERR_NO_ERROR;
}

/* src_process (samplerate.c) + sinc_mono_vari_process (src_sinc.c) */
int orc_src_process(orc_src_state *f, orc_src_data *d)
{
    if (!f) return SRC_ERR_BAD_STATE;
    if (!d) return SRC_ERR_BAD_DATA;
    if (!d->data_in || !d->data_out) return SRC_ERR_BAD_DATA_PTR;
    if (is_bad_src_ratio(d->src_ratio)) return SRC_ERR_BAD_SRC_RATIO;
    if (d->input_frames < 0) d->input_frames = 0;
    if (d->output_frames < 0) d->output_frames = 0;
    const int nch = f->channels > 0 ? f->channels : 1;
    if (d->data_in < d->data_out) {
        if (d->data_in + d->input_frames * nch > d->data_out) return SRC_ERR_DATA_OVERLAP;
    } else if (d->data_out + d->output_frames * nch > d->data_in) {
        return SRC_ERR_DATA_OVERLAP;
    }
    d->input_frames_used = 0;
    d->output_frames_gen = 0;
    if (f->last_ratio < (1.0 / SRC_MAX_RATIO)) f->last_ratio = d->src_ratio;
    if (f->type >= 3) return zoh_linear_process(f, d);
    if (f->sub) { /* interleaved channels: every channel through its own mono state, same arguments */
        float *in = (float *)malloc((size_t)(d->input_frames + 1) * sizeof(float));
        float *out = (float *)malloc((size_t)(d->output_frames + 1) * sizeof(float));
        int err = 0;
        for (int c = 0; c < nch && !err; ++c) {
            for (long i = 0; i < d->input_frames; ++i) in[i] = d->data_in[i * nch + c];
            orc_src_data m = *d;
            m.data_in = in; m.data_out = out;
            err = orc_src_process(f->sub[c], &m);
            for (long i = 0; i < m.output_frames_gen; ++i) d->data_out[i * nch + c] = out[i];
            d->input_frames_used = m.input_frames_used;
            d->output_frames_gen = m.output_frames_gen;
        }
        free(in); free(out);
        f->last_ratio = f->sub[0]->last_ratio;
        return err;
    }

    f->in_count = d->input_frames;
    f->out_count = d->output_frames;
    f->in_used = f->out_gen = 0;
    double src_ratio = f->last_ratio;
    if (is_bad_src_ratio(src_ratio)) return SRC_ERR_BAD_INTERNAL_STATE;

    double count = (f->coeff_half_len + 2.0) / f->index_inc;
    double minr = f->last_ratio < d->src_ratio ? f->last_ratio : d->src_ratio;
    if (minr < 1.0) count /= minr;
    int half = (int)lrint(count) + 1; /* half_filter_chan_len, mono */

    double input_index = f->last_position;
    double float_increment = f->index_inc;
    double rem = fmod_one(input_index);
    f->b_current = (f->b_current + (int)lrint(input_index - rem)) % f->b_len;
    input_index = rem;
    double terminate = 1.0 / src_ratio + 1e-20;

    while (f->out_gen < f->out_count) {
        int samples_in_hand = (f->b_end - f->b_current + f->b_len) % f->b_len;
        if (samples_in_hand <= half) {
            int err = prepare_data(f, d, half);
            if (err) return err;
            samples_in_hand = (f->b_end - f->b_current + f->b_len) % f->b_len;
            if (samples_in_hand <= half) break;
        }
        if (f->b_real_end >= 0) {
            if (f->b_current + input_index + terminate > f->b_real_end) break;
        }
        if (f->out_count > 0 && fabs(f->last_ratio - d->src_ratio) > 1e-10)
            src_ratio = f->last_ratio + f->out_gen * (d->src_ratio - f->last_ratio) / f->out_count;
        float_increment = f->index_inc * (src_ratio < 1.0 ? src_ratio : 1.0);
        int32_t increment = (int32_t)lrint(float_increment * FP_ONE);
        int32_t start_filter_index = (int32_t)lrint(input_index * float_increment * FP_ONE);
        d->data_out[f->out_gen] =
            (float)((float_increment / f->index_inc) * calc_output(f, increment, start_filter_index));
        f->out_gen++;
        input_index += 1.0 / src_ratio;
        rem = fmod_one(input_index);
        f->b_current = (f->b_current + (int)lrint(input_index - rem)) % f->b_len;
        input_index = rem;
    }
    f->last_position = input_index;
    f->last_ratio = src_ratio;
    d->input_frames_used = f->in_used;
    d->output_frames_gen = f->out_gen;
    return SRC_ERR_NO_ERROR;
}

/* one message through samplerate::resample, src/samplerate/src/samplerate.rs:62-86:
 * lout = ((ratio * len) + 1) as usize (:64); end_of_input = 0 (:73); the block sends
 * output_frames_gen samples (:84) and never looks at input_frames_used (:71). */
long orc_resample_block(orc_src_state *s, const float *in, long len, double ratio, float *out, long cap)
{
    long lout = (long)((ratio * (double)len) + 1.0);
    if (lout > cap) return -1;
    orc_src_data d;
    d.data_in = in;
    d.data_out = out;
    d.input_frames = len;
    d.output_frames = lout;
    d.input_frames_used = 0;
    d.output_frames_gen = 0;
    d.end_of_input = 0;
    d.src_ratio = ratio;
    int err = orc_src_process(s, &d);
    if (err) return -(long)err - 1000;
    return d.output_frames_gen;
}
