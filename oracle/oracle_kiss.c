/*
 * oracle_kiss.c -- CPU restatement of the kissfft complex FFT that src/kissfft/src/kissfft.rs:11-31
 * binds (kiss_fft_alloc / kiss_fft / kiss_fft_cleanup).  TEST INFRASTRUCTURE ONLY.
 *
 * The C source is NOT in /root/reference (empty submodule, .gitmodules:1-3).  This file restates the
 * published algorithm of kissfft 1.3.0 kiss_fft.c (float scalar build): the factoriser that pulls
 * 4s, then 2s, then 3, 5, 7, ... (kf_factor), a twiddle table tw[i] = (float)cos/sin(-/+ 2 pi i/n)
 * evaluated in double (kiss_fft_alloc), recursive decimation in time (kf_work) and the radix-2/3/4/5
 * and generic butterflies with their published operation order (kf_bfly2/3/4/5/_generic).  The
 * operation order matters: the GPU kernels reproduce it so that results are bit-identical.
 * Parity unpinned: no kissfft vectors exist in the reference; pinned by numpy float64 FFT checks.
 *
 * Build with -ffp-contract=off (kissfft built with plain cc on x86-64 has no FMA contraction).
 */
#include "redio_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXSTAGES 32

struct orc_kiss_state {
    int nfft;
    int inverse;
    int nstages;
    int radix[ORC_MAXSTAGES];  /* p of each stage, outermost first */
    int sublen[ORC_MAXSTAGES]; /* m of each stage */
    orc_cpx *tw;               /* nfft twiddles */
};

/* kf_factor: n = p0*m0, m0 = p1*m1, ...; tries 4, then 2, then 3, 5, 7, ...; a candidate above
 * floor(sqrt(n)) (n = the ORIGINAL length) is replaced by the remaining n itself. */
int orc_kiss_factors(int nfft, int *facbuf)
{
    int n = nfft, p = 4, cnt = 0;
    double floor_sqrt = floor(sqrt((double)n));
    do {
        while (n % p) {
            switch (p) {
            case 4: p = 2; break;
            case 2: p = 3; break;
            default: p += 2; break;
            }
            if (p > floor_sqrt) p = n;
        }
        n /= p;
        facbuf[2 * cnt] = p;
        facbuf[2 * cnt + 1] = n;
        ++cnt;
    } while (n > 1 && cnt < ORC_MAXSTAGES);
    return cnt;
}

orc_kiss_state *orc_kiss_fft_alloc(int nfft, int inverse)
{
    if (nfft <= 0) return NULL;
    orc_kiss_state *st = (orc_kiss_state *)calloc(1, sizeof(*st));
    st->nfft = nfft;
    st->inverse = inverse ? 1 : 0;
    st->tw = (orc_cpx *)malloc((size_t)nfft * sizeof(orc_cpx));
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    for (int i = 0; i < nfft; ++i) {
        double phase = -2 * pi * i / nfft;
        if (st->inverse) phase *= -1;
        st->tw[i].r = (float)cos(phase);
        st->tw[i].i = (float)sin(phase);
    }
    int fac[2 * ORC_MAXSTAGES];
    st->nstages = orc_kiss_factors(nfft, fac);
    for (int s = 0; s < st->nstages; ++s) {
        st->radix[s] = fac[2 * s];
        st->sublen[s] = fac[2 * s + 1];
    }
    return st;
}

void orc_kiss_fft_free(orc_kiss_state *st)
{
    if (!st) return;
    free(st->tw);
    free(st);
}

/* complex helpers in the published macro order: C_MUL = (ar*br - ai*bi, ar*bi + ai*br) */
static inline orc_cpx cmul(orc_cpx a, orc_cpx b)
{
    orc_cpx m;
    m.r = a.r * b.r - a.i * b.i;
    m.i = a.r * b.i + a.i * b.r;
    return m;
}
static inline orc_cpx cadd(orc_cpx a, orc_cpx b) { orc_cpx c = {a.r + b.r, a.i + b.i}; return c; }
static inline orc_cpx csub(orc_cpx a, orc_cpx b) { orc_cpx c = {a.r - b.r, a.i - b.i}; return c; }

/* kf_bfly2 */
static void bfly2(orc_cpx *F, size_t fstride, const orc_kiss_state *st, int m)
{
    const orc_cpx *tw1 = st->tw;
    orc_cpx *F2 = F + m;
    for (int k = 0; k < m; ++k) {
        orc_cpx t = cmul(F2[k], *tw1);
        tw1 += fstride;
        F2[k] = csub(F[k], t);
        F[k] = cadd(F[k], t);
    }
}

/* kf_bfly4 */
static void bfly4(orc_cpx *F, size_t fstride, const orc_kiss_state *st, size_t m)
{
    const orc_cpx *tw1 = st->tw, *tw2 = st->tw, *tw3 = st->tw;
    const size_t m2 = 2 * m, m3 = 3 * m;
    for (size_t k = 0; k < m; ++k) {
        orc_cpx s0 = cmul(F[k + m], *tw1);
        orc_cpx s1 = cmul(F[k + m2], *tw2);
        orc_cpx s2 = cmul(F[k + m3], *tw3);
        orc_cpx s5 = csub(F[k], s1);
        F[k] = cadd(F[k], s1);
        orc_cpx s3 = cadd(s0, s2);
        orc_cpx s4 = csub(s0, s2);
        F[k + m2] = csub(F[k], s3);
        tw1 += fstride;
        tw2 += fstride * 2;
        tw3 += fstride * 3;
        F[k] = cadd(F[k], s3);
        if (st->inverse) {
            F[k + m].r = s5.r - s4.i;
            F[k + m].i = s5.i + s4.r;
            F[k + m3].r = s5.r + s4.i;
            F[k + m3].i = s5.i - s4.r;
        } else {
            F[k + m].r = s5.r + s4.i;
            F[k + m].i = s5.i - s4.r;
            F[k + m3].r = s5.r - s4.i;
            F[k + m3].i = s5.i + s4.r;
        }
    }
}

/* kf_bfly3 */
static void bfly3(orc_cpx *F, size_t fstride, const orc_kiss_state *st, size_t m)
{
    const size_t m2 = 2 * m;
    const orc_cpx *tw1 = st->tw, *tw2 = st->tw;
    const orc_cpx epi3 = st->tw[fstride * m];
    for (size_t k = 0; k < m; ++k) {
        orc_cpx s1 = cmul(F[k + m], *tw1);
        orc_cpx s2 = cmul(F[k + m2], *tw2);
        orc_cpx s3 = cadd(s1, s2);
        orc_cpx s0 = csub(s1, s2);
        tw1 += fstride;
        tw2 += fstride * 2;
        /* HALF_OF(x) is ((x)*.5) with a DOUBLE literal: the subtraction happens in double and is
         * rounded to float on assignment */
        F[k + m].r = (float)((double)F[k].r - (double)s3.r * .5);
        F[k + m].i = (float)((double)F[k].i - (double)s3.i * .5);
        s0.r *= epi3.i;
        s0.i *= epi3.i;
        F[k] = cadd(F[k], s3);
        F[k + m2].r = F[k + m].r + s0.i;
        F[k + m2].i = F[k + m].i - s0.r;
        F[k + m].r -= s0.i;
        F[k + m].i += s0.r;
    }
}

/* kf_bfly5 */
static void bfly5(orc_cpx *F, size_t fstride, const orc_kiss_state *st, int m)
{
    const orc_cpx *tw = st->tw;
    const orc_cpx ya = tw[fstride * m], yb = tw[fstride * 2 * m];
    orc_cpx *F0 = F, *F1 = F + m, *F2 = F + 2 * m, *F3 = F + 3 * m, *F4 = F + 4 * m;
    for (int u = 0; u < m; ++u) {
        orc_cpx s0 = F0[u];
        orc_cpx s1 = cmul(F1[u], tw[u * fstride]);
        orc_cpx s2 = cmul(F2[u], tw[2 * u * fstride]);
        orc_cpx s3 = cmul(F3[u], tw[3 * u * fstride]);
        orc_cpx s4 = cmul(F4[u], tw[4 * u * fstride]);
        orc_cpx s7 = cadd(s1, s4), s10 = csub(s1, s4);
        orc_cpx s8 = cadd(s2, s3), s9 = csub(s2, s3);
        F0[u].r += s7.r + s8.r;
        F0[u].i += s7.i + s8.i;
        orc_cpx s5, s6, s11, s12;
        s5.r = s0.r + s7.r * ya.r + s8.r * yb.r;
        s5.i = s0.i + s7.i * ya.r + s8.i * yb.r;
        s6.r = s10.i * ya.i + s9.i * yb.i;
        s6.i = -(s10.r * ya.i) - s9.r * yb.i;
        F1[u] = csub(s5, s6);
        F4[u] = cadd(s5, s6);
        s11.r = s0.r + s7.r * yb.r + s8.r * ya.r;
        s11.i = s0.i + s7.i * yb.r + s8.i * ya.r;
        s12.r = -(s10.i * yb.i) + s9.i * ya.i;
        s12.i = s10.r * yb.i - s9.r * ya.i;
        F2[u] = cadd(s11, s12);
        F3[u] = csub(s11, s12);
    }
}

/* kf_bfly_generic */
static void bfly_generic(orc_cpx *F, size_t fstride, const orc_kiss_state *st, int m, int p)
{
    const orc_cpx *tw = st->tw;
    const int norig = st->nfft;
    orc_cpx *scratch = (orc_cpx *)malloc((size_t)p * sizeof(orc_cpx));
    for (int u = 0; u < m; ++u) {
        int k = u;
        for (int q1 = 0; q1 < p; ++q1) {
            scratch[q1] = F[k];
            k += m;
        }
        k = u;
        for (int q1 = 0; q1 < p; ++q1) {
            int twidx = 0;
            F[k] = scratch[0];
            for (int q = 1; q < p; ++q) {
                twidx += (int)fstride * k;
                if (twidx >= norig) twidx -= norig;
                orc_cpx t = cmul(scratch[q], tw[twidx]);
                F[k] = cadd(F[k], t);
            }
            k += m;
        }
    }
    free(scratch);
}

/* kf_work: decimation in time.  Stage s has radix p and sub-length m; the p sub-transforms read
 * the input at offsets q*fstride with stride fstride*p, then one butterfly pass combines them. */
static void work(orc_cpx *F, const orc_cpx *f, size_t fstride, int stage, const orc_kiss_state *st)
{
    const int p = st->radix[stage], m = st->sublen[stage];
    if (m == 1) {
        for (int q = 0; q < p; ++q) F[q] = f[(size_t)q * fstride];
    } else {
        for (int q = 0; q < p; ++q) work(F + (size_t)q * m, f + (size_t)q * fstride, fstride * p, stage + 1, st);
    }
    switch (p) {
    case 2: bfly2(F, fstride, st, m); break;
    case 3: bfly3(F, fstride, st, (size_t)m); break;
    case 4: bfly4(F, fstride, st, (size_t)m); break;
    case 5: bfly5(F, fstride, st, m); break;
    default: bfly_generic(F, fstride, st, m, p); break;
    }
}

/* kiss_fft (= kiss_fft_stride with in_stride 1): in-place goes through a temporary */
void orc_kiss_fft(const orc_kiss_state *st, const orc_cpx *fin, orc_cpx *fout)
{
    if (fin == fout) {
        orc_cpx *tmp = (orc_cpx *)malloc((size_t)st->nfft * sizeof(orc_cpx));
        work(tmp, fin, 1, 0, st);
        memcpy(fout, tmp, (size_t)st->nfft * sizeof(orc_cpx));
        free(tmp);
    } else {
        work(fout, fin, 1, 0, st);
    }
}

/* the block contract of kissfft::fft (src/kissfft/src/kissfft.rs:18-31): one cfg for the life of
 * the block (:19), every message exactly nfft samples (:24), a fresh output buffer per message. */
void orc_fft_blocks(int nfft, int inverse, const orc_cpx *in, orc_cpx *out, size_t nblocks)
{
    orc_kiss_state *st = orc_kiss_fft_alloc(nfft, inverse);
    if (!st) return;
    for (size_t b = 0; b < nblocks; ++b) orc_kiss_fft(st, in + b * (size_t)nfft, out + b * (size_t)nfft);
    orc_kiss_fft_free(st);
}
