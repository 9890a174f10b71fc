/*
 * oracle_dsp.c -- CPU restatement of dsputils (FIR + tap generators), the synthetic-IQ hash and
 * the C2 chain.  TEST INFRASTRUCTURE ONLY (see redio_oracle.h).  Parity unpinned: the reference has
 * no tests or vectors for these functions; pinned by numpy float64 cross-checks + tests/golden/.
 *
 * Build with -ffp-contract=off: Rust never contracts a*b+c into an FMA, and the strict left fold of
 * src/dsputils/src/dsputils.rs:31 must keep one rounding per multiply and one per add.
 */
#include "redio_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- A1: convolve */

/* src/dsputils/src/dsputils.rs:30-32
 *   u.windows(v.len()).map(|x| x.iter().zip(v.iter()).map(|(&x,&y)| x*y).fold(zero, |a,b| a+b))
 * windows(K) yields nu-K+1 slices (none when nu<K; K==0 panics). */
size_t orc_convolve_f32(const float *u, size_t nu, const float *v, size_t nv, float *out)
{
    if (nv == 0) return (size_t)-1;
    if (nu < nv) return 0;
    size_t nout = nu - nv + 1;
    for (size_t i = 0; i < nout; ++i) {
        float a = 0.0f;
        for (size_t j = 0; j < nv; ++j) {
            float p = u[i + j] * v[j];
            a = a + p;
        }
        out[i] = a;
    }
    return nout;
}

size_t orc_convolve_f64(const double *u, size_t nu, const double *v, size_t nv, double *out)
{
    if (nv == 0) return (size_t)-1;
    if (nu < nv) return 0;
    size_t nout = nu - nv + 1;
    for (size_t i = 0; i < nout; ++i) {
        double a = 0.0;
        for (size_t j = 0; j < nv; ++j) {
            double p = u[i + j] * v[j];
            a = a + p;
        }
        out[i] = a;
    }
    return nout;
}

/* number of kept outputs of a valid-mode FIR with decimation: indices 0, D, 2D, ... <= n-k */
static size_t fir_nout(size_t n, size_t k, size_t decim)
{
    if (k == 0 || decim == 0) return (size_t)-1;
    if (n < k) return 0;
    return (n - k) / decim + 1;
}

/* same fold as :31, applied to re and im separately (complex x real taps), out[i] = conv[decim*i] */
size_t orc_fir_c32(const orc_cpx *x, size_t n, const float *taps, size_t k, size_t decim,
                   int fused, orc_cpx *out)
{
    size_t nout = fir_nout(n, k, decim);
    if (nout == (size_t)-1 || nout == 0) return nout;
    for (size_t i = 0; i < nout; ++i) {
        const orc_cpx *w = x + i * decim;
        float ar = 0.0f, ai = 0.0f;
        if (fused) {
            for (size_t j = 0; j < k; ++j) {
                ar = fmaf(w[j].r, taps[j], ar);
                ai = fmaf(w[j].i, taps[j], ai);
            }
        } else {
            for (size_t j = 0; j < k; ++j) {
                float pr = w[j].r * taps[j];
                float pi = w[j].i * taps[j];
                ar = ar + pr;
                ai = ai + pi;
            }
        }
        out[i].r = ar;
        out[i].i = ai;
    }
    return nout;
}

size_t orc_fir_f32(const float *x, size_t n, const float *taps, size_t k, size_t decim,
                   int fused, float *out)
{
    size_t nout = fir_nout(n, k, decim);
    if (nout == (size_t)-1 || nout == 0) return nout;
    for (size_t i = 0; i < nout; ++i) {
        const float *w = x + i * decim;
        float a = 0.0f;
        if (fused) {
            for (size_t j = 0; j < k; ++j) a = fmaf(w[j], taps[j], a);
        } else {
            for (size_t j = 0; j < k; ++j) {
                float p = w[j] * taps[j];
                a = a + p;
            }
        }
        out[i] = a;
    }
    return nout;
}

/* ---------------------------------------------------------------- A2-A4: tap generators */

static const float ORC_PI_F = 3.14159274101257324219f; /* f32::consts::PI */
/* blackman-nuttall coefficients, src/dsputils/src/dsputils.rs:42 */
static const float BN[4] = {0.3635819f, 0.4891775f, 0.1365995f, 0.0106411f};

/* src/dsputils/src/dsputils.rs:38-51 -- as written, quirks kept:
 *   m+1 values; argument is n/(nn-1) with n=m, nn=x; last term is a3*(6*pi*n / cos(nn-1)).
 *   x==1 gives n/0 = +inf, cos(inf) = NaN. */
void orc_window(size_t m, float *out)
{
    float n = (float)m;
    for (size_t x = 0; x <= m; ++x) {
        float nn = (float)x;
        float d = nn - 1.0f;
        float c1 = cosf(2.0f * ORC_PI_F * n / d);
        float c2 = cosf(4.0f * ORC_PI_F * n / d);
        float q = 6.0f * ORC_PI_F * n / cosf(d);
        float t1 = BN[1] * c1;
        float t2 = BN[2] * c2;
        float t3 = BN[3] * q;
        float s = BN[0] - t1;
        s = s + t2;
        s = s - t3;
        out[x] = s;
    }
}

/* src/dsputils/src/dsputils.rs:53-63 */
int orc_sinc(size_t m, float fc, float *out)
{
    if (!(fc < 0.5f)) return -1; /* assert!(fc < 0.5) :55 */
    float half = (float)m / 2.0f;
    for (size_t x = 0; x < m; ++x) {
        float n = (float)x - half;
        float r = 2.0f * fc;
        if (n != 0.0f) {
            float num = sinf(2.0f * ORC_PI_F * fc * n);
            float den = ORC_PI_F * n;
            r = num / den;
        }
        out[x] = r;
    }
    return 0;
}

/* src/dsputils/src/dsputils.rs:66-71 -- zip truncates the (m+1)-long window to m */
int orc_lpf(size_t m, float fc, float *out)
{
    float *w = (float *)malloc((m + 1) * sizeof(float));
    float *s = (float *)malloc((m ? m : 1) * sizeof(float));
    orc_window(m, w);
    int rc = orc_sinc(m, fc, s);
    if (rc == 0)
        for (size_t x = 0; x < m; ++x) out[x] = w[x] * s[x];
    free(w);
    free(s);
    return rc;
}

/* src/dsputils/src/dsputils.rs:74-79 -- negate, then +1.0 at index m/2-1 (integer division) */
int orc_hpf(size_t m, float fc, float *out)
{
    if (m < 2) return -1; /* m/2-1 underflows -> get_mut(None).unwrap() panics */
    int rc = orc_lpf(m, fc, out);
    if (rc) return rc;
    for (size_t x = 0; x < m; ++x) out[x] = -out[x];
    out[m / 2 - 1] += 1.0f;
    return 0;
}

/* src/dsputils/src/dsputils.rs:82-88 -- lpf(fc1)+hpf(fc2); the "-= 0.0" at :86 changes nothing */
int orc_bsf(size_t m, float fc1, float fc2, float *out)
{
    if (m < 2) return -1;
    float *lp = (float *)malloc(m * sizeof(float));
    float *hp = (float *)malloc(m * sizeof(float));
    int rc = orc_lpf(m, fc1, lp);
    if (rc == 0) rc = orc_hpf(m, fc2, hp);
    if (rc == 0) {
        for (size_t x = 0; x < m; ++x) out[x] = lp[x] + hp[x];
        out[m / 2 - 1] -= 0.0f;
    }
    free(lp);
    free(hp);
    return rc;
}

/* src/dsputils/src/dsputils.rs:91-94 */
int orc_bpf(size_t m, float fc1, float fc2, float *out)
{
    int rc = orc_bsf(m, fc1, fc2, out);
    if (rc == 0)
        for (size_t x = 0; x < m; ++x) out[x] = -out[x];
    return rc;
}

/* Corrected designer (documented deviation from :38-71): true Blackman-Nuttall window
 * w[x] = a0 - a1 cos(2 pi x/(m-1)) + a2 cos(4 pi x/(m-1)) - a3 cos(6 pi x/(m-1)) times a sinc that
 * is symmetric about (m-1)/2; evaluated in double, rounded once to f32. */
int orc_lpf_corrected(size_t m, float fc, float *out)
{
    if (!(fc < 0.5f) || m == 0) return -1;
    const double pi = 3.14159265358979323846;
    const double a0 = 0.3635819, a1 = 0.4891775, a2 = 0.1365995, a3 = 0.0106411;
    double c = ((double)m - 1.0) / 2.0;
    for (size_t x = 0; x < m; ++x) {
        double w = 1.0;
        if (m > 1) {
            double ph = (double)x / ((double)m - 1.0);
            w = a0 - a1 * cos(2.0 * pi * ph) + a2 * cos(4.0 * pi * ph) - a3 * cos(6.0 * pi * ph);
        }
        double n = (double)x - c;
        double s = (n == 0.0) ? 2.0 * (double)fc : sin(2.0 * pi * (double)fc * n) / (pi * n);
        out[x] = (float)(w * s);
    }
    return 0;
}

/* ---------------------------------------------------------------- synthetic IQ */

static inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}

uint32_t orc_hash32(uint32_t seed, uint64_t index)
{
    uint32_t lo = (uint32_t)index, hi = (uint32_t)(index >> 32);
    uint32_t h = fmix32(lo ^ seed);
    return fmix32(h ^ (hi * 0x9E3779B9u + 0x7F4A7C15u));
}

static inline float unit_from_hash(uint32_t h)
{
    return (float)(h >> 8) * 1.1920928955078125e-07f - 1.0f; /* (h>>8) * 2^-23 - 1, exact in f32 */
}

void orc_synth_iq(uint32_t seed, uint64_t first_sample, size_t n, orc_cpx *out)
{
    for (size_t i = 0; i < n; ++i) {
        uint64_t s = first_sample + i;
        out[i].r = unit_from_hash(orc_hash32(seed, 2 * s));
        out[i].i = unit_from_hash(orc_hash32(seed, 2 * s + 1));
    }
}

void orc_synth_f32(uint32_t seed, uint64_t first_sample, size_t n, float *out)
{
    for (size_t i = 0; i < n; ++i) out[i] = unit_from_hash(orc_hash32(seed, first_sample + i));
}

/* ---------------------------------------------------------------- C2 chain */

/* FIR decimate over the whole buffer (valid mode), then forward FFT over consecutive full blocks of
 * the decimated stream; a trailing partial block is dropped, as kpn::shaper would (kpn.rs:278-282). */
size_t orc_chain_fir_fft(const orc_cpx *x, size_t n, const float *taps, size_t k, size_t decim,
                         int nfft, int fused, orc_cpx *out)
{
    size_t ny = fir_nout(n, k, decim);
    if (ny == (size_t)-1 || ny == 0 || nfft <= 0) return 0;
    size_t nblk = ny / (size_t)nfft;
    if (nblk == 0) return 0;
    size_t need = (nblk * (size_t)nfft - 1) * decim + k; /* inputs that feed the kept blocks */
    orc_cpx *y = (orc_cpx *)malloc(nblk * (size_t)nfft * sizeof(orc_cpx));
    orc_fir_c32(x, need, taps, k, decim, fused, y);
    orc_fft_blocks(nfft, 0, y, out, nblk);
    free(y);
    return nblk;
}

/* ---------------------------------------------------------------- C4 polyphase channelizer */

/* M-channel critically-sampled analysis filterbank composed from the two reference primitives
 * (a new composition): with rows x_t[m] = x[M t + m] and branch filters
 * g_m[p] = h[M p + m] (prototype h of M*P taps),
 *     v_t[m] = fold_{p=0..P-1} x_{t+p}[m] * g_m[p]      -- dsputils::convolve per branch (dsputils.rs:31)
 *     out_t  = kiss_fft_M(v_t)  forward                  -- kissfft::fft per row (kissfft.rs:26)
 * for t = 0 .. T-P where T = floor(n / M).  Returns the number of output rows. */
size_t orc_pfb_channelizer(const orc_cpx *x, size_t n, const float *h, int M, int P, int fused, orc_cpx *out)
{
    if (M <= 0 || P <= 0) return 0;
    size_t T = n / (size_t)M;
    if (T < (size_t)P) return 0;
    size_t rows = T - (size_t)P + 1;
    orc_kiss_state *st = orc_kiss_fft_alloc(M, 0);
    orc_cpx *v = (orc_cpx *)malloc((size_t)M * sizeof(orc_cpx));
    for (size_t t = 0; t < rows; ++t) {
        for (int m = 0; m < M; ++m) {
            float ar = 0.0f, ai = 0.0f;
            for (int p = 0; p < P; ++p) {
                const orc_cpx s = x[(size_t)M * (t + (size_t)p) + (size_t)m];
                const float g = h[(size_t)M * (size_t)p + (size_t)m];
                if (fused) {
                    ar = fmaf(s.r, g, ar);
                    ai = fmaf(s.i, g, ai);
                } else {
                    float pr = s.r * g, pi = s.i * g;
                    ar = ar + pr;
                    ai = ai + pi;
                }
            }
            v[m].r = ar;
            v[m].i = ai;
        }
        orc_kiss_fft(st, v, out + t * (size_t)M);
    }
    free(v);
    orc_kiss_fft_free(st);
    return rows;
}

/* ---------------------------------------------------------------- C5 overlap-save FFT convolution */

/* Valid-mode correlation y[i] = sum_j x[i+j] h[j] (the semantics of dsputils::convolve,
 * dsputils.rs:30-32, on cf32 with real taps) computed block-wise in the frequency domain with
 * kissfft: H = kiss_fft(h zero-padded to nfft); for every block b of nfft input samples starting at
 * b*hop (hop = nfft-k+1): X = kiss_fft(block), Y[q] = X[q] * conj(H[q]) (C_MUL order),
 * y = kiss_fft_inverse(Y), out[b*hop + i] = y[i] * (1.0f/nfft) for i < hop.  Only whole blocks are
 * produced (a trailing partial block is dropped, as kpn::shaper would).  Returns outputs written. */
size_t orc_overlap_save(const orc_cpx *x, size_t n, const float *h, size_t k, int nfft, orc_cpx *out)
{
    if (nfft <= 0 || k == 0 || k > (size_t)nfft || n < (size_t)nfft) return 0;
    const size_t N = (size_t)nfft, hop = N - k + 1;
    const size_t nblk = (n - N) / hop + 1;
    orc_kiss_state *fw = orc_kiss_fft_alloc(nfft, 0), *bw = orc_kiss_fft_alloc(nfft, 1);
    orc_cpx *hp = (orc_cpx *)calloc(N, sizeof(orc_cpx)), *H = (orc_cpx *)malloc(N * sizeof(orc_cpx));
    orc_cpx *X = (orc_cpx *)malloc(N * sizeof(orc_cpx)), *Y = (orc_cpx *)malloc(N * sizeof(orc_cpx));
    for (size_t j = 0; j < k; ++j) hp[j].r = h[j];
    orc_kiss_fft(fw, hp, H);
    for (size_t q = 0; q < N; ++q) H[q].i = -H[q].i; /* conj */
    const float scale = 1.0f / (float)nfft;
    for (size_t b = 0; b < nblk; ++b) {
        orc_kiss_fft(fw, x + b * hop, X);
        for (size_t q = 0; q < N; ++q) {
            Y[q].r = X[q].r * H[q].r - X[q].i * H[q].i;
            Y[q].i = X[q].r * H[q].i + X[q].i * H[q].r;
        }
        orc_kiss_fft(bw, Y, X);
        for (size_t i = 0; i < hop; ++i) {
            out[b * hop + i].r = X[i].r * scale;
            out[b * hop + i].i = X[i].i * scale;
        }
    }
    free(hp); free(H); free(X); free(Y);
    orc_kiss_fft_free(fw); orc_kiss_fft_free(bw);
    return nblk * hop;
}
