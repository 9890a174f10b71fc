// taps.cpp -- host-side FIR tap generators of dsputils (src/dsputils/src/dsputils.rs:38-94).
// They run once per filter, so they stay on the host exactly as in the reference.  Built with
// -ffp-contract=off: every f32 operation below rounds once, as the Rust does.
//
// redio_window/sinc/lpf/hpf/bsf/bpf reproduce the reference AS WRITTEN, including its defects
// (SURVEY.md 0.6): window() returns m+1 values, its cosine argument is n/(nn-1) with n = m, the last
// term divides by cos(nn-1) instead of taking a cosine, and x == 1 yields NaN -- so lpf()[1] is NaN.
// redio_lpf_corrected is the windowed-sinc the reference evidently meant; benchmark chains use it.
#include "../../include/redio.h"
#include <math.h>
#include <vector>

static const float PI_F = 3.14159274101257324219f;                          // f32::consts::PI
static const float BN[4] = {0.3635819f, 0.4891775f, 0.1365995f, 0.0106411f}; // dsputils.rs:42

extern "C" int redio_window(size_t m, float *out)
{
    if (!out) return REDIO_ERR_ARG;
    const float n = (float)m;
    for (size_t x = 0; x <= m; ++x) {
        const float nn = (float)x;
        const float d = nn - 1.0f;
        const float c1 = cosf(2.0f * PI_F * n / d);
        const float c2 = cosf(4.0f * PI_F * n / d);
        const float q = 6.0f * PI_F * n / cosf(d); // dsputils.rs:49: .cos() binds to (nn-1) only
        out[x] = BN[0] - BN[1] * c1 + BN[2] * c2 - BN[3] * q;
    }
    return REDIO_OK;
}

extern "C" int redio_sinc(size_t m, float fc, float *out)
{
    if (!out) return REDIO_ERR_ARG;
    if (!(fc < 0.5f)) return REDIO_ERR_ASSERT; // assert!(fc < 0.5), dsputils.rs:55
    const float half = (float)m / 2.0f;
    for (size_t x = 0; x < m; ++x) {
        const float n = (float)x - half;
        float r = 2.0f * fc;
        if (n != 0.0f) r = sinf(2.0f * PI_F * fc * n) / (PI_F * n);
        out[x] = r;
    }
    return REDIO_OK;
}

extern "C" int redio_lpf(size_t m, float fc, float *out)
{
    if (!out) return REDIO_ERR_ARG;
    std::vector<float> w(m + 1), s(m ? m : 1);
    int rc = redio_sinc(m, fc, s.data());
    if (rc) return rc;
    redio_window(m, w.data());
    for (size_t x = 0; x < m; ++x) out[x] = w[x] * s[x]; // zip stops at the shorter (m) vector
    return REDIO_OK;
}

extern "C" int redio_hpf(size_t m, float fc, float *out)
{
    if (m < 2) return REDIO_ERR_ASSERT; // get_mut(m/2-1).unwrap() panics
    int rc = redio_lpf(m, fc, out);
    if (rc) return rc;
    for (size_t x = 0; x < m; ++x) out[x] = -out[x];
    out[m / 2 - 1] += 1.0f;
    return REDIO_OK;
}

extern "C" int redio_bsf(size_t m, float fc1, float fc2, float *out)
{
    if (m < 2) return REDIO_ERR_ASSERT;
    std::vector<float> lp(m), hp(m);
    int rc = redio_lpf(m, fc1, lp.data());
    if (rc == REDIO_OK) rc = redio_hpf(m, fc2, hp.data());
    if (rc) return rc;
    for (size_t x = 0; x < m; ++x) out[x] = lp[x] + hp[x];
    out[m / 2 - 1] -= 0.0f; // dsputils.rs:86
    return REDIO_OK;
}

extern "C" int redio_bpf(size_t m, float fc1, float fc2, float *out)
{
    int rc = redio_bsf(m, fc1, fc2, out);
    if (rc) return rc;
    for (size_t x = 0; x < m; ++x) out[x] = -out[x];
    return REDIO_OK;
}

extern "C" int redio_lpf_corrected(size_t m, float fc, float *out)
{
    if (!out || m == 0) return REDIO_ERR_ARG;
    if (!(fc < 0.5f)) return REDIO_ERR_ASSERT;
    const double pi = 3.14159265358979323846;
    const double a0 = 0.3635819, a1 = 0.4891775, a2 = 0.1365995, a3 = 0.0106411;
    const double c = ((double)m - 1.0) / 2.0;
    for (size_t x = 0; x < m; ++x) {
        double w = 1.0;
        if (m > 1) {
            const double ph = (double)x / ((double)m - 1.0);
            w = a0 - a1 * cos(2.0 * pi * ph) + a2 * cos(4.0 * pi * ph) - a3 * cos(6.0 * pi * ph);
        }
        const double n = (double)x - c;
        const double s = (n == 0.0) ? 2.0 * (double)fc : sin(2.0 * pi * (double)fc * n) / (pi * n);
        out[x] = (float)(w * s);
    }
    return REDIO_OK;
}
