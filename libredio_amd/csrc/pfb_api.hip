// pfb_api.hip -- C ABI of the polyphase channelizer plan (include/redio.h, redio_pfb_*).
#include "../../include/redio.h"
#include "pfb_core.h"
#include "redio_internal.h"
#include <new>
#include <string.h>
#include <vector>

namespace redio {
bool pfb_supported(int nchan, int taps_per_branch);
hipError_t launch_pfb(const float2 *x, const float *h, const float2 *tw64, float2 *out, long rows, int taps_per_branch,
                      int ngroups, bool fused, hipStream_t s);
} // namespace redio
using namespace redio;

static inline int hip_rc(hipError_t e) { return e == hipSuccess ? REDIO_OK : REDIO_ERR_HIP_BASE - (int)e; }

struct redio_pfb {
    int device, nchan, taps_per_branch;
    unsigned flags;
    float *d_h;
    redio_fft *fft; // owns the 64-entry twiddle table in the published kissfft form
    float2 *d_tw;
};

extern "C" int redio_pfb_create(redio_pfb **h, const float *proto, int nchan, int taps_per_branch, unsigned flags)
{
    if (!h) return REDIO_ERR_ARG;
    *h = nullptr;
    if (!proto || nchan <= 0 || taps_per_branch <= 0) return REDIO_ERR_ARG;
    if (!pfb_supported(nchan, taps_per_branch)) return REDIO_ERR_UNSUPPORTED;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return REDIO_ERR_NO_DEVICE;
    redio_pfb *p = new (std::nothrow) redio_pfb();
    if (!p) return REDIO_ERR_NOMEM;
    p->device = dev; p->nchan = nchan; p->taps_per_branch = taps_per_branch; p->flags = flags; p->d_h = nullptr; p->d_tw = nullptr;
    const size_t nt = (size_t)nchan * taps_per_branch;
    std::vector<float2> tw((size_t)nchan);
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    for (int i = 0; i < nchan; ++i) {
        const double phase = -2 * pi * i / nchan;
        tw[(size_t)i] = make_float2((float)cos(phase), (float)sin(phase));
    }
    hipError_t e = hipMalloc((void **)&p->d_h, nt * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(p->d_h, proto, nt * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&p->d_tw, (size_t)nchan * sizeof(float2));
    if (e == hipSuccess) e = hipMemcpy(p->d_tw, tw.data(), (size_t)nchan * sizeof(float2), hipMemcpyHostToDevice);
    if (e != hipSuccess) { hipFree(p->d_h); hipFree(p->d_tw); delete p; return hip_rc(e); }
    *h = p;
    return REDIO_OK;
}

extern "C" int redio_pfb_destroy(redio_pfb *h)
{
    if (!h) return REDIO_OK;
    hipFree(h->d_h);
    hipFree(h->d_tw);
    delete h;
    return REDIO_OK;
}

extern "C" size_t redio_pfb_nrows(const redio_pfb *h, size_t n_in)
{
    if (!h) return 0;
    const size_t T = n_in / (size_t)h->nchan;
    return T < (size_t)h->taps_per_branch ? 0 : T - (size_t)h->taps_per_branch + 1;
}

extern "C" int redio_pfb_enqueue(redio_pfb *h, const void *d_in, size_t n_in, void *d_out, int ngroups, void *stream)
{
    if (!h) return REDIO_ERR_ARG;
    const size_t rows = redio_pfb_nrows(h, n_in);
    if (rows == 0) return REDIO_OK;
    if (!d_in || !d_out || d_in == d_out) return REDIO_ERR_ARG;
    if (ngroups < 1 || h->nchan % ngroups) return REDIO_ERR_ARG;
    hipError_t e = hipSetDevice(h->device);
    if (e != hipSuccess) return hip_rc(e);
    e = launch_pfb((const float2 *)d_in, h->d_h, h->d_tw, (float2 *)d_out, (long)rows, h->taps_per_branch, ngroups,
                   (h->flags & REDIO_FIR_FUSED) != 0, (hipStream_t)stream);
    if (e == hipErrorNotSupported) return REDIO_ERR_UNSUPPORTED;
    return hip_rc(e);
}
