// misc_kernels.hip -- synthetic IQ generator: the same integer hash as the oracle, so
// host and device inputs are bit-identical without a transfer.
#include "redio_internal.h"

namespace redio {

__global__ __launch_bounds__(256) void synth_iq_kernel(float2 *out, uint32_t seed, uint64_t first, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t s = first + (uint64_t)i;
        out[i] = make_float2(unit_from_hash(hash32(seed, 2 * s)), unit_from_hash(hash32(seed, 2 * s + 1)));
    }
}

__global__ __launch_bounds__(256) void synth_f32_kernel(float *out, uint32_t seed, uint64_t first, long n)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        out[i] = unit_from_hash(hash32(seed, first + (uint64_t)i));
}

// hardware probe behind tests/test_gpu_channelizer.py::test_buffer_range_check_covers_the_vector_offset_only: one raw buffer load of a dword
// per lane from a descriptor of num_records bytes, with the offset split between the vector and the scalar operand as the caller says
__global__ void buffer_load_probe_kernel(const uint32_t *base, uint32_t num_records, uint32_t voffset, uint32_t soffset, uint32_t *out)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(base), 0, (int)num_records, 0x00020000);
    out[threadIdx.x] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, voffset + 4u * threadIdx.x, soffset, 0);
}
hipError_t launch_buffer_load_probe(const void *base, uint32_t num_records, uint32_t voffset, uint32_t soffset, void *out, hipStream_t s)
{
    hipLaunchKernelGGL(buffer_load_probe_kernel, dim3(1), dim3(64), 0, s, (const uint32_t *)base, num_records, voffset, soffset, (uint32_t *)out);
    return hipGetLastError();
}

static unsigned grid_for(long n)
{
    long g = (n + 255) / 256;
    return (unsigned)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

hipError_t launch_synth_iq(float2 *out, uint32_t seed, uint64_t first, long n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(synth_iq_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, seed, first, n);
    return hipGetLastError();
}

hipError_t launch_synth_f32(float *out, uint32_t seed, uint64_t first, long n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(synth_f32_kernel, dim3(grid_for(n)), dim3(256), 0, s, out, seed, first, n);
    return hipGetLastError();
}

} // namespace redio
