// fir_tile.h -- cooperative global -> LDS tile load shared by the FIR and chain kernels (device only).
#pragma once
#include "fir_core.h"

namespace redio {

// ---- cooperative tile load: global (16 B per lane, coalesced) -> padded LDS image -------------
template <typename T, typename G, int NT, int TILE_IN>
__device__ __forceinline__ void load_tile(const T *__restrict__ x, long n_in, long in0, T *xs, bool vec_ok)
{
    constexpr int VEC = 16 / sizeof(T);
    constexpr int NV = (TILE_IN + VEC - 1) / VEC;
    const int tid = threadIdx.x;
    if (vec_ok) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x + in0);
#pragma unroll 4
        for (int v = tid; v < NV; v += NT) {
            const int n = v * VEC;
            if (in0 + n + VEC <= n_in) {
                float4 q = x4[v];
                if constexpr (sizeof(T) == 8) {
                    xs[G::lds_index(n)] = make_float2(q.x, q.y);
                    xs[G::lds_index(n + 1)] = make_float2(q.z, q.w);
                } else {
                    xs[G::lds_index(n)] = q.x;
                    xs[G::lds_index(n + 1)] = q.y;
                    xs[G::lds_index(n + 2)] = q.z;
                    xs[G::lds_index(n + 3)] = q.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    T val{};
                    if (in0 + n + e < n_in) val = x[in0 + n + e];
                    if (n + e < TILE_IN) xs[G::lds_index(n + e)] = val;
                }
            }
        }
    } else {
        for (int n = tid; n < TILE_IN; n += NT) {
            T val{};
            if (in0 + n < n_in) val = x[in0 + n];
            xs[G::lds_index(n)] = val;
        }
    }
}

} // namespace redio
