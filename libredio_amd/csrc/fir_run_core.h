// fir_run_core.h -- geometry of the wave-private "run" form of the FIR on REAL samples (round 6): the data path of the chain kernel
// (chain_v4.hip: one wavefront owns a run of consecutive sub-tiles, the part of the window two consecutive sub-tiles share is carried
// inside its LDS image, the next sub-tile's samples are prefetched into registers during the multiply-adds, no workgroup barrier) with
// the scalar lane program of fir_core.h (fir_lane<float, K, D, R>), for dsputils::convolve's own sample type (f32, dsputils.rs:30-32) --
// BASELINE.json configs[0]'s 63-tap low-pass is this shape.  The tiled kernel spends a third to a half of its wave cycles in its
// load phase and the barrier behind it (profiles/r05_fir_mid_shapes.txt); this form has neither.
//
// A sub-tile = 64 lanes x R kept outputs = SUB_OUT outputs from SUB_NEW = SUB_OUT * D new samples.  Its image holds input samples
// [A, A + SUB_NEW + HALO_A), A = (index of the sub-tile) * SUB_NEW: the windows touch the first SUB_NEW + K - D of them; HALO_A is K - D
// rounded UP to a multiple of 4, so that the next sub-tile's new samples [A + SUB_NEW + HALO_A, ...) start on a 16-byte boundary and
// arrive as whole 16-byte loads (the 0-3 samples behind the last window ride along).  The last HALO_A samples of an image are the first
// HALO_A of the next: they move inside LDS, one dword per lane.  Image sample n sits at FirGeom<K, D, R>::lds_index(n) (odd lane stride
// in dwords: the 32 lanes of a ds_read_b32 hit 32 banks).  Host-compilable: tests/emu runs the wave program lane by lane.
#pragma once
#include "fir_core.h"

namespace redio {

template <int K, int D, int R>
struct FirRunReal {
    using G = FirGeom<K, D, R>;
    static constexpr int SUB_OUT = 64 * R;
    static constexpr int SUB_NEW = SUB_OUT * D;
    static constexpr int HALO = K - D;                     // samples a sub-tile's windows share with the next one
    static constexpr int HALO_A = (HALO + 3) & ~3;         // carried: a multiple of 4
    static constexpr int IMG = SUB_NEW + HALO_A;           // samples in the image
    static constexpr int NLD = SUB_NEW / 4 / 64;           // 16-byte loads per lane and sub-tile
    static constexpr int NHV = (HALO_A + 63) / 64;         // dwords per lane of the carried part
    static_assert(K >= D && SUB_NEW % 256 == 0 && HALO_A / 4 <= 64, "geometry");
    RD_HD static constexpr int lds_floats() { return G::lds_index(IMG - 1) + 1; }
    // image sample of component e of this lane's i-th 16-byte load of new samples / of the head
    RD_HD static constexpr int new_sample(int lane, int i, int e) { return HALO_A + 4 * (lane + 64 * i) + e; }
    RD_HD static constexpr int head_sample(int lane, int e) { return 4 * lane + e; }
    // sub-tiles a call of n_in input samples can run in this form (the rest goes to the tiled kernel)
    RD_HD static long whole_subtiles(long n_in, long n_out)
    {
        if (n_in < HALO_A + SUB_NEW) return 0;
        const long a = (n_in - HALO_A) / SUB_NEW, b = n_out / SUB_OUT;
        return a < b ? a : b;
    }
};

} // namespace redio
