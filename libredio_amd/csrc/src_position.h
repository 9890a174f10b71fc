// src_position.h -- the resampler's position recurrence (host side of src_host.hip; CPU-tested through tests/emu)
// libsamplerate 0.1.8, sinc_mono_vari_process: after every output
//     input_index += 1.0 / src_ratio;  rem = fmod_one (input_index);  b_current += lrint (input_index - rem);  input_index = rem;
// with fmod_one (x) = x - lrint (x), plus 1.0 if that is negative.  One serial chain of doubles per OUTPUT (shared by all channels):
// add -> lrint -> int-to-double -> subtract -> compare -> add, about 20 cycles, which is what bounds rational upsampling.
#pragma once
#include <math.h>

namespace redio {

inline double src_fmod_one(double x)
{
    double res = x - (double)lrint(x);
    if (res < 0.0) return res + 1.0;
    return res;
}

// One step of the recurrence: x in [0, 1) on entry and on exit, returns the whole samples to advance by.
// step < 1 (every upsampling ratio): t = fl (x + step) lies in [0, 2) and the library's expression reduces EXACTLY to
//     t < 1:  rem = t, advance 0          t >= 1:  rem = t - 1, advance 1
// -- t < 0.5: lrint (t) = 0.  0.5 <= t < 1: lrint is 0 (t = 0.5, ties to even) or 1; then res = t - 1 is exact (Sterbenz), negative,
// and (t - 1) + 1 = t is representable, so the sum is exact.  1 <= t < 1.5: lrint = 1, res = t - 1 exact.  1.5 <= t < 2: lrint = 2,
// res = t - 2 exact and negative, (t - 2) + 1 = t - 1 representable (t has ulp 2^-52, t - 1 in [0.5, 1) has ulp 2^-53).  In each case
// input_index - rem is exactly 0 or 1.  The short form is add -> subtract / compare -> select: half the chain.
// (tests/test_emu_lane_programs.py checks it against the literal form on random and on edge operands.)
inline int src_advance(double &x, double step)
{
    double t = x + step;
    if (step < 1.0) {
        const double t1 = t - 1.0;
        const bool wrapped = t >= 1.0;
        x = wrapped ? t1 : t;
        return wrapped ? 1 : 0;
    }
    const double rem = src_fmod_one(t);
    const int adv = (int)lrint(t - rem);
    x = rem;
    return adv;
}

} // namespace redio
