// stream_split.h -- how one call of a carried-history stream (stream_carry.hip) divides its work.  Pure host
// arithmetic, shared with tests/emu so that the segmentation logic is checked on the CPU against the oracle.
//
// The stream's tail of `hist` samples (hist < W) sits in a staging buffer; a call brings n new samples, of which the
// first min(n, W-1) are appended to the staging buffer.  Output unit u needs the W samples starting at u*H.
//   head: the nh units that START inside the tail, computed from the staging buffer (head_in samples of it);
//   body: the nb units that start in the new data, computed in place from new + off (body_in samples);
//   the new tail starts at (nh + nb)*H in [tail | new] coordinates.
#pragma once
#include <stddef.h>

namespace redio {

struct StreamSplit { size_t nh, head_in, nb, off, body_in; };

inline StreamSplit stream_split(size_t hist, size_t W, size_t H, size_t n)
{
    StreamSplit s = {0, 0, 0, 0, 0};
    const size_t m = n < W - 1 ? n : W - 1;
    const size_t avail = hist + m;
    if (hist > 0 && avail >= W) {
        const size_t by_start = (hist + H - 1) / H;  // units that start inside the tail
        const size_t by_data = (avail - W) / H + 1;  // units whose window the staging buffer holds
        s.nh = by_start < by_data ? by_start : by_data;
        s.head_in = (s.nh - 1) * H + W;
    }
    const size_t start = s.nh * H;                   // first unit the head does not serve, [tail | new] coordinates
    if (start >= hist) {
        s.off = start - hist;
        if (n >= s.off + W) {
            s.nb = (n - s.off - W) / H + 1;
            s.body_in = (s.nb - 1) * H + W;
        }
    }
    return s;
}

} // namespace redio
