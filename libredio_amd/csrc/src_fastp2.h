// src_fastp2.h -- REDIO_SRC_FAST, round 5: the phase-split f32 polyphase decimator of src_kernels.hip (src_window_fastp_kernel,
// same arithmetic, same image layout, same tap table) with the loader taken out from between the barriers.
// Included by src_kernels.hip inside namespace redio (needs SrcWindow, src_v2f, add_rn, RD_SCHED_BARRIER).
//
// What the round-3 kernel pays per tile beside its multiply-adds (profiles/r04_c3_counters.txt: the vector units issue in 52 % of the
// cycles): after the arithmetic every wavefront writes its partial sums, BARRIER, the sums are reduced and stored, the 64 samples a
// thread requested before the arithmetic are stored into the image (all eight wavefronts at once: an LDS-bound burst), BARRIER, and
// every wavefront fills its 50-register window from the new image (another LDS-bound burst) before the first multiply-add.
//
// Here the image is used as two halves BY PHASE (a phase's rows belong to ONE wavefront while it is computed, and the wavefronts take
// their phases in ascending order): with the wavefront steps a0 .. a(nA-1) on the phases below PA and b0 .. on the phases from PA,
//     a0   multiply-adds on rows A(t)   | parks B(t) <- request registers pfB (requested behind a0 of tile t-1)
//          then reduces and stores the sums of tile t-1 and requests B(t+1) into pfB
//     -- barrier: B(t) complete --
//     a1 .. the last A step refills its window registers with the first B step's samples as they fall free
//     -- barrier: A rows free --
//     b0   multiply-adds on rows B(t)   | parks A(t+1) <- request registers pfA (requested behind b0 of tile t-1); then requests A(t+2) into pfA
//     -- barrier: A(t+1) complete --
//     b1 .. the last B step refills its window registers with a0's samples of tile t+1
//     partial sums -> red
//     -- barrier: B rows free, sums complete --
// so the image stores ride inside the unrolled multiply-add stream (one ds_write_b32 and one address add per tap pair), no wavefront
// ever fills its window in a burst, and nothing but the partial-sum write sits between the arithmetic of two tiles.  Four barriers
// per tile instead of two, each at a point all wavefronts reach together (they run the same instruction count per step).
// Loader mapping per half: thread -> (phase of the half, group), consecutive threads on consecutive phases: runs of PA (or S - PA)
// contiguous floats, every 128-byte line is requested by both halves' loads half a tile apart (L2 / MALL hits; the kernel moves
// 4.08 B per input sample against 196 flop).
#pragma once

template <int NPAIR, int PFH_ = 32> // tap pairs per phase (table rows zero filled to whole chunks of 16 pairs); request registers per half
struct SrcFastP2 {
    static constexpr int R = 8, W = 8, NT = 64 * W, NO = 64 * R, NC = (NPAIR + 15) / 16, NTAP = 32 * NC, NE = NPAIR + R / 2, NI = 2 * NE;
    static constexpr int NGROUPS = 63 * R + NI;                          // groups of S samples a tile's image holds (sample n = g*S + p)
    static constexpr int PFH = PFH_;
    static constexpr int GI = (NGROUPS + PFH - 1) / PFH;                 // groups per loader round: PFH rounds cover the image, for BOTH halves
    static constexpr int NCOLW = 64 + (NI + R - 1) / R, NCOLL = (GI * PFH + 7) / 8; // columns the window reads / the loader's last round reaches
    static constexpr int NCOL = NCOLW > NCOLL ? NCOLW : NCOLL;
    static constexpr int ring(int ne) { int m = ne; for (int d = 26; d >= 22; --d) if (ne % d == 0) { m = d; break; } return m; }
    static constexpr int M = ring(NE);                                   // window ring: a divisor of NE that holds a chunk's 12 pairs and the next chunk's (22 at most)
    static int steps(int S) { return (S + W - 1) / W; }
    static int pa(int S) { return W * (steps(S) / 2); }                  // phases [0, PA): half A, [PA, S): half B
    static size_t lds_bytes(int S) { return ((size_t)R * S * NCOL + (size_t)W * NO) * sizeof(float); } // image, partial sums
    static bool fits(int S)
    {
        if (S < 25 || S > NT / 2) return false;                          // two full steps in each half: steps(S) >= 4
        const int PA = pa(S), PB = S - PA;
        return GI * (PA > PB ? PA : PB) <= NT && lds_bytes(S) <= 160 * 1024; // a round's GI groups x the half's phases: one sample per thread
    }
};

// One wavefront step: the NPAIR tap pairs of phase p (9 v_pk_fma_f32 per tap pair) in chunks of CH = 8 tap pairs on a RING of M window
// registers (window pair v of the stream [this phase's NE pairs | the next phase's ...] lives in slot v % M; M divides NE, so every phase
// finds its pair j in slot j % M).  All memory instructions of a chunk are issued TOGETHER at its head, for the chunk behind it: the
// scalar load of the next 8 tap pairs, the LDS reads of the window pairs the next chunk adds (`rowc`: this phase, `rown`: the phase this
// wavefront computes next), and -- MODE >= 1 -- the chunk's share of the request registers parked in the image, and -- MODE == 2 -- the
// NEXT tile's samples requested into the registers the chunk before parked.  Scalar loads return out of order, so the wait for the taps
// is a wait for EVERYTHING outstanding (lgkmcnt(0)): with the LDS instructions at the chunk head that wait, 72 multiply-adds later,
// finds them complete (issued one per tap pair, as round 3's kernel does, the most recent read is ~100 cycles short at every tap-chunk
// boundary).  The multiply-adds are written as instructions (accumulators in place, the tap a scalar register pair broadcast by op_sel):
// the compiler's own allocation renames every accumulator and needs 118 registers for this body, which leaves no room for two request sets.
#define SRC_PK_FMA_LO(acc, x, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(x), "s"(h))
#define SRC_PK_FMA_HI(acc, x, h) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(x), "s"(h))
struct SrcFastReq { __amdgpu_buffer_rsrc_t r; unsigned doff; }; // the descriptor of the tile a step requests, the byte advance of a round
template <int NPAIR, int M, int PFH, int GI, int MODE>
__device__ __forceinline__ void src_fastp2_step(src_v2f (&E)[M], src_v2f (&hc)[8], src_v2f (&accA)[4], src_v2f (&accB)[5],
                                                const src_v2f *hp, const src_v2f *hpn, const unsigned (&rowc)[4], const unsigned (&rown)[4],
                                                float (&pf)[PFH], const unsigned (&a8)[8], const SrcFastReq &rq, unsigned off, char *lds)
{
    constexpr int CH = 8, NC = (NPAIR + CH - 1) / CH, NE = NPAIR + 4, SPC = (PFH + NC - 1) / NC; // SPC request registers parked per chunk
    static_assert(NE % M == 0, "ring size");
    // distinct markers keep the compiler from merging the common code of instantiations that sit in the arms of one branch
    if (MODE == 2) asm volatile("; fastp2 step, parking and requesting" ::: "memory");
    else if (MODE == 1) asm volatile("; fastp2 step, parking" ::: "memory");
    else asm volatile("; fastp2 step" ::: "memory");
    auto request_share = [&](int c) { // chunk c's share of the request registers <- the next tile (they were parked one chunk ago)
#pragma unroll
        for (int u = c * SPC; u < (c + 1) * SPC && u < PFH; ++u) {
            pf[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rq.r, off, 0, 0));
            off += rq.doff;
            asm volatile("" : "+v"(off)); // ONE running offset register
        }
    };
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        // highest window pair chunk x reads: its last tap pair + 4; resident on entry to chunk c: everything up to top(c)
        constexpr auto top = [](int x) { const int last = CH * x + CH - 1 < NPAIR - 1 ? CH * x + CH - 1 : NPAIR - 1; return last + 4; };
        const int v_lo = top(c) + 1, v_hi = c + 1 < NC ? top(c + 1) : NE + top(0);
        static_assert(NE + top(0) - CH * (NC - 1) + 1 <= M && top(1) + 1 <= M, "the ring holds a chunk's window and the next chunk's additions");
        src_v2f hn[CH];
        const src_v2f *nextc = c + 1 < NC ? hp + CH * (c + 1) : hpn; // (the table rows are zero filled to whole chunks of 16 pairs)
#pragma unroll
        for (int m = 0; m < CH; ++m) hn[m] = nextc[m];
        // (written as instructions: the compiler pairs two of these reads into one ds_read2_b64 whose four result registers it then
        // copies into the ring slots -- 17 copies and their temporaries per step; they complete before the chunk-end wait below)
#pragma unroll
        for (int v = v_lo; v <= v_hi; ++v) {
            if (v < NE) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(E[v % M]) : "v"(rowc[v % 4]), "n"(8 * (v / 4)));
            else asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(E[v % M]) : "v"(rown[(v - NE) % 4]), "n"(8 * ((v - NE) / 4)));
        }
        if (MODE >= 1) {
#pragma unroll
            for (int u = c * SPC; u < (c + 1) * SPC && u < PFH; ++u) // round u adds the constant 8*GI*(u/8) bytes (an immediate)
                *reinterpret_cast<float *>(lds + (a8[u % 8] + 8u * (unsigned)GI * (unsigned)(u / 8))) = pf[u];
        }
        if (MODE == 2 && c > 0) request_share(c - 1);
        asm volatile("" ::: "memory"); // the memory instructions stay at the head of the chunk
        RD_SCHED_BARRIER();
#pragma unroll
        for (int m = 0; m < CH; ++m) {
            const int a0i = CH * c + m;
            if (a0i >= NPAIR) continue;
#pragma unroll
            for (int ca = 0; ca < 4; ++ca) SRC_PK_FMA_LO(accA[ca], E[(ca + a0i) % M], hc[m]);
#pragma unroll
            for (int cb = 0; cb <= 4; ++cb) SRC_PK_FMA_HI(accB[cb], E[(cb + a0i) % M], hc[m]);
        }
        RD_SCHED_BARRIER();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the next chunk's taps and window pairs (the reads above are not the compiler's to count)
#pragma unroll
        for (int m = 0; m < CH; ++m) { asm volatile("" : "+s"(hn[m])); hc[m] = hn[m]; }
    }
    if (MODE == 2) request_share(NC - 1);
}

// The barriers of the tile loop order LDS traffic only (image rows, partial sums).  __syncthreads() also waits for every outstanding
// GLOBAL access (vmcnt(0): the workgroup-scope fence of the memory model) -- here the requests in flight, i.e. a full HBM latency at
// every barrier (measured: 0.94 against 0.87 ms).  This one waits for the LDS instructions and nothing else.
__device__ __forceinline__ void src_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NPAIR, int PFH_, int ABL = 0> // ABL: timing-only ablations of tools/fastp_lab.hip (wrong results; 1 no requests, 8 no parking, 2 no barriers inside the tile, 4 no output stores, 16 every request reads the first tile), 0 in the product
__global__ __launch_bounds__(512) void src_window_fastp2_kernel(SrcWindow w, const float *__restrict__ Hp, int KH, int cl, long a0, int S,
                                                                float *__restrict__ out, long out_stride, long nout, int TPW)
{
    using G = SrcFastP2<NPAIR, PFH_>;
    constexpr int R = G::R, W = G::W, PFH = G::PFH, NO = G::NO, NTAP = G::NTAP, M = G::M, NCOL = G::NCOL, GI = G::GI;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *xs = reinterpret_cast<float *>(smem);   // [4*S rows][NCOL] cells of two floats
    const int tid = threadIdx.x, ch = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    float *red = xs + R * S * NCOL;                // [W][NO] partial sums (reduced behind a0 of the next tile, rewritten three barriers later)
    const src_v2f *xs2 = reinterpret_cast<const src_v2f *>(xs);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)smem; // LDS byte address of the image
    const long ntiles = (nout + NO - 1) / NO;
    const long t0 = (long)blockIdx.x * TPW;
    const int ntile = (int)(ntiles - t0 < TPW ? ntiles - t0 : TPW);
    if (ntile <= 0) return;
    const int nsteps = (S + W - 1) / W, nA = nsteps / 2, PA = W * nA;
    // Two sets of request registers, one per half, each requested again inside the step that parks it (one chunk behind) and parked one
    // whole tile later: every workgroup of the launch runs the same instruction stream in step, so a half's requests of all 256 CUs
    // arrive in one burst (30 MB) that HBM needs ~6 us to serve -- with ONE set (parked two steps behind its request) every parking step
    // waited ~2 us, and the 32 requests issued back to back behind the step held every wavefront ~1 us at the texture addresser.
    float pfA[PFH], pfB[PFH];

    // Loader mapping of a half with nph phases from phase P0: GI groups per round (a constant: PFH rounds cover the image), thread ->
    // (phase pl, group g0) with consecutive threads on consecutive phases; the threads past GI*nph repeat the work of thread
    // tid % (GI*nph) (same sample, same cell, same value): no lane is ever masked.
    struct Map { unsigned g0, pl; };
    auto mapping = [&](int P0, int nph) {
        Map mp;
        const int t = tid % (GI * nph);
        mp.g0 = (unsigned)(t / nph);
        mp.pl = (unsigned)(P0 + t - (int)mp.g0 * nph);
        return mp;
    };
    const Map mapA = mapping(0, PA), mapB = mapping(PA, S - PA); // four registers kept across the kernel
    // The window of a tile: round u of a request reads the tile samples (g0 + GI*u)*S + pl, i.e. one running byte offset per lane and ONE
    // buffer descriptor whose range check returns +0.0f at or behind `need` (samples of no valid output, which may not exist).  The one
    // tile of a call that straddles [old image | new input] cannot be read through one descriptor: it is requested in the prologue, from both
    // sources (request_now), and pays the wait.
    struct Tile { const float *src_old, *src_new; int need, nsplit; };
    auto tile_window = [&](long tile) {
        Tile t;
        const long k0 = tile * NO, tile_base = a0 + (long)S * k0 - cl;
        const long nvalid = (nout - k0 < NO) ? nout - k0 : NO;
        t.need = nvalid > 0 ? (int)((nvalid - 1) * S) + KH : 0; // (a tile past the last one: nothing)
        const long ns = w.a_in0 - tile_base;
        t.nsplit = ns < 0 ? 0 : (ns < t.need ? (int)ns : t.need); // [0, nsplit): old image, [nsplit, need): new input
        t.src_old = w.old_img + (long)ch * w.old_stride + tile_base;
        t.src_new = w.input + (long)ch * w.in_stride + (tile_base - w.a_in0); // tile sample 0 (valid from nsplit on)
        return t;
    };
    // The descriptor a step requests a tile through: the NEW input, the whole range.  The launcher guarantees that every tile requested inside the steps
    // (B of the workgroup's second tile, everything of its later tiles) lies in the new input: the old buffer image reaches two filter half-lengths into a
    // call's window, a tile is 512 x S samples, so only the call's FIRST tile straddles [old image | new input] -- and a workgroup's first tile (and the A
    // half of its second) is requested in the prologue, from both sources.  (A call whose second tile would straddle runs round 3's kernel:
    // src_kernels.hip.)  No branch around a request: with the straddling tile requested in a branch of its own the compiler copied both register sets
    // at the loop's back edge, and the copies waited for every request in flight.
    auto descriptor = [&](const Tile &t) {
        SrcFastReq rq;
        rq.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(t.src_new), 0, 4 * t.need, 0x00020000);
        rq.doff = 4u * (unsigned)(GI * S);
        return rq;
    };
    auto offset0 = [&](const Map &mp) { return 4u * (mp.g0 * (unsigned)S + mp.pl); };
    auto request_now = [&](long tile, const Map &mp, float(&pf)[PFH]) { // outside the steps: the prologue, the straddling tile
        const Tile t = tile_window(tile);
        unsigned off = offset0(mp);
        const unsigned doff = 4u * (unsigned)(GI * S);
        const auto r_old = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(t.src_old), 0, t.nsplit * 4, 0x00020000);
        const auto r_new = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(t.src_new + t.nsplit), 0, (t.need - t.nsplit) * 4, 0x00020000);
#pragma unroll
        for (int u = 0; u < PFH; ++u) { // each source is out of its range (+0.0f) where the other is in
            pf[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r_old, off, 0, 0) | __builtin_amdgcn_raw_buffer_load_b32(r_new, off - 4u * (unsigned)t.nsplit, 0, 0));
            off += doff;
            asm volatile("" : "+v"(off));
        }
    };
    // image addresses of a thread's requests: group g = 8*q + k of phase pl -> byte 8*(((k/2)*S + pl)*NCOL + q) + 4*(k%2); g advances by GI
    // per round, so k returns after eight rounds with q advanced by GI: eight addresses, then 8*GI bytes (an immediate) per eight rounds
    auto addresses = [&](unsigned(&a8)[8], const Map &mp) {
        const unsigned rowbytes = 8u * (unsigned)(S * NCOL), base = 8u * mp.pl * (unsigned)NCOL;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned g = mp.g0 + (unsigned)(GI * u);
            a8[u] = __umul24((g >> 1) & 3u, rowbytes) + base + ((g >> 3) << 3) + ((g & 1u) << 2);
        }
    };
    auto reduce_store = [&](long tile) { // the eight wavefronts' partial sums of `tile` -> out
        const long k0 = tile * NO;
        const long nvalid = (nout - k0 < NO) ? nout - k0 : NO;
        float sum = red[tid];
#pragma unroll
        for (int q = 1; q < W; ++q) sum = add_rn(sum, red[q * NO + tid]);
        if (tid < nvalid && !(ABL & 4)) out[(long)ch * out_stride + k0 + tid] = sum;
    };

    src_v2f E[M], hc[8], accA[R / 2], accB[R / 2 + 1];
    auto fill_window = [&](int p) { // the window of phase p's first chunk (12 pairs) in one burst (prologue; wavefronts whose last step is b0)
#pragma unroll
        for (int a = 0; a < 12; ++a) E[a] = xs2[((a % (R / 2)) * S + p) * NCOL + lane + a / (R / 2)];
    };

    // ---- prologue: half A of the first tile requested and parked, A of the second and B of the first requested, the first window filled
    unsigned a8[8];
    request_now(t0, mapA, pfA);
    {
        addresses(a8, mapA);
#pragma unroll
        for (int u = 0; u < PFH; ++u) *reinterpret_cast<float *>(smem + (a8[u % 8] + 8u * (unsigned)GI * (unsigned)(u / 8))) = pfA[u];
        const src_v2f *hp = reinterpret_cast<const src_v2f *>(Hp + (long)wave * NTAP);
#pragma unroll
        for (int m = 0; m < 8; ++m) hc[m] = hp[m];
    }
    request_now(t0, mapB, pfB);
    request_now(t0 + 1, mapA, pfA);
    __syncthreads();
    fill_window(wave);
#pragma unroll
    for (int m = 0; m < 8; ++m) asm volatile("" : "+s"(hc[m]));

    // One straight-line tile body (no branch decides whether the request registers are written or read: the compiler otherwise copies
    // them at the joins, and a copy waits for the loads it copies).  The workgroup's last tiles request and park tiles that do not exist:
    // their descriptors have an empty range (+0.0f), nobody computes on those rows.
    auto run_step = [&](int k, auto mode, float(&pf)[PFH], const SrcFastReq &rq, unsigned off) {
        const int p = wave + W * k;
        const bool next_valid = p + W < S;
        // the phase this wavefront computes next: p + W, or its first phase of the next tile (rows A(t+1): complete from the barrier
        // behind b0 on, so a wavefront whose LAST step is b0 refills with its own rows -- unused -- and fills in a burst behind that barrier)
        const bool wrap_in_b0 = !next_valid && k == nA;
        const int pn = next_valid ? p + W : (wrap_in_b0 ? p : wave), pn_taps = next_valid ? p + W : wave;
        unsigned rowc[R / 2], rown[R / 2]; // LDS byte addresses of this lane's first cell in the four rows of a phase
#pragma unroll
        for (int kk = 0; kk < R / 2; ++kk) { rowc[kk] = lds0 + 8u * (unsigned)((kk * S + p) * NCOL + lane); rown[kk] = lds0 + 8u * (unsigned)((kk * S + pn) * NCOL + lane); }
        const src_v2f *hp = reinterpret_cast<const src_v2f *>(Hp + (long)p * NTAP), *hpn = reinterpret_cast<const src_v2f *>(Hp + (long)pn_taps * NTAP);
        src_fastp2_step<NPAIR, M, PFH, GI, decltype(mode)::value>(E, hc, accA, accB, hp, hpn, rowc, rown, pf, a8, rq, off, smem);
    };
    using Plain = std::integral_constant<int, 0>;
    using ParkRequest = std::integral_constant<int, (ABL & 8) ? 0 : ((ABL & 1) ? 1 : 2)>;
    const SrcFastReq none = {};
    for (int ti = 0; ti < ntile; ++ti) {
#pragma unroll
        for (int c = 0; c < R / 2; ++c) accA[c] = src_v2f{0.f, 0.f};
#pragma unroll
        for (int c = 0; c <= R / 2; ++c) accB[c] = src_v2f{0.f, 0.f};
        // a0: parks B(t) and requests B(t+1) into the same registers, one chunk behind; then the sums of tile t-1 leave
        addresses(a8, mapB);
        run_step(0, ParkRequest{}, pfB, descriptor(tile_window((ABL & 16) ? t0 : t0 + ti + 1)), offset0(mapB));
        if (ti > 0) reduce_store(t0 + ti - 1);
        if (!(ABL & 2)) src_lds_barrier(); // B(t) complete
        for (int k = 1; k < nA; ++k) run_step(k, Plain{}, pfA, none, 0u);
        if (!(ABL & 2)) src_lds_barrier(); // A rows free
        // b0: parks A(t+1) and requests A(t+2)
        addresses(a8, mapA);
        run_step(nA, ParkRequest{}, pfA, descriptor(tile_window((ABL & 16) ? t0 : t0 + ti + 2)), offset0(mapA));
        if (!(ABL & 2)) src_lds_barrier(); // A(t+1) complete
        if (wave + W * (nA + 1) >= S) fill_window(wave); // a wavefront whose last step is b0
        for (int k = nA + 1; k < nsteps; ++k)
            if (wave + W * k < S) run_step(k, Plain{}, pfA, none, 0u);
        float *myred = red + wave * NO + lane * R;
#pragma unroll
        for (int c = 0; c < R / 2; ++c) {
            myred[2 * c] = accA[c].x + accB[c].y;
            myred[2 * c + 1] = accA[c].y + accB[c + 1].x;
        }
        src_lds_barrier(); // B rows free; the partial sums of this tile are complete
    }
    reduce_store(t0 + ntile - 1);
}
