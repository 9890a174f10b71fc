// emu_kernels.cpp -- CPU emulation of the lane programs in libredio_amd/csrc/{fir,fft}_core.h.
// There is no GPU in the build container, so the index maps (lane <-> sample, LDS padding, digit
// order of the three-pass 1024-point transform, generic mixed-radix stages) are exercised here with
// the very same headers, one lane at a time, against the oracle.  g++ -ffp-contract=off.
#include "../../libredio_amd/csrc/fft_core.h"
#include "../../libredio_amd/csrc/fir_core.h"
#include "../../libredio_amd/csrc/fir_run_core.h"
#include "../../libredio_amd/csrc/pfb_core.h"
#include <stdlib.h>
#include <string.h>
#include <vector>

using namespace redio;

static std::vector<float2> make_tw(int n, int inverse)
{
    std::vector<float2> tw((size_t)n);
    const double pi = 3.141592653589793238462643383279502884197169399375105820974944;
    for (int i = 0; i < n; ++i) {
        double phase = -2 * pi * i / n;
        if (inverse) phase *= -1;
        tw[i] = make_float2((float)cos(phase), (float)sin(phase));
    }
    return tw;
}

template <bool INV>
static void emu_fft1k_t(const float2 *in, float2 *out)
{
    std::vector<float2> tw = make_tw(1024, INV);
    std::vector<float2> ex(FFT1K_LDS), ex2(FFT1K_LDS);
    float2 v[16];
    for (int lane = 0; lane < 64; ++lane) { // pass A, every lane
        for (int t = 0; t < 16; ++t) v[t] = in[lane + 64 * t];
        fft1k_passA<INV>(v, tw.data());
        for (int k4 = 0; k4 < 4; ++k4)
            for (int k3 = 0; k3 < 4; ++k3) ex[fft1k_A_store(lane, k3, k4)] = v[k3 + 4 * k4];
    }
    for (int lane = 0; lane < 64; ++lane) { // pass B
        for (int e = 0; e < 16; ++e) v[e] = ex[fft1k_B_load(lane, e)];
        Fft1kTw t;
        fft1k_load_tw(t, lane, tw.data());
        fft1k_passB<INV>(v, t);
        for (int k2 = 0; k2 < 4; ++k2)
            for (int k1 = 0; k1 < 4; ++k1) ex2[fft1k_B_store(lane, k1, k2)] = v[k1 + 4 * k2];
    }
    for (int lane = 0; lane < 64; ++lane) { // pass C
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 4; ++j) v[4 * q + j] = ex2[fft1k_C_load(lane, q, j)];
        Fft1kTw t;
        fft1k_load_tw(t, lane, tw.data());
        fft1k_passC<INV>(v, t);
        for (int q = 0; q < 4; ++q)
            for (int j = 0; j < 4; ++j) out[lane + 64 * q + 256 * j] = v[4 * q + j];
    }
}

extern "C" void emu_fft1k(const float2 *in, float2 *out, int inverse)
{
    if (inverse) emu_fft1k_t<true>(in, out);
    else emu_fft1k_t<false>(in, out);
}

// the "native" transform of the fused chain: input already in the FIR register layout
template <bool INV>
static void emu_fft1kn_t(const float2 *in, float2 *out)
{
    std::vector<float2> tw = make_tw(1024, INV);
    std::vector<float2> x1(FFT1KN_LDS), x2(FFT1KN_LDS);
    float2 v[16];
    for (int lane = 0; lane < 64; ++lane) {
        for (int s = 0; s < 4; ++s)
            for (int r = 0; r < 4; ++r) v[4 * s + r] = in[256 * s + 4 * lane + r];
        fft1kn_stage0<INV>(v, tw.data());
        for (int k4 = 0; k4 < 4; ++k4)
            for (int d0 = 0; d0 < 4; ++d0) x1[fft1kn_x1_store(lane, k4, d0)] = v[4 * k4 + d0];
    }
    for (int lane = 0; lane < 64; ++lane) {
        for (int e = 0; e < 16; ++e) v[e] = x1[fft1kn_x1_load(lane, e)];
        Fft1knTw12 t;
        fft1kn_load_tw12(t, lane, tw.data());
        fft1kn_pass12<INV>(v, t);
        for (int k3 = 0; k3 < 4; ++k3)
            for (int k2 = 0; k2 < 4; ++k2) x2[fft1kn_x2_store(lane, k2, k3)] = v[k2 + 4 * k3];
    }
    for (int lane = 0; lane < 64; ++lane) {
        for (int f = 0; f < 16; ++f) v[f] = x2[fft1kn_x2_load(lane, f)];
        Fft1knTw34 t;
        fft1kn_load_tw34(t, lane, tw.data());
        fft1kn_pass34<INV>(v, t);
        for (int k0 = 0; k0 < 4; ++k0)
            for (int k1 = 0; k1 < 4; ++k1) out[lane + 64 * k1 + 256 * k0] = v[k1 + 4 * k0];
    }
}
extern "C" void emu_fft1kn(const float2 *in, float2 *out, int inverse)
{
    if (inverse) emu_fft1kn_t<true>(in, out);
    else emu_fft1kn_t<false>(in, out);
}
extern "C" int emu_fft1kn_bank_conflicts(void)
{
    int worst = 1;
    auto check = [&](int (*addr)(int, int), int nacc, int group) {
        for (int a = 0; a < nacc; ++a)
            for (int g0 = 0; g0 < 64; g0 += group) {
                int cnt[32] = {0};
                for (int l = g0; l < g0 + group; ++l) cnt[addr(l, a) % 32]++;
                for (int b = 0; b < 32; ++b) if (cnt[b] > worst) worst = cnt[b];
            }
    };
    check([](int l, int a) { return fft1kn_x1_store(l, a >> 2, a & 3); }, 16, 16);
    check([](int l, int a) { return fft1kn_x1_load(l, a); }, 16, 32);
    check([](int l, int a) { return fft1kn_x2_store(l, a & 3, a >> 2); }, 16, 16);
    check([](int l, int a) { return fft1kn_x2_load(l, a); }, 16, 32);
    return worst;
}

// LDS bank check of the 1024-point exchanges: returns the worst number of distinct addresses that
// share a bank inside one lane group (1 = conflict free).  group = 16 lanes for ds_write_b64,
// 32 lanes for ds_read_b64; bank of an 8-byte access = (dword address / 2) % 32 pairs.
extern "C" int emu_fft1k_bank_conflicts(void)
{
    int worst = 1;
    auto check = [&](int (*addr)(int, int), int nacc, int group) {
        for (int a = 0; a < nacc; ++a)
            for (int g0 = 0; g0 < 64; g0 += group) {
                int cnt[32] = {0};
                for (int l = g0; l < g0 + group; ++l) cnt[addr(l, a) % 32]++;
                for (int b = 0; b < 32; ++b) if (cnt[b] > worst) worst = cnt[b];
            }
    };
    check([](int l, int a) { return fft1k_A_store(l, a & 3, a >> 2); }, 16, 16);
    check([](int l, int a) { return fft1k_B_load(l, a); }, 16, 32);
    check([](int l, int a) { return fft1k_B_store(l, a & 3, a >> 2); }, 16, 16);
    check([](int l, int a) { return fft1k_C_load(l, a >> 2, a & 3); }, 16, 32);
    return worst;
}

extern "C" int emu_fft_generic(int n, int inverse, const float2 *in, float2 *out)
{
    FftStage st[32];
    int ns = fft_plan_stages(n, st, 32);
    if (ns < 0) return -1;
    std::vector<float2> tw = make_tw(n, inverse);
    std::vector<float2> A((size_t)n), B((size_t)n);
    for (int P = 0; P < n; ++P) A[P] = in[fft_leaf_source(P, st, ns)];
    for (int s = ns - 1; s >= 0; --s) {
        if (st[s].p <= 5) {
            for (int b = 0; b < n / st[s].p; ++b) {
                if (inverse) fft_stage_butterfly<true>(A.data(), tw.data(), st[s], b);
                else fft_stage_butterfly<false>(A.data(), tw.data(), st[s], b);
            }
        } else {
            const int pm = st[s].p * st[s].m;
            for (int e = 0; e < n; ++e) {
                const int g = e / pm, r = e - g * pm, q1 = r / st[s].m, u = r - q1 * st[s].m;
                B[e] = fft_generic_output(A.data(), tw.data(), st[s], n, g, u, q1);
            }
            A.swap(B);
        }
    }
    memcpy(out, A.data(), (size_t)n * sizeof(float2));
    return ns;
}

// one workgroup tile of the tiled FIR kernel: padded LDS image + fir_lane per thread
template <typename T, int K, int D, int R, bool FUSED>
static long emu_fir_tiles(const T *x, long n_in, const float *taps, T *y)
{
    using G = FirGeom<K, D, R>;
    constexpr int NT = 256, TILE_OUT = NT * R, TILE_IN = G::tile_in(TILE_OUT);
    const long n_out = n_in < K ? 0 : (n_in - K) / D + 1;
    std::vector<T> xs((size_t)G::lds_elems(TILE_OUT));
    for (long tile = 0; tile * TILE_OUT < n_out; ++tile) {
        const long in0 = tile * (long)TILE_OUT * D;
        for (int n = 0; n < TILE_IN; ++n) {
            T v{};
            if (in0 + n < n_in) v = x[in0 + n];
            xs[(size_t)G::lds_index(n)] = v;
        }
        for (int tid = 0; tid < NT; ++tid) {
            T acc[R];
            for (int r = 0; r < R; ++r) acc[r] = T{};
            fir_lane<T, K, D, R, FUSED>(xs.data(), tid, taps, acc);
            for (int r = 0; r < R; ++r) {
                const long o = tile * (long)TILE_OUT + (long)tid * R + r;
                if (o < n_out) y[o] = acc[r];
            }
        }
    }
    return n_out;
}

extern "C" long emu_fir_c32(const float2 *x, long n_in, const float *taps, int K, int D, int fused, float2 *y)
{
    if (K == 127 && D == 5) return fused ? emu_fir_tiles<float2, 127, 5, 4, true>(x, n_in, taps, y) : emu_fir_tiles<float2, 127, 5, 4, false>(x, n_in, taps, y);
    if (K == 127 && D == 1) return fused ? emu_fir_tiles<float2, 127, 1, 8, true>(x, n_in, taps, y) : emu_fir_tiles<float2, 127, 1, 8, false>(x, n_in, taps, y);
    if (K == 63 && D == 1) return fused ? emu_fir_tiles<float2, 63, 1, 8, true>(x, n_in, taps, y) : emu_fir_tiles<float2, 63, 1, 8, false>(x, n_in, taps, y);
    if (K == 63 && D == 5) return fused ? emu_fir_tiles<float2, 63, 5, 4, true>(x, n_in, taps, y) : emu_fir_tiles<float2, 63, 5, 4, false>(x, n_in, taps, y);
    return -1;
}

extern "C" long emu_fir_f32(const float *x, long n_in, const float *taps, int K, int D, int fused, float *y)
{
    if (K == 127 && D == 5) return fused ? emu_fir_tiles<float, 127, 5, 4, true>(x, n_in, taps, y) : emu_fir_tiles<float, 127, 5, 4, false>(x, n_in, taps, y);
    if (K == 63 && D == 1) return fused ? emu_fir_tiles<float, 63, 1, 8, true>(x, n_in, taps, y) : emu_fir_tiles<float, 63, 1, 8, false>(x, n_in, taps, y);
    return -1;
}

// the wave-private run form of the real-sample FIR (libredio_amd/csrc/fir_run_core.h; device side fir_run.hip): one "wavefront" per
// run of sub_per_wave sub-tiles, lane by lane -- head, parked new samples, the HALO_A samples carried from image to image -- on an LDS
// image that starts as NaN, so a window that reads a slot nobody wrote cannot go unnoticed.  Returns the outputs produced
// (whole sub-tiles only; the device runs the remainder on the tiled kernel).
template <int K, int D, int R, bool FUSED>
static long emu_fir_run_t(const float *x, long n_in, const float *taps, float *y, long sub_per_wave)
{
    using U = FirRunReal<K, D, R>;
    using G = typename U::G;
    const long n_out = n_in < K ? 0 : (n_in - K) / D + 1;
    const long nsub = U::whole_subtiles(n_in, n_out);
    for (long s0 = 0; s0 < nsub; s0 += sub_per_wave) {
        const long s1 = std::min(nsub, s0 + sub_per_wave), n = s1 - s0;
        std::vector<float> xs((size_t)U::lds_floats(), std::nanf(""));
        const float *src0 = x + s0 * U::SUB_NEW;
        auto park = [&](long j) { // every lane's NLD 16-byte loads of sub-tile j's new samples
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < U::NLD; ++i)
                    for (int e = 0; e < 4; ++e) {
                        const long g = U::HALO_A + j * U::SUB_NEW + 4 * (lane + 64 * i) + e; // run-relative input sample
                        if (s0 * U::SUB_NEW + g >= n_in) return -1;                          // the launcher must never ask for this
                        xs[(size_t)G::lds_index(U::new_sample(lane, i, e))] = src0[g];
                    }
            return 0;
        };
        for (int lane = 0; lane < U::HALO_A / 4; ++lane)
            for (int e = 0; e < 4; ++e) xs[(size_t)G::lds_index(U::head_sample(lane, e))] = src0[4 * lane + e];
        if (park(0)) return -1;
        for (long j = 0; j < n; ++j) {
            for (int lane = 0; lane < 64; ++lane) {
                float acc[R];
                for (int r = 0; r < R; ++r) acc[r] = 0.f;
                fir_lane<float, K, D, R, FUSED>(xs.data(), lane, taps, acc);
                for (int r = 0; r < R; ++r) y[(s0 + j) * U::SUB_OUT + (long)lane * R + r] = acc[r];
            }
            if (j + 1 < n) {
                std::vector<float> halo((size_t)U::HALO_A);
                for (int k = 0; k < U::HALO_A; ++k) halo[(size_t)k] = xs[(size_t)G::lds_index(U::SUB_NEW + k)];
                std::fill(xs.begin(), xs.end(), std::nanf("")); // nothing else of the old image may survive into the next
                for (int k = 0; k < U::HALO_A; ++k) xs[(size_t)G::lds_index(k)] = halo[(size_t)k];
                if (park(j + 1)) return -1;
            }
        }
    }
    return nsub * U::SUB_OUT;
}
extern "C" long emu_fir_run_f32(const float *x, long n_in, const float *taps, int K, int D, int fused, float *y, long sub_per_wave)
{
    if (K == 63 && D == 1) return fused ? emu_fir_run_t<63, 1, 8, true>(x, n_in, taps, y, sub_per_wave) : emu_fir_run_t<63, 1, 8, false>(x, n_in, taps, y, sub_per_wave);
    return -2;
}

// the pair-image tile of the real-sample FIR (fir_core.h FirGeomPairs / fir_lane_pairs; device side fir_pairs_kernel): both copies of
// the tile built as the kernel builds them from whole 16-byte loads (NaN where nothing was written), every lane's packed fold
template <int K, int R, int NT, bool FUSED>
static long emu_fir_pairs_t(const float *x, long n_in, const float *taps, float *y)
{
    using G = FirGeomPairs<K, R>;
    constexpr int TILE_OUT = NT * R, TILE_IN = G::tile_in(TILE_OUT), COPY = G::copy_elems(TILE_OUT), NV = (TILE_IN + 3) / 4;
    const long n_out = n_in < K ? 0 : n_in - K + 1;
    for (long tile = 0; tile * TILE_OUT < n_out; ++tile) {
        std::vector<float2> e2((size_t)COPY, make_float2(std::nanf(""), std::nanf(""))), o2 = e2;
        float *ef = reinterpret_cast<float *>(e2.data()), *of = reinterpret_cast<float *>(o2.data());
        const long in0 = tile * (long)TILE_OUT;
        for (int v = 0; v < NV; ++v) {
            float q[4];
            for (int e = 0; e < 4; ++e) q[e] = in0 + 4 * v + e < n_in ? x[in0 + 4 * v + e] : 0.f;
            e2[(size_t)G::lds_index(2 * v)] = make_float2(q[0], q[1]);
            e2[(size_t)G::lds_index(2 * v + 1)] = make_float2(q[2], q[3]);
            o2[(size_t)G::lds_index(2 * v)] = make_float2(q[1], q[2]);
            if (v > 0) of[2 * G::lds_index(2 * v - 1) + 1] = q[0];
            of[2 * G::lds_index(2 * v + 1)] = q[3];
        }
        (void)ef;
        for (int tid = 0; tid < NT; ++tid) {
            float2 acc[R / 2];
            for (int p = 0; p < R / 2; ++p) acc[p] = make_float2(0.f, 0.f);
            fir_lane_pairs<K, R, FUSED>(e2.data(), o2.data(), tid, taps, acc);
            for (int p = 0; p < R / 2; ++p) {
                const long o = in0 + (long)tid * R + 2 * p;
                if (o < n_out) y[o] = acc[p].x;
                if (o + 1 < n_out) y[o + 1] = acc[p].y;
            }
        }
    }
    return n_out;
}
extern "C" long emu_fir_pairs_f32(const float *x, long n_in, const float *taps, int K, int R, int NT, int fused, float *y)
{
    if (K == 63 && R == 16 && NT == 128) return fused ? emu_fir_pairs_t<63, 16, 128, true>(x, n_in, taps, y) : emu_fir_pairs_t<63, 16, 128, false>(x, n_in, taps, y);
    if (K == 63 && R == 8 && NT == 256) return fused ? emu_fir_pairs_t<63, 8, 256, true>(x, n_in, taps, y) : emu_fir_pairs_t<63, 8, 256, false>(x, n_in, taps, y);
    return -2;
}
// ds_read_b64 of the pair images: 32 lanes reading pair slot m of their window must hit 32 different bank pairs
extern "C" int emu_fir_pairs_bank_conflicts(void)
{
    int worst = 1;
    auto chk = [&](auto g) {
        using G = decltype(g);
        for (int m = 0; m < G::NPAIR; ++m) {
            int cnt[32] = {0};
            for (int l = 0; l < 32; ++l) cnt[(l * G::LANE_STRIDE + G::lds_index(m)) % 32]++;
            for (int b = 0; b < 32; ++b) if (cnt[b] > worst) worst = cnt[b];
        }
    };
    chk(FirGeomPairs<63, 16>{});
    chk(FirGeomPairs<63, 8>{});
    return worst;
}

// FIR LDS bank check: 32 lanes reading sample m of their window must hit 32 different banks
// (ds_read_b64 over 64 banks for float2, ds_read_b32 over 32 banks for float)
template <int K, int D, int R>
static int fir_banks(int elem_dwords)
{
    using G = FirGeom<K, D, R>;
    int worst = 1;
    const int nb = 64 / elem_dwords; // distinct element slots per bank row (b64: 32; b32 uses 32 banks)
    for (int m = 0; m < G::SPAN; ++m) {
        int cnt[64] = {0};
        for (int l = 0; l < 32; ++l) {
            int idx = l * (G::LSTR + (G::PAD ? 1 : 0)) + G::lds_index(m);
            cnt[idx % (elem_dwords == 2 ? nb : 32)]++;
        }
        for (int b = 0; b < 64; ++b) if (cnt[b] > worst) worst = cnt[b];
    }
    return worst;
}
extern "C" int emu_fir_bank_conflicts(void)
{
    int w = 1, t;
    if ((t = fir_banks<127, 5, 4>(2)) > w) w = t;
    if ((t = fir_banks<127, 1, 8>(2)) > w) w = t;
    if ((t = fir_banks<63, 1, 8>(2)) > w) w = t;
    if ((t = fir_banks<63, 5, 4>(2)) > w) w = t;
    if ((t = fir_banks<127, 5, 4>(1)) > w) w = t;
    if ((t = fir_banks<63, 1, 8>(1)) > w) w = t;
    return w;
}

// 64-channel polyphase channelizer: one wave tile (16 rows) at a time, lane programs of pfb_core.h
template <int P, bool FUSED>
static long emu_pfb_t(const float2 *x, long n, const float *h, float2 *out, int ngroups)
{
    const long T = n / PFB_M;
    if (T < P) return 0;
    const long rows = T - P + 1;
    std::vector<float2> tw = make_tw(64, 0);
    std::vector<float2> l1(PFB_LDS), l2(PFB_LDS);
    for (long tb = 0; tb < rows; tb += PFB_TILE) {
        for (int lane = 0; lane < 64; ++lane)      // branch FIRs, lane = branch
            for (int ti = 0; ti < PFB_TILE; ++ti) {
                long t = tb + ti;
                if (t >= rows) t = rows - 1;       // clamped (masked at the store)
                float2 acc = make_float2(0.f, 0.f);
                for (int p = 0; p < P; ++p) acc = mac<FUSED>(x[PFB_M * (t + p) + lane], h[PFB_M * p + lane], acc);
                l1[pfb_x1_store(ti, lane)] = acc;
            }
        for (int lane = 0; lane < 64; ++lane) {
            float2 v[16];
            for (int e = 0; e < 16; ++e) v[e] = l1[pfb_x1_load(lane, e)];
            pfb_fft64_passAB<false>(v, tw.data());
            for (int k2 = 0; k2 < 4; ++k2)
                for (int k1 = 0; k1 < 4; ++k1) l2[pfb_x2_store(lane, k1, k2)] = v[k1 + 4 * k2];
        }
        for (int lane = 0; lane < 64; ++lane) {
            float2 w[16];
            for (int f = 0; f < 16; ++f) w[f] = l2[pfb_x2_load(lane, f)];
            pfb_fft64_passC<false>(w, lane, tw.data());
            const long row = tb + (lane >> 2);
            if (row < rows)
                for (int k2 = 0; k2 < 4; ++k2)
                    for (int k0 = 0; k0 < 4; ++k0) out[pfb_out_index(row, pfb_out_channel(lane, k0, k2), rows, ngroups)] = w[k0 + 4 * k2];
        }
    }
    return rows;
}
extern "C" long emu_pfb(const float2 *x, long n, const float *h, int P, int fused, float2 *out, int ngroups)
{
    if (P == 16) return fused ? emu_pfb_t<16, true>(x, n, h, out, ngroups) : emu_pfb_t<16, false>(x, n, h, out, ngroups);
    if (P == 8) return fused ? emu_pfb_t<8, true>(x, n, h, out, ngroups) : emu_pfb_t<8, false>(x, n, h, out, ngroups);
    if (P == 4) return fused ? emu_pfb_t<4, true>(x, n, h, out, ngroups) : emu_pfb_t<4, false>(x, n, h, out, ngroups);
    return -1;
}

// 65536-point two-pass transform: tiles of 256 rows x 16 columns, four in-tile stages per pass
extern "C" void emu_fft64k(const float2 *in, float2 *out, int inverse)
{
    std::vector<float2> tw = make_tw(F64K_N, inverse);
    std::vector<float2> mid(F64K_N), L(256 * F64K_LD);
    for (int pass = 0; pass < 2; ++pass) {
        const float2 *src = pass == 0 ? in : mid.data();
        float2 *dst = pass == 0 ? mid.data() : out;
        for (int c = 0; c < 16; ++c) {
            for (int row = 0; row < 256; ++row)
                for (int col = 0; col < F64K_COLS; ++col) {
                    if (pass == 0) L[rev4_of_8bit(row) * F64K_LD + col] = src[f64k_p0_src(c, row, col)];
                    else L[row * F64K_LD + col] = src[f64k_p1_pos(c, row, col)];
                }
            for (int t = 0; t < 4; ++t)
                for (int col = 0; col < F64K_COLS; ++col)
                    for (int b = 0; b < 64; ++b) {
                        if (inverse) f64k_tile_butterfly<true>(L.data(), tw.data(), pass, t, col, b, F64K_COLS * c + col);
                        else f64k_tile_butterfly<false>(L.data(), tw.data(), pass, t, col, b, F64K_COLS * c + col);
                    }
            for (int row = 0; row < 256; ++row)
                for (int col = 0; col < F64K_COLS; ++col) {
                    if (pass == 0) dst[f64k_p0_dst(c, row, col)] = L[row * F64K_LD + col];
                    else dst[f64k_p1_pos(c, row, col)] = L[row * F64K_LD + col];
                }
        }
    }
}

// the segmentation arithmetic of the carried-history layer (libredio_amd/csrc/stream_split.h)
#include "../../libredio_amd/csrc/stream_split.h"
extern "C" void emu_stream_split(size_t hist, size_t W, size_t H, size_t n, size_t *out5)
{
    const redio::StreamSplit s = redio::stream_split(hist, W, H, n);
    out5[0] = s.nh; out5[1] = s.head_in; out5[2] = s.nb; out5[3] = s.off; out5[4] = s.body_in;
}

// the resampler's position recurrence: short form (libredio_amd/csrc/src_position.h) against the library's literal expression
#include "../../libredio_amd/csrc/src_position.h"
extern "C" long emu_src_advance_mismatches(const double *x, const double *step, long n, long chain)
{
    long bad = 0;
    for (long i = 0; i < n; ++i) {
        double a = x[i], b = x[i];
        for (long k = 0; k < chain; ++k) { // follow both recurrences for `chain` outputs
            const int adv_a = redio::src_advance(a, step[i]);
            b += step[i];
            const double rem = redio::src_fmod_one(b);
            const int adv_b = (int)lrint(b - rem);
            b = rem;
            if (adv_a != adv_b || memcmp(&a, &b, sizeof(double)) != 0) { ++bad; break; }
        }
    }
    return bad;
}

// the register-blocked uniform-phase resampler lane program (libredio_amd/csrc/src_core.h): one tile, every lane of both
// wings, on an LDS image built as the kernel builds it -- slack floats and table guard zones filled with NaN, so a
// sample or coefficient that belongs to no tap of an output cannot reach it unnoticed.  out_ref: the plain per-output
// sums in the order calc_output_single (libsamplerate 0.1.8) runs them.
#include "../../libredio_amd/csrc/src_core.h"
template <int R, int U>
static void emu_src_rb_t(const float *xt, const double *L, int ncl, const double *Rt, int ncr, int S, int NO, double scale, float *out)
{
    const int cl = ncl - 1, cr = ncr - 1, c = cl + 1 + cr, B = R * S, P = redio::src_rb_pad(B), LW = NO / R;
    const long span = (long)(NO - 1) * S + cl + cr + 2;
    const double qnan = nan("");
    std::vector<float> xs((size_t)redio::src_rb_tile_floats(NO, R, S, cl, cr), nanf(""));
    for (long n = 0; n < span; ++n) xs[(size_t)(n + P * (n / B))] = xt[n];
    const int guard = (R - 1) * S + 2 * U;
    std::vector<double> Lg((size_t)(ncl + 2 * guard), qnan), Rg((size_t)(ncr + 2 * guard), qnan);
    memcpy(Lg.data() + guard, L, sizeof(double) * (size_t)ncl);
    memcpy(Rg.data() + guard, Rt, sizeof(double) * (size_t)ncr);
    std::vector<double> rsum((size_t)NO);
    for (int q = 0; q < LW; ++q) {
        double acc[R];
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
        redio::src_rb_wing<R, U, -1>(xs.data() + (size_t)q * (B + P), B, P, (R - 1) * S + c, Rg.data() + guard, ncr, S, acc);
        for (int r = 0; r < R; ++r) rsum[(size_t)(q * R + (R - 1 - r))] = acc[r];
    }
    for (int q = 0; q < LW; ++q) {
        double acc[R];
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
        redio::src_rb_wing<R, U, +1>(xs.data() + (size_t)q * (B + P), B, P, 0, Lg.data() + guard, ncl, S, acc);
        for (int r = 0; r < R; ++r) out[q * R + r] = (float)(scale * (acc[r] + rsum[(size_t)(q * R + r)]));
    }
}

extern "C" int emu_src_rb(const float *xt, const double *L, int ncl, const double *Rt, int ncr, int S, int R, int U, int NO, double scale,
                          float *out, float *out_ref)
{
    const int cl = ncl - 1, cr = ncr - 1, c = cl + 1 + cr;
    for (int o = 0; o < NO; ++o) {
        double left = 0.0, right = 0.0;
        for (int t = 0; t <= cl; ++t) left += L[t] * (double)xt[(long)S * o + t];
        for (int t = 0; t <= cr; ++t) right += Rt[t] * (double)xt[(long)S * o + c - t];
        out_ref[o] = (float)(scale * (left + right));
    }
    if ((R * S) % 4 != 0 || NO % R != 0) return -1;
    if (R == 2 && U == 8) emu_src_rb_t<2, 8>(xt, L, ncl, Rt, ncr, S, NO, scale, out);
    else if (R == 2 && U == 4) emu_src_rb_t<2, 4>(xt, L, ncl, Rt, ncr, S, NO, scale, out);
    else if (R == 4 && U == 4) emu_src_rb_t<4, 4>(xt, L, ncl, Rt, ncr, S, NO, scale, out);
    else if (R == 4 && U == 8) emu_src_rb_t<4, 8>(xt, L, ncl, Rt, ncr, S, NO, scale, out);
    else if (R == 8 && U == 4) emu_src_rb_t<8, 4>(xt, L, ncl, Rt, ncr, S, NO, scale, out);
    else return -1;
    return 0;
}

// ================================================================================================================================
// The two-column ("pair") tile program of the four-stage passes (libredio_amd/csrc/fft_big_core.h; device side: fft_pair.h), one
// lane at a time: the same lane <-> row / column maps, LDS images, group maps and twiddle batches as the kernels, so that the
// 65536-point transform and the three-pass overlap-save block can be checked against the oracle without a GPU.
// ================================================================================================================================
#include "../../libredio_amd/csrc/fft_big_core.h"

namespace {
struct PairLane { float2 a[2][2][16], b[2][2][16]; };

// the tables the plans build on the device (fftbig_tables_kernel / fftbig_tables_inter_kernel): same formulas
std::vector<float2> pair_ordered_table(const std::vector<float2> &tw, unsigned m_lo, int nstages, unsigned N)
{
    const unsigned total = m_lo * ((1u << (2 * nstages)) - 1);
    std::vector<float2> T(total);
    for (unsigned i = 0; i < total; ++i) {
        int t = 0;
        while (i >= m_lo * ((1u << (2 * (t + 1))) - 1)) ++t;
        const unsigned m = m_lo << (2 * t), r = i - m_lo * ((1u << (2 * t)) - 1), n = r / m + 1, k = r - (n - 1) * m;
        T[i] = tw[(size_t)n * k * (N / (4 * m))];
    }
    return T;
}
std::vector<float2> pair_inter_table(const std::vector<float2> &tw, unsigned m_lo, int nstages, unsigned N)
{
    const unsigned total = m_lo * (((1u << (2 * nstages)) - 1) / 3);
    std::vector<float2> T((size_t)4 * total + 8);
    for (unsigned i = 0; i < total; ++i) {
        int t = 0;
        while (i >= m_lo * (((1u << (2 * (t + 1))) - 1) / 3)) ++t;
        const unsigned m = m_lo << (2 * t), k = i - m_lo * (((1u << (2 * t)) - 1) / 3), fs = N / (4 * m);
        T[4 * (size_t)i] = tw[(size_t)k * fs]; T[4 * (size_t)i + 1] = tw[(size_t)2 * k * fs]; T[4 * (size_t)i + 2] = tw[(size_t)3 * k * fs];
        T[4 * (size_t)i + 3] = make_float2(0.f, 0.f);
    }
    return T;
}

template <typename G> void pair_exchange_plain(std::vector<PairLane> &L, std::vector<float4> &img)
{
    auto round = [&](auto wr, auto rd) {
        for (int lane = 0; lane < 64; ++lane) wr(lane);
        for (int lane = 0; lane < 64; ++lane) rd(lane);
    };
    round([&](int l) { pw_plain_write<0, 0>(L[l].a, img.data(), l); }, [&](int l) { pw_plain_read<G, 0, 0>(L[l].b, img.data(), l); });
    round([&](int l) { pw_plain_write<0, 1>(L[l].a, img.data(), l); }, [&](int l) { pw_plain_read<G, 0, 1>(L[l].b, img.data(), l); });
    round([&](int l) { pw_plain_write<1, 0>(L[l].a, img.data(), l); }, [&](int l) { pw_plain_read<G, 1, 0>(L[l].b, img.data(), l); });
    round([&](int l) { pw_plain_write<1, 1>(L[l].a, img.data(), l); }, [&](int l) { pw_plain_read<G, 1, 1>(L[l].b, img.data(), l); });
}
template <typename G> void pair_exchange_tr(std::vector<PairLane> &L, std::vector<float4> &img)
{
    auto round = [&](auto wr, auto rd) {
        for (int lane = 0; lane < 64; ++lane) wr(lane);
        for (int lane = 0; lane < 64; ++lane) rd(lane);
    };
    round([&](int l) { pw_tr_write<0, 0>(L[l].a, img.data(), l); }, [&](int l) { pw_tr_read<G, 0, 0>(L[l].b, img.data(), l); });
    round([&](int l) { pw_tr_write<0, 1>(L[l].a, img.data(), l); }, [&](int l) { pw_tr_read<G, 0, 1>(L[l].b, img.data(), l); });
    round([&](int l) { pw_tr_write<1, 0>(L[l].a, img.data(), l); }, [&](int l) { pw_tr_read<G, 1, 0>(L[l].b, img.data(), l); });
    round([&](int l) { pw_tr_write<1, 1>(L[l].a, img.data(), l); }, [&](int l) { pw_tr_read<G, 1, 1>(L[l].b, img.data(), l); });
}

// pw_mid_stages of fft_pair.h: load, stages 0-1, plain regrouping, stages 2-3
template <bool INV> void pair_mid_stages(std::vector<PairLane> &L, const float2 *base, long m_lo, unsigned l0, const float2 *T, std::vector<float4> &img)
{
    const unsigned ml = (unsigned)m_lo;
    for (int lane = 0; lane < 64; ++lane) {
        const int cp = lane & 7, q = lane >> 3;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 16; ++j) {
                const float2 *p = base + pw_mid_ld(m_lo, 0, 0, i, j) + pw_mid_ld(m_lo, q, cp, 0, 0);
                L[lane].a[i][0][j] = p[0]; L[lane].a[i][1][j] = p[1];
            }
        {
            FftTw15 T0, T1;
            big_tw15x2(T0, T1, tw_pair_stage(T, ml, 0), tw_pair_stage(T, ml, 1), l0 + 2u * cp, ml, 0u, 1u);
            macro16_apply<INV>(L[lane].a[0][0], T0); macro16_apply<INV>(L[lane].a[1][0], T0);
            macro16_apply<INV>(L[lane].a[0][1], T1); macro16_apply<INV>(L[lane].a[1][1], T1);
        }
    }
    pair_exchange_plain<PwGroupsLinear>(L, img);
    for (int lane = 0; lane < 64; ++lane) {
        const int cp = lane & 7, q = lane >> 3;
        for (int x = 0; x < 2; ++x) {
            FftTw15 T0, T1;
            big_tw15x2(T0, T1, tw_pair_stage(T, ml, 2), tw_pair_stage(T, ml, 3), l0 + 2u * cp, ml, (unsigned)(q + 8 * x), 16u);
            macro16_apply<INV>(L[lane].b[x][0], T0); macro16_apply<INV>(L[lane].b[x][1], T1);
        }
    }
}
template <bool INV> void pair_mid_tile(float2 *base, long m_lo, unsigned l0, const float2 *T, std::vector<float4> &img)
{
    std::vector<PairLane> L(64);
    pair_mid_stages<INV>(L, base, m_lo, l0, T, img);
    for (int lane = 0; lane < 64; ++lane) {
        const int cp = lane & 7, q = lane >> 3;
        for (int x = 0; x < 2; ++x)
            for (int j = 0; j < 16; ++j) {
                float2 *p = base + pw_mid_st(m_lo, 0, 0, x, j) + pw_mid_st(m_lo, q, cp, 0, 0);
                p[0] = L[lane].b[x][0][j]; p[1] = L[lane].b[x][1][j];
            }
    }
}
// pw_first_tile
template <bool INV> void pair_first_tile(const float2 *in_blk, float2 *out_blk, int Lg, unsigned c, const float2 *T1, std::vector<float4> &img)
{
    const long S = 1l << (2 * Lg - 8);
    std::vector<PairLane> L(64);
    for (int lane = 0; lane < 64; ++lane) {
        const int cp = lane & 7, q = lane >> 3;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 16; ++j) {
                const float2 *p = in_blk + 16 * c + pw_first_ld(S, 0, 0, i, j) + pw_first_ld(S, q, cp, 0, 0);
                L[lane].a[i][0][j] = p[0]; L[lane].a[i][1][j] = p[1];
            }
        FftTw15 T0;
        big_tw15(T0, tw_ordered_stage(T1, 1u, 0), tw_ordered_stage(T1, 1u, 1), 0u, 1u, 0u, 1u);
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 2; ++e) macro16_apply<INV>(L[lane].a[i][e], T0);
    }
    pair_exchange_tr<PwGroupsLinear>(L, img);
    unsigned rc = 0;
    for (int d = 0, cc = (int)c; d < Lg - 6; ++d, cc >>= 2) rc = (rc << 2) | (cc & 3);
    for (int lane = 0; lane < 64; ++lane) {
        const int sp = lane & 7, qq = lane >> 3;
        {
            FftTw15 T0, Tb;
            big_tw15x2(T0, Tb, tw_pair_stage_u(T1, 1u, 2), tw_pair_stage_u(T1, 1u, 3), 0u, 1u, (unsigned)(2 * sp), 16u);
            for (int x = 0; x < 2; ++x) { macro16_apply<INV>(L[lane].b[x][0], T0); macro16_apply<INV>(L[lane].b[x][1], Tb); }
        }
        for (int x = 0; x < 2; ++x)
            for (int j = 0; j < 16; ++j) {
                float2 *p = out_blk + 256l * rc + pw_first_st(Lg, 0, 0, x, j) + pw_first_st(Lg, qq, sp, 0, 0);
                p[0] = L[lane].b[x][0][j]; p[1] = L[lane].b[x][1][j];
            }
    }
}
} // namespace

// 65536-point transform = gather pass + one in-place pass, every tile through the pair program
extern "C" void emu_pair_fft64k(const float2 *in, float2 *out, int inverse)
{
    const unsigned N = 65536;
    std::vector<float2> tw = make_tw((int)N, inverse);
    std::vector<float2> T1 = pair_ordered_table(tw, 1u, 5, N), T = pair_ordered_table(tw, 256u, 4, N);
    std::vector<float4> img(PW_UNITS);
    for (unsigned c = 0; c < 16; ++c) {
        if (inverse) pair_first_tile<true>(in, out, 8, c, T1.data(), img);
        else pair_first_tile<false>(in, out, 8, c, T1.data(), img);
    }
    for (unsigned c = 0; c < 16; ++c) {
        if (inverse) pair_mid_tile<true>(out + 16 * c, 256l, 16 * c, T.data(), img);
        else pair_mid_tile<false>(out + 16 * c, 256l, 16 * c, T.data(), img);
    }
}

// one 65536-point overlap-save block through the three passes: x (65536 samples), Hc = conj(kiss_fft(h padded)), out (hop samples)
extern "C" void emu_pair_ovsave64k(const float2 *x, const float2 *Hc, float2 *out, long hop)
{
    const unsigned N = 65536;
    std::vector<float2> twf = make_tw((int)N, 0), twi = make_tw((int)N, 1);
    std::vector<float2> T1 = pair_ordered_table(twf, 1u, 5, N), T1i = pair_ordered_table(twi, 1u, 5, N), Tf = pair_ordered_table(twf, 256u, 4, N), Ti = pair_ordered_table(twi, 256u, 4, N);
    std::vector<float4> img(PW_UNITS);
    std::vector<float2> A(N), B(N);
    for (unsigned c = 0; c < 16; ++c) pair_first_tile<false>(x, A.data(), 8, c, T1.data(), img);
    for (int c = 0; c < 16; ++c) { // pw_ovsave64k_mid_tile
        std::vector<PairLane> L(64);
        pair_mid_stages<false>(L, A.data() + 16 * c, 256l, (unsigned)(16 * c), Tf.data(), img);
        for (int lane = 0; lane < 64; ++lane) {
            const int cp = lane & 7, q = lane >> 3;
            for (int xx = 0; xx < 2; ++xx)
                for (int j = 0; j < 16; ++j) {
                    const float2 *h = Hc + 16 * c + pw_mid_st(256l, 0, 0, xx, j) + pw_mid_st(256l, q, cp, 0, 0);
                    L[lane].a[xx][0][pw_rev2(j)] = cmul_rn(L[lane].b[xx][0][j], h[0]);
                    L[lane].a[xx][1][pw_rev2(j)] = cmul_rn(L[lane].b[xx][1][j], h[1]);
                }
            FftTw15 T0;
            big_tw15(T0, tw_ordered_stage(T1i.data(), 1u, 0), tw_ordered_stage(T1i.data(), 1u, 1), 0u, 1u, 0u, 1u);
            for (int xx = 0; xx < 2; ++xx)
                for (int e = 0; e < 2; ++e) macro16_apply<true>(L[lane].a[xx][e], T0);
        }
        pair_exchange_tr<PwGroupsRev>(L, img);
        for (int lane = 0; lane < 64; ++lane) {
            const int sp = lane & 7, qq = lane >> 3;
            {
                FftTw15 T0, Tb;
                big_tw15x2(T0, Tb, tw_pair_stage_u(T1i.data(), 1u, 2), tw_pair_stage_u(T1i.data(), 1u, 3), 0u, 1u, (unsigned)(2 * sp), 16u);
                for (int xx = 0; xx < 2; ++xx) { macro16_apply<true>(L[lane].b[xx][0], T0); macro16_apply<true>(L[lane].b[xx][1], Tb); }
            }
            for (int xx = 0; xx < 2; ++xx)
                for (int j = 0; j < 16; ++j) {
                    float2 *p = B.data() + 256 * pw_rev2(c) + pw_first_st(8, 0, 0, xx, j) + pw_first_st(8, qq, sp, 0, 0);
                    p[0] = L[lane].b[xx][0][j]; p[1] = L[lane].b[xx][1][j];
                }
        }
    }
    const float scale = 1.0f / 65536.0f;
    for (int c = 0; c < 16; ++c) { // pw_ovsave64k_last_tile
        std::vector<PairLane> L(64);
        pair_mid_stages<true>(L, B.data() + 16 * c, 256l, (unsigned)(16 * c), Ti.data(), img);
        for (int lane = 0; lane < 64; ++lane) {
            const int cp = lane & 7, q = lane >> 3;
            const long lo = pw_mid_st(256l, q, cp, 0, 0), lim = hop - 16 * c - lo;
            for (int xx = 0; xx < 2; ++xx)
                for (int j = 0; j < 16; ++j) {
                    const long r = pw_mid_st(256l, 0, 0, xx, j);
                    for (int e = 0; e < 2; ++e)
                        if (r + e < lim) out[16 * c + r + lo + e] = make_float2(mul_rn(L[lane].b[xx][e][j].x, scale), mul_rn(L[lane].b[xx][e][j].y, scale));
                }
        }
    }
}

// ds_write_b128 / ds_read_b128 of the pair images: a b128 access is served in four groups of 16 lanes ({0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}, the same + 32); returns the worst number of lanes of one group that share a 16-byte bank position (1 = conflict free)
extern "C" int emu_pair_bank_conflicts(void)
{
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int worst = 1;
    auto check = [&](auto unit_of /* (lane, access) -> unit */) {
        for (int acc = 0; acc < 8; ++acc)
            for (int half = 0; half < 2; ++half)
                for (int g = 0; g < 2; ++g) {
                    int cnt[16] = {0};
                    for (int t = 0; t < 16; ++t) cnt[unit_of(grp[g][t] + 32 * half, acc) % 16]++;
                    for (int b = 0; b < 16; ++b) if (cnt[b] > worst) worst = cnt[b];
                }
    };
    check([](int l, int jj) { return pw_unit_plain(l >> 3, jj, l & 7); });  // pw_plain_write
    check([](int l, int qw) { return pw_unit_plain(qw, l >> 3, l & 7); });  // pw_plain_read
    check([](int l, int jp) { return pw_unit_tr(l & 7, l >> 3, jp); });     // pw_tr_write
    check([](int l, int qw) { return pw_unit_tr(l >> 3, qw, l & 7); });     // pw_tr_read
    // the images are permutations of the 512 units
    bool seen[2][PW_UNITS] = {{false}};
    for (int a = 0; a < 8; ++a)
        for (int b = 0; b < 8; ++b)
            for (int c = 0; c < 8; ++c) {
                const int u = pw_unit_plain(a, b, c), v = pw_unit_tr(a, b, c);
                if (u < 0 || u >= PW_UNITS || v < 0 || v >= PW_UNITS || seen[0][u] || seen[1][v]) return 99;
                seen[0][u] = seen[1][v] = true;
            }
    return worst;
}

// ---- G128: the four-stage gather pass of N = 2 * 4^L' points (fft_big_core.h; device side pw_g128_tile / pw_ovsave32k_mid_tile) ----------
namespace {
struct GLane { float2 a[2][4][8], b[2][4][2][4]; };
std::vector<float2> g_table(const std::vector<float2> &tw, unsigned N)
{
    std::vector<float2> T(PW_G_TABLE);
    for (int i = 0; i < PW_G_TABLE; ++i) pw_g_table_entry(tw.data(), N, i, T[i]);
    return T;
}
template <bool INV> void g128_stages(std::vector<GLane> &L, const float2 *Tg, std::vector<float4> &img)
{
    for (int lane = 0; lane < 64; ++lane) { pw_g_inlane<INV>(L[lane].a[0], Tg); pw_g_inlane<INV>(L[lane].a[1], Tg); }
    for (int lane = 0; lane < 64; ++lane) pw_g_write<0>(L[lane].a, img.data(), lane);
    for (int lane = 0; lane < 64; ++lane) pw_g_read<0>(L[lane].b, img.data(), lane);
    for (int lane = 0; lane < 64; ++lane) pw_g_write<1>(L[lane].a, img.data(), lane);
    for (int lane = 0; lane < 64; ++lane) pw_g_read<1>(L[lane].b, img.data(), lane);
    for (int lane = 0; lane < 64; ++lane) pw_g_last<INV>(L[lane].b, Tg, lane & 7);
}
template <bool INV> void g128_tile(const float2 *in_blk, float2 *out_blk, int lgN, unsigned ctile, const float2 *Tg, std::vector<float4> &img)
{
    const long S = 1l << (lgN - 7);
    const int nd = (lgN - 7) / 2;
    std::vector<GLane> L(64);
    for (int lane = 0; lane < 64; ++lane) {
        const int cp = lane & 15, q = lane >> 4;
        for (int d2 = 0; d2 < 4; ++d2)
            for (int jb = 0; jb < 8; ++jb) {
                const float2 *p = in_blk + 32 * ctile + pw_g_ld(S, 0, 0, d2, jb) + pw_g_ld(S, q, cp, 0, 0);
                L[lane].a[0][d2][jb] = p[0]; L[lane].a[1][d2][jb] = p[1];
            }
    }
    g128_stages<INV>(L, Tg, img);
    unsigned hc = 0;
    for (int d = 0, cc = (int)(ctile >> 1); d < nd - 3; ++d, cc >>= 2) hc = (hc << 2) | (cc & 3);
    float2 *dst = out_blk + 128l * (((long)(2 * (ctile & 1))) * (1l << (2 * (nd - 3))) + hc);
    for (int lane = 0; lane < 64; ++lane) {
        const int kp = lane & 7, cg = lane >> 3;
        for (int r = 0; r < 2; ++r)
            for (int x = 0; x < 4; ++x)
                for (int d3 = 0; d3 < 4; ++d3) {
                    float2 *p = dst + pw_g_st(nd, 0, 0, x, r, d3) + pw_g_st(nd, cg, kp, 0, 0, 0);
                    p[0] = L[lane].b[r][x][0][d3]; p[1] = L[lane].b[r][x][1][d3];
                }
    }
}
} // namespace

// N = 2^lgN (odd lgN >= 15): G128 through the pair program's maps, then the remaining radix-4 stages with the generic in-place stage
// (lgN = 15: through the pair program's four-stage in-place pass on rows 128 apart instead, as the plan runs it)
extern "C" int emu_pair_g128_fft(int lgN, const float2 *in, float2 *out, int inverse)
{
    const unsigned N = 1u << lgN;
    FftStage st[32];
    const int ns = fft_plan_stages((int)N, st, 32);
    if (ns < 5 || st[ns - 1].p != 2) return -1;
    std::vector<float2> tw = make_tw((int)N, inverse), Tg = g_table(tw, N);
    std::vector<float4> img(PW_G_UNITS);
    for (unsigned ct = 0; ct < (N >> 12); ++ct) {
        if (inverse) g128_tile<true>(in, out, lgN, ct, Tg.data(), img);
        else g128_tile<false>(in, out, lgN, ct, Tg.data(), img);
    }
    if (lgN == 15) {
        std::vector<float2> Tm = pair_ordered_table(tw, 128u, 4, N);
        std::vector<float4> img2(PW_UNITS);
        for (unsigned c = 0; c < 8; ++c) {
            if (inverse) pair_mid_tile<true>(out + 16 * c, 128l, 16 * c, Tm.data(), img2);
            else pair_mid_tile<false>(out + 16 * c, 128l, 16 * c, Tm.data(), img2);
        }
        return ns;
    }
    for (int s = ns - 5; s >= 0; --s)
        for (unsigned b = 0; b < N / 4; ++b) {
            if (inverse) fft_stage_butterfly<true>(out, tw.data(), st[s], (int)b);
            else fft_stage_butterfly<false>(out, tw.data(), st[s], (int)b);
        }
    return ns;
}

// one 32768-point overlap-save block through the three passes of the plan: G128 forward; [in-place forward pass x conj H x inverse G128] per tile;
// inverse in-place pass with the masked, scaled store
extern "C" void emu_pair_ovsave32k(const float2 *x, const float2 *Hc, float2 *out, long hop)
{
    const unsigned N = 32768;
    std::vector<float2> twf = make_tw((int)N, 0), twi = make_tw((int)N, 1);
    std::vector<float2> Tgf = g_table(twf, N), Tgi = g_table(twi, N), Tf = pair_ordered_table(twf, 128u, 4, N), Ti = pair_ordered_table(twi, 128u, 4, N);
    std::vector<float4> img(PW_G_UNITS);
    std::vector<float2> A(N), B(N);
    for (unsigned ct = 0; ct < 8; ++ct) g128_tile<false>(x, A.data(), 15, ct, Tgf.data(), img);
    for (int c = 0; c < 8; ++c) { // pw_ovsave32k_mid_tile
        std::vector<PairLane> F(64);
        std::vector<GLane> L(64);
        pair_mid_stages<false>(F, A.data() + 16 * c, 128l, (unsigned)(16 * c), Tf.data(), img);
        for (int lane = 0; lane < 64; ++lane) {
            const int cp = lane & 7, q = lane >> 3;
            for (int xx = 0; xx < 2; ++xx)
                for (int j = 0; j < 16; ++j) {
                    const float2 *h = Hc + 16 * c + pw_mid_st(128l, 0, 0, xx, j) + pw_mid_st(128l, q, cp, 0, 0);
                    const int d2 = xx + 2 * (j & 1), jb = (j >> 3) + 2 * ((j >> 1) & 3);
                    L[lane].a[0][d2][jb] = cmul_rn(F[lane].b[xx][0][j], h[0]);
                    L[lane].a[1][d2][jb] = cmul_rn(F[lane].b[xx][1][j], h[1]);
                }
        }
        g128_stages<true>(L, Tgi.data(), img);
        for (int lane = 0; lane < 64; ++lane) {
            const int kp = lane & 7, cg = lane >> 3;
            float2 *dst = B.data() + 128l * (4 * (c & 3) + (c >> 2)) + (128 * (64 * (cg & 3) + 16 * (cg >> 2)) + 2 * kp);
            for (int r = 0; r < 2; ++r)
                for (int xx = 0; xx < 4; ++xx)
                    for (int d3 = 0; d3 < 4; ++d3) {
                        float2 *p = dst + (128 * (32 * (xx & 1) + 2 * (xx >> 1)) + 32 * d3 + 16 * r);
                        p[0] = L[lane].b[r][xx][0][d3]; p[1] = L[lane].b[r][xx][1][d3];
                    }
        }
    }
    const float scale = 1.0f / 32768.0f;
    std::vector<float4> img2(PW_UNITS);
    for (int c = 0; c < 8; ++c) { // the inverse in-place pass (pw_mid_tile with vout)
        std::vector<PairLane> L(64);
        pair_mid_stages<true>(L, B.data() + 16 * c, 128l, (unsigned)(16 * c), Ti.data(), img2);
        for (int lane = 0; lane < 64; ++lane) {
            const int cp = lane & 7, q = lane >> 3;
            for (int xx = 0; xx < 2; ++xx)
                for (int j = 0; j < 16; ++j)
                    for (int e = 0; e < 2; ++e) {
                        const long pos = 16 * c + pw_mid_st(128l, 0, 0, xx, j) + pw_mid_st(128l, q, cp, 0, 0) + e;
                        if (pos < hop) out[pos] = make_float2(mul_rn(L[lane].b[xx][e][j].x, scale), mul_rn(L[lane].b[xx][e][j].y, scale));
                    }
        }
    }
}

extern "C" int emu_pair_g128_bank_conflicts(void)
{
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int worst = 1;
    auto tally = [&](const int *lanes, auto unit_of, int acc) {
        int cnt[16] = {0};
        for (int t = 0; t < 16; ++t) cnt[unit_of(lanes[t], acc) % 16]++;
        for (int b = 0; b < 16; ++b) if (cnt[b] > worst) worst = cnt[b];
    };
    auto wr = [](int l, int acc) { return pw_unit_g(2 * (l & 15) + (acc >> 3), l >> 4, acc & 7); };           // acc = 8 e + kp
    auto rd = [](int l, int acc) { return pw_unit_g((l >> 3) + 8 * (acc >> 2), acc & 3, l & 7); };            // acc = 4 x + d3
    for (int acc = 0; acc < 16; ++acc)
        for (int half = 0; half < 2; ++half)
            for (int g = 0; g < 2; ++g) {
                int lanes[16], seq[16];
                for (int t = 0; t < 16; ++t) { lanes[t] = grp[g][t] + 32 * half; seq[t] = 16 * (2 * half + g) + t; }
                tally(lanes, rd, acc);  // ds_read_b128: the documented lane groups
                tally(lanes, wr, acc);  // ds_write_b128: the same groups ...
                tally(seq, wr, acc);    // ... and sixteen consecutive lanes
            }
    bool seen[PW_G_UNITS] = {false};
    for (int g = 0; g < 32; ++g)
        for (int d = 0; d < 4; ++d)
            for (int k = 0; k < 8; ++k) {
                const int u = pw_unit_g(g, d, k);
                if (u < 0 || u >= PW_G_UNITS || seen[u]) return 99;
                seen[u] = true;
            }
    return worst;
}

// ---- the five-stage passes in the pair layout (pw_mid5_tile / pw_first5_tile): four "wavefronts" per tile, the fifth stage across them ----
namespace {
template <bool INV, typename TP, typename KFn, typename StFn>
void pair_x5_rounds(std::vector<PairLane> (&L)[4], std::vector<float4> &X, TP t4, unsigned dk, KFn k0_of, StFn store)
{
    auto round = [&](auto wr, int x, int jh) {
        float4 *Xi = X.data() + ((((2 * x + jh) & 1) != 0) ? PW_X5_UNITS : 0);
        for (int w = 0; w < 4; ++w)
            for (int lane = 0; lane < 64; ++lane) wr(L[w][lane].b, Xi, lane, w);
        for (int w = 0; w < 4; ++w)
            for (int lane = 0; lane < 64; ++lane) {
                float2 v[2][4][2];
                pw_x5_read(v, Xi, lane, w);
                pw_x5_stage<INV>(v, t4, k0_of(x, jh, lane, w), dk);
                store(x, jh, lane, w, v);
            }
    };
    round([](const float2 (&b)[2][2][16], float4 *Xi, int lane, int w) { pw_x5_write<0, 0>(b, Xi, lane, w); }, 0, 0);
    round([](const float2 (&b)[2][2][16], float4 *Xi, int lane, int w) { pw_x5_write<0, 1>(b, Xi, lane, w); }, 0, 1);
    round([](const float2 (&b)[2][2][16], float4 *Xi, int lane, int w) { pw_x5_write<1, 0>(b, Xi, lane, w); }, 1, 0);
    round([](const float2 (&b)[2][2][16], float4 *Xi, int lane, int w) { pw_x5_write<1, 1>(b, Xi, lane, w); }, 1, 1);
}
template <bool INV> void pair_mid5_tile(float2 *tile, long m_lo, unsigned l0, const float2 *T, std::vector<float4> &X)
{
    std::vector<PairLane> L[4];
    std::vector<float4> img(PW_UNITS);
    const unsigned ml = (unsigned)m_lo;
    for (int w = 0; w < 4; ++w) { L[w].resize(64); pair_mid_stages<INV>(L[w], tile + 256 * m_lo * w, m_lo, l0, T, img); }
    pair_x5_rounds<INV>(L, X, tw_pair_stage(T, ml, 4), 16u * ml,
        [&](int x, int jh, int lane, int w) { const int cp = lane & 7, q = lane >> 3; return l0 + 2u * cp + ml * (unsigned)(q + 8 * x + 16 * (8 * jh + 2 * w)); },
        [&](int x, int jh, int lane, int w, float2 (&v)[2][4][2]) {
            const int cp = lane & 7, q = lane >> 3;
            for (int jp = 0; jp < 2; ++jp)
                for (int n = 0; n < 4; ++n) {
                    float2 *p = tile + m_lo * (256 * n + 8 * x + 16 * (8 * jh + 2 * w + jp)) + (m_lo * q + 2 * cp);
                    p[0] = v[jp][n][0]; p[1] = v[jp][n][1];
                }
        });
}
// pw_first_stages for one "wavefront"
template <bool INV> void pair_first_stages(std::vector<PairLane> &L, const float2 *src, long S, const float2 *T1, std::vector<float4> &img)
{
    for (int lane = 0; lane < 64; ++lane) {
        const int cp = lane & 7, q = lane >> 3;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 16; ++j) {
                const float2 *p = src + pw_first_ld(S, 0, 0, i, j) + pw_first_ld(S, q, cp, 0, 0);
                L[lane].a[i][0][j] = p[0]; L[lane].a[i][1][j] = p[1];
            }
        FftTw15 T0;
        big_tw15(T0, tw_ordered_stage(T1, 1u, 0), tw_ordered_stage(T1, 1u, 1), 0u, 1u, 0u, 1u);
        for (int i = 0; i < 2; ++i)
            for (int e = 0; e < 2; ++e) macro16_apply<INV>(L[lane].a[i][e], T0);
    }
    pair_exchange_tr<PwGroupsLinear>(L, img);
    for (int lane = 0; lane < 64; ++lane) {
        const int sp = lane & 7;
        FftTw15 T0, Tb;
        big_tw15x2(T0, Tb, tw_pair_stage_u(T1, 1u, 2), tw_pair_stage_u(T1, 1u, 3), 0u, 1u, (unsigned)(2 * sp), 16u);
        for (int x = 0; x < 2; ++x) { macro16_apply<INV>(L[lane].b[x][0], T0); macro16_apply<INV>(L[lane].b[x][1], Tb); }
    }
}
template <bool INV> void pair_first5_tile(const float2 *in_blk, float2 *out_blk, int Lg, unsigned c, const float2 *T1, std::vector<float4> &X)
{
    const long S = 1l << (2 * Lg - 8), S5 = 1l << (2 * Lg - 10);
    std::vector<PairLane> L[4];
    std::vector<float4> img(PW_UNITS);
    for (int w = 0; w < 4; ++w) { L[w].resize(64); pair_first_stages<INV>(L[w], in_blk + 16 * c + S5 * w, S, T1, img); }
    unsigned rc = 0;
    for (int d = 0, cc = (int)c; d < Lg - 7; ++d, cc >>= 2) rc = (rc << 2) | (cc & 3);
    float2 *dst = out_blk + 1024l * rc;
    pair_x5_rounds<INV>(L, X, tw_pair_stage_u(T1, 1u, 4), 16u,
        [&](int x, int jh, int lane, int w) { (void)x; return (unsigned)(2 * (lane & 7) + 16 * (8 * jh + 2 * w)); },
        [&](int x, int jh, int lane, int w, float2 (&v)[2][4][2]) {
            const int sp = lane & 7, qq = lane >> 3;
            for (int jp = 0; jp < 2; ++jp)
                for (int n = 0; n < 4; ++n) {
                    float2 *p = dst + pw_first5_st(Lg, 0, 0, x, n, 0) + 16 * (8 * jh + 2 * w + jp) + pw_first5_st(Lg, qq, sp, 0, 0, 0);
                    p[0] = v[jp][n][0]; p[1] = v[jp][n][1];
                }
        });
}
} // namespace

// 2^18 = four-stage gather pass + five-stage in-place pass; 2^20 = five-stage gather pass + five-stage in-place pass (plan B), all in the pair layout
extern "C" int emu_pair_fft_five(int lgN, const float2 *in, float2 *out, int inverse)
{
    if (lgN != 18 && lgN != 20) return -1;
    const unsigned N = 1u << lgN;
    const int Lg = lgN / 2;
    std::vector<float2> tw = make_tw((int)N, inverse);
    std::vector<float2> T1 = pair_ordered_table(tw, 1u, 5, N);
    std::vector<float4> img(PW_UNITS), X(2 * PW_X5_UNITS);
    int lm;
    if (lgN == 18) {
        for (unsigned c = 0; c < (N >> 12); ++c) { if (inverse) pair_first_tile<true>(in, out, Lg, c, T1.data(), img); else pair_first_tile<false>(in, out, Lg, c, T1.data(), img); }
        lm = 8;
    } else {
        for (unsigned c = 0; c < (N >> 14); ++c) { if (inverse) pair_first5_tile<true>(in, out, Lg, c, T1.data(), X); else pair_first5_tile<false>(in, out, Lg, c, T1.data(), X); }
        lm = 10;
    }
    const unsigned m_lo = 1u << lm;
    std::vector<float2> T = pair_ordered_table(tw, m_lo, 5, N);
    for (unsigned g = 0; g < (N >> 14); ++g) { // tiles of 1024 rows x 16 columns
        const unsigned c = g & ((m_lo >> 4) - 1), H = g >> (lm - 4);
        float2 *tile = out + (long)H * 1024 * m_lo + 16 * c;
        if (inverse) pair_mid5_tile<true>(tile, (long)m_lo, 16 * c, T.data(), X); else pair_mid5_tile<false>(tile, (long)m_lo, 16 * c, T.data(), X);
    }
    return 0;
}

extern "C" int emu_pair_x5_bank_conflicts(void)
{
    static const int grp[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int worst = 1;
    for (int n = 0; n < 4; ++n)
        for (int jj = 0; jj < 8; ++jj)
            for (int half = 0; half < 2; ++half)
                for (int g = 0; g < 2; ++g) {
                    int cnt[16] = {0}, cnt2[16] = {0};
                    for (int t = 0; t < 16; ++t) {
                        const int l = grp[g][t] + 32 * half, l2 = 16 * (2 * half + g) + t;
                        cnt[pw_unit_x5(n, l >> 3, jj, l & 7) % 16]++;
                        cnt2[pw_unit_x5(n, l2 >> 3, jj, l2 & 7) % 16]++;
                    }
                    for (int b = 0; b < 16; ++b) { if (cnt[b] > worst) worst = cnt[b]; if (cnt2[b] > worst) worst = cnt2[b]; }
                }
    bool seen[PW_X5_UNITS] = {false};
    for (int n = 0; n < 4; ++n)
        for (int q = 0; q < 8; ++q)
            for (int jj = 0; jj < 8; ++jj)
                for (int cp = 0; cp < 8; ++cp) {
                    const int u = pw_unit_x5(n, q, jj, cp);
                    if (u < 0 || u >= PW_X5_UNITS || seen[u] || (u >> 9) != n) return 99; // slice n = wavefront n's private image
                    seen[u] = true;
                }
    return worst;
}

// ---- the four-wave kernels' input deal (fft_big_core.h deal_write_cell / deal_read_cell; fft_kernels.hip f16k_deal_load, f8k_deal_load) ----
// Runs the two rounds for all four "wavefronts" on a block of n16k = 16384 or 8192 samples and checks that every register ends up with
// sample 4 pos + q of the block; returns the number of wrong registers (0), or -1 when a cell is written twice in a round / lies outside the planes.
extern "C" long emu_four_wave_deal(int nfft, const float2 *blk)
{
    const bool big = nfft == 16384;
    const int PS = big ? F16K_PS : F8K_PS, round_samples = nfft / 2, loads = round_samples / 256, pos_per_round = round_samples / 4;
    std::vector<float2> lds((size_t)4 * PS);
    std::vector<char> seen((size_t)4 * PS);
    long wrong = 0;
    for (int r = 0; r < 2; ++r) {
        for (auto &c : seen) c = 0;
        for (int w = 0; w < 4; ++w)
            for (int lane = 0; lane < 64; ++lane)
                for (int t = 0; t < loads; ++t) {
                    const int cell = deal_write_cell(PS, w, lane, t);
                    if (cell < 0 || cell >= 4 * PS || seen[(size_t)cell]) return -1;
                    seen[(size_t)cell] = 1;
                    lds[(size_t)cell] = blk[round_samples * r + 256 * t + 64 * w + lane];
                }
        for (int q = 0; q < 4; ++q)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 4; ++i)
                    for (int j = 0; j < (big ? 16 : 8); ++j) {
                        const int pos = big ? f4k_reg_pos(i, j) : f2k_reg_pos(i, j);
                        if (pos / pos_per_round != r) continue;
                        const float2 got = lds[(size_t)deal_read_cell(PS, q, lane, pos - pos_per_round * r)], want = blk[4 * (pos + lane) + q];
                        if (memcmp(&got, &want, sizeof got)) ++wrong;
                    }
    }
    return wrong;
}

// worst number of 8-byte cells of one lane group that share a bank pair (1 = conflict free): ds_write_b64 groups of 16 consecutive lanes
// and half-waves of 32 for the deal's writes, half-waves for its reads; 32 bank pairs
extern "C" int emu_four_wave_deal_bank_conflicts(void)
{
    int worst = 1;
    for (int PS : {F16K_PS, F8K_PS})
        for (int w = 0; w < 4; ++w)
            for (int grp : {16, 32})
                for (int g0 = 0; g0 < 64; g0 += grp) {
                    int wr[32] = {0}, rd[32] = {0};
                    for (int l = g0; l < g0 + grp; ++l) { wr[deal_write_cell(PS, w, l, 3) % 32]++; rd[deal_read_cell(PS, w, l, 128) % 32]++; }
                    for (int b = 0; b < 32; ++b) { worst = wr[b] > worst ? wr[b] : worst; worst = rd[b] > worst ? rd[b] : worst; }
                }
    return worst;
}

// ---- G512: G128 on four "wavefronts" + the radix-4 stage of sub-length 128 across them (fft_big_core.h; lane program only this round) ----
namespace {
template <bool INV> void g512_tile(const float2 *in_blk, float2 *out_blk, int lgN, unsigned ctile, const float2 *Tg5, std::vector<float4> &img, std::vector<float4> &X)
{
    const long S5 = 1l << (lgN - 9), S = 4 * S5;
    const int nd = (lgN - 9) / 2; // base-4 digits of the N / 512 source columns
    std::vector<GLane> L[4];
    for (int n = 0; n < 4; ++n) {
        L[n].resize(64);
        for (int lane = 0; lane < 64; ++lane) {
            const int cp = lane & 15, q = lane >> 4;
            for (int d2 = 0; d2 < 4; ++d2)
                for (int jb = 0; jb < 8; ++jb) {
                    const float2 *p = in_blk + S5 * n + 32 * ctile + pw_g_ld(S, 0, 0, d2, jb) + pw_g_ld(S, q, cp, 0, 0);
                    L[n][lane].a[0][d2][jb] = p[0]; L[n][lane].a[1][d2][jb] = p[1];
                }
        }
        g128_stages<INV>(L[n], Tg5, img);
    }
    auto rev = [&](unsigned col) { unsigned h = 0; for (int d = 0; d < nd; ++d, col >>= 2) h = (h << 2) | (col & 3); return h; };
    const TwPairOrderedT<false> t5{Tg5 + PW_G_TABLE, 128u};
    auto round = [&](auto wr, int r, int xh) {
        for (int n = 0; n < 4; ++n)
            for (int lane = 0; lane < 64; ++lane) wr(L[n][lane].b, X.data(), lane, n);
        for (int w = 0; w < 4; ++w)
            for (int lane = 0; lane < 64; ++lane) {
                float2 v[2][4][2];
                pw_x5_read(v, X.data(), lane, w);
                const int kp = lane & 7, cg = lane >> 3, x = pw_g5_x(xh, w);
                pw_x5_stage<INV>(v, t5, pw_g5_k0(r, kp, w), 32u);
                const unsigned col = 32u * ctile + (unsigned)(cg + 8 * x);
                for (int jp = 0; jp < 2; ++jp)
                    for (int u = 0; u < 4; ++u)
                        for (int e = 0; e < 2; ++e)
                            out_blk[512l * rev(col) + 16 * r + 2 * kp + e + 32 * pw_g5_d3(w, jp) + 128 * u] = v[jp][u][e];
            }
    };
    round([](auto &b, float4 *Xi, int lane, int n) { pw_g5_write<0, 0>(b, Xi, lane, n); }, 0, 0);
    round([](auto &b, float4 *Xi, int lane, int n) { pw_g5_write<0, 1>(b, Xi, lane, n); }, 0, 1);
    round([](auto &b, float4 *Xi, int lane, int n) { pw_g5_write<1, 0>(b, Xi, lane, n); }, 1, 0);
    round([](auto &b, float4 *Xi, int lane, int n) { pw_g5_write<1, 1>(b, Xi, lane, n); }, 1, 1);
}
} // namespace

// N = 2^lgN (odd lgN >= 17): the five-stage gather pass G512 through the lane programs, then the remaining radix-4 stages with the generic
// in-place stage.  Returns the number of stages, -1 if the size does not fit.
extern "C" int emu_pair_g512_fft(int lgN, const float2 *in, float2 *out, int inverse)
{
    const unsigned N = 1u << lgN;
    FftStage st[32];
    const int ns = fft_plan_stages((int)N, st, 32);
    if (ns < 6 || st[ns - 1].p != 2 || lgN < 17) return -1;
    std::vector<float2> tw = make_tw((int)N, inverse), Tg5(PW_G5_TABLE);
    for (int i = 0; i < PW_G5_TABLE; ++i) pw_g5_table_entry(tw.data(), N, i, Tg5[i]);
    std::vector<float4> img(PW_G_UNITS), X(PW_X5_UNITS);
    for (unsigned ct = 0; ct < (N >> 14); ++ct) {
        if (inverse) g512_tile<true>(in, out, lgN, ct, Tg5.data(), img, X);
        else g512_tile<false>(in, out, lgN, ct, Tg5.data(), img, X);
    }
    for (int s = ns - 6; s >= 0; --s)
        for (unsigned b = 0; b < N / 4; ++b) {
            if (inverse) fft_stage_butterfly<true>(out, tw.data(), st[s], (int)b);
            else fft_stage_butterfly<false>(out, tw.data(), st[s], (int)b);
        }
    return ns;
}
