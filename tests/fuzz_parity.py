#!/usr/bin/env python3
"""Randomised differential run of the device kernels against the CPU oracle (bit-exact), beyond the fixed cases
of tests/: python tests/fuzz_parity.py [seconds] [seed]  (FUZZ_ONLY=5,9 restricts the run to those branches: here the two resampler ones).  Prints one line per failure and a summary."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import libredio_amd as R
import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bits = lambda a: np.ascontiguousarray(a).view(np.uint32)
fails, runs = 0, {}
SPECIALS = np.array([0.0, -0.0, 1e-40, -1e-40, 1.4e-45, 1e-30, -1e-30, 1e30, -1e30, 3e38, -3e38, np.inf, -np.inf, np.nan, 1.0, -1.0, 2.0 ** -126, 2.0 ** 127], np.float32)


def spice(x, p=0.12):
    """with probability p: signed zeros, subnormals, huge and tiny magnitudes (and, half of those times, inf / NaN) over a random share of the words"""
    if rng.random() >= p or x.size == 0:
        return x
    w = x.view(np.float32).reshape(-1)
    pool = SPECIALS if rng.random() < 0.5 else SPECIALS[np.isfinite(SPECIALS)]
    k = max(1, int(len(w) * 10.0 ** -rng.uniform(0.3, 4.0)))
    w[rng.integers(0, len(w), k)] = pool[rng.integers(0, len(pool), k)]
    runs["spiced"] = runs.get("spiced", 0) + 1
    return x


def same_nan(got, want):
    """identical bits wherever the oracle's value is not a NaN, a NaN exactly where it has one (payloads differ between x86 and gfx950)"""
    got, want = np.ascontiguousarray(got), np.ascontiguousarray(want)
    if got.shape != want.shape:
        return False
    g, w = got.view(np.float32).reshape(-1), want.view(np.float32).reshape(-1)
    wn = np.isnan(w)
    if not wn.any():
        return np.array_equal(g.view(np.uint32), w.view(np.uint32))
    return np.array_equal(np.isnan(g), wn) and np.array_equal(g.view(np.uint32)[~wn], w.view(np.uint32)[~wn])


HARD = bool(os.environ.get("FUZZ_SRC_HARD"))   # the resampler drop-in branch changes its ratio on most messages, over [1/256, 256]
ONLY = [int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v]   # e.g. FUZZ_ONLY=5,9: the two resampler branches only


def check(name, ok, detail):
    global fails
    runs[name] = runs.get(name, 0) + 1
    if not ok:
        fails += 1
        print("FAIL", name, detail, flush=True)


# FUZZ_TRACE=file: the generator's state is written there before every case, so that a case that kills the process can be replayed:
# FUZZ_STATE=file runs exactly that one case
import json
TRACE, STATE = os.environ.get("FUZZ_TRACE"), os.environ.get("FUZZ_STATE")
if STATE:
    rng.bit_generator.state = json.load(open(STATE))
    budget = 1e9
ncases = 0
t_end = time.time() + budget
while time.time() < t_end and not (STATE and ncases):
    ncases += 1
    if TRACE:
        with open(TRACE, "w") as fh:
            json.dump(rng.bit_generator.state, fh); fh.flush(); os.fsync(fh.fileno())
    which = rng.integers(0, 14) if not ONLY else int(rng.choice(ONLY))
    if which == 0:      # FIR, any K / D / length / alignment
        k = int(rng.choice([1, 2, 3, 15, 16, 17, 31, 32, 33, 63, 64, 100, 127, 128, 255, 500, int(rng.integers(1, 2000)), int(rng.integers(2000, 20000))]))
        d = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 13]))
        cplx, fused = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        n = k - 1 + int(rng.integers(0, 30000 if k < 2000 else 3000)); off = int(rng.integers(0, 4))
        taps = spice(O.synth_f32(int(rng.integers(1, 1 << 30)), 0, k), 0.05)
        x = spice((O.synth_iq if cplx else O.synth_f32)(int(rng.integers(1, 1 << 30)), 0, n + off))
        got = R.Fir(taps, d, complex_input=cplx, fused=fused)(torch.from_numpy(x).cuda()[off:]).cpu().numpy()
        want = O.fir(x[off:], taps, d, fused)
        check("fir", same_nan(got, want), (k, d, cplx, fused, n, off))
    elif which == 1:    # FFT, any size
        smooth = lambda lim: int(min(2 ** int(rng.integers(0, 15)) * 3 ** int(rng.integers(0, 9)) * 5 ** int(rng.integers(0, 6)), lim))
        n = int(rng.choice([int(rng.integers(1, 3000)), 2 ** int(rng.integers(0, 21)), 2 ** int(rng.integers(15, 19)), 3 * 2 ** int(rng.integers(0, 12)), 5 ** int(rng.integers(0, 5)) * 2 ** int(rng.integers(0, 8)),
                            smooth(16384), smooth(16384), int(rng.integers(3000, 70000))]))
        inv = bool(rng.integers(0, 2)); nb = int(rng.integers(1, 70 if n < 4000 else 4))
        x = spice(O.synth_iq(int(rng.integers(1, 1 << 30)), 0, n * nb))
        d = torch.from_numpy(x).cuda()
        plan = R.Fft(n, inv)
        got = plan(d).cpu().numpy(); want = O.fft(x, n, inv)
        ok = same_nan(got, want)
        plan(d, out=d)
        ok = ok and same_nan(d.cpu().numpy(), want)
        check("fft", ok, (n, inv, nb))
    elif which == 2:    # chain shapes
        k, dd = [(127, 5), (63, 5), (127, 3), (127, 1), (63, 1), (100, 2), (31, 4)][int(rng.integers(0, 7))]
        nfc = int(rng.choice([1024, 1024, 256, 4096, 1000, 64, 2048]))
        if HARD:  # any tap count, decimation and block size (the FIR and FFT kernels back to back through the plan's intermediate)
            k, dd, nfc = int(rng.integers(1, 400)), int(rng.integers(1, 17)), int(rng.choice([nfc, int(rng.integers(1, 3000)), 2 ** int(rng.integers(0, 15))]))
        fused = bool(rng.integers(0, 2)); nb = int(rng.integers(1, 40 if nfc <= 1024 else 6)); extra = int(rng.integers(0, nfc * dd))
        taps = O.synth_f32(int(rng.integers(1, 1 << 30)), 0, k)
        n = nb * nfc * dd + (k - dd) + extra
        x = spice(O.synth_iq(int(rng.integers(1, 1 << 30)), 0, n))
        got = R.Chain(taps, dd, nfc, fused=fused)(torch.from_numpy(x).cuda()).cpu().numpy()
        want = O.chain_fir_fft(x, taps, dd, nfc, fused=fused)
        check("chain", same_nan(got, want), (k, dd, nfc, fused, nb, extra))
        # the same plan from the receiver's u8 I/Q bytes (redio_chain_enqueue_u8), any byte alignment
        off = int(rng.integers(0, 4))
        raw = rng.integers(0, 256, 2 * n + off, dtype=np.uint8)
        got = R.Chain(taps, dd, nfc, fused=fused).from_bytes(torch.from_numpy(raw).cuda()[off:]).cpu().numpy()
        want = O.chain_fir_fft(O.data_to_samples(raw[off:]), taps, dd, nfc, fused=fused)
        check("chain_u8", got.shape == want.shape and np.array_equal(bits(got), bits(want)), (k, dd, nfc, fused, nb, extra, off))
    elif which == 3:    # overlap-save
        nfft = int(rng.choice([64, 256, 1024, 4096, 16384, 2048, 8192, 8192, 32768, 32768, 65536, 65536, 131072, 1000, int(rng.integers(2, 12000))]))  # 32768 / 65536: the three-pass tile schemes, 131072: four passes
        k = int(rng.integers(1, nfft + 1)); hop = nfft - k + 1
        n = nfft + int(rng.integers(0, 6)) * hop + int(rng.integers(0, hop))
        taps = O.synth_f32(int(rng.integers(1, 1 << 30)), 0, k)
        x = spice(O.synth_iq(int(rng.integers(1, 1 << 30)), 0, n))
        got = R.OverlapSave(taps, nfft)(torch.from_numpy(x).cuda()).cpu().numpy()
        want = O.overlap_save(x, taps, nfft)
        check("ovsave", same_nan(got, want), (nfft, k, n))
    elif which == 4:    # channelizer, any M / P
        M = int(rng.choice([64, 32, 16, 128, 256, 512, 1024, 100, 7, int(rng.integers(1, 300)), int(rng.integers(300, 9000))])); P = int(rng.choice([4, 8, 16, int(rng.integers(1, 20))]))
        fused = bool(rng.integers(0, 2)); rows = int(rng.integers(0, 200 if M < 300 else 12))
        if M in (32, 128, 256, 512, 1024) and P in (4, 8, 16): rows = int(rng.integers(0, 3000000 // M))  # the one-kernel shapes (pfb_p2_kernel): several workgroups and iterations, ragged last stream
        h = O.synth_f32(int(rng.integers(1, 1 << 30)), 0, M * P)
        x = spice(O.synth_iq(int(rng.integers(1, 1 << 30)), 0, M * (P - 1 + rows) + int(rng.integers(0, M))))
        got = R.Channelizer(h, M, P, fused=fused)(torch.from_numpy(x).cuda()).cpu().numpy()
        want = O.pfb_channelizer(x, h, M, P, fused)
        check("pfb", same_nan(got, want), (M, P, fused, rows))
        g = int(rng.choice([2, 4, 8, M]))
        if g > 1 and M % g == 0 and rows > 0:  # the per-destination layout the exchange sends: [group][row][M / g]
            grp = R.Channelizer(h, M, P, fused=fused)(torch.from_numpy(x).cuda(), ngroups=g).cpu().numpy()
            check("pfb_grouped", same_nan(grp.transpose(1, 0, 2).reshape(want.shape), want), (M, P, fused, rows, g))
        off = int(rng.integers(0, 4))   # the same plan from u8 I/Q bytes (redio_pfb_enqueue_u8), any byte alignment
        raw = rng.integers(0, 256, 2 * len(x) + off, dtype=np.uint8)
        got = R.Channelizer(h, M, P, fused=fused).from_bytes(torch.from_numpy(raw).cuda()[off:]).cpu().numpy()
        want = O.pfb_channelizer(O.data_to_samples(raw[off:]), h, M, P, fused)
        check("pfb_u8", got.shape == want.shape and np.array_equal(bits(got), bits(want)), (M, P, fused, rows, off))
        if g > 1 and M % g == 0 and rows > 0:  # ... and into the grouped layout (64 x 16, 4-byte aligned: two rows per load instruction)
            grp = R.Channelizer(h, M, P, fused=fused).from_bytes(torch.from_numpy(raw).cuda()[off:], ngroups=g).cpu().numpy()
            check("pfb_u8_grouped", np.array_equal(bits(np.ascontiguousarray(grp.transpose(1, 0, 2)).reshape(want.shape)), bits(want)), (M, P, fused, rows, g, off))
    elif which == 6:    # ingest: bytes -> samples -> |x| -> block sums -> slicer, any length / offset
        from libredio_amd import bitfount as B
        n = int(rng.integers(1, 60000)); off = 8 * int(rng.integers(0, 3))
        raw = rng.integers(0, 256, 2 * n + off, dtype=np.uint8)
        d = torch.from_numpy(raw).cuda()[off:]
        xs = O.data_to_samples(raw[off:])
        ok = np.array_equal(bits(B.data_to_samples(d).cpu().numpy()), bits(xs))
        mag = O.norm(xs)
        ok = ok and np.array_equal(bits(B.norm(torch.from_numpy(xs).cuda()).cpu().numpy()), bits(mag))
        blk = int(rng.choice([512, 64, 100, 7]))
        nb = n // blk
        if nb:
            got = B.block_sums(torch.from_numpy(mag[: nb * blk].copy()).cuda(), blk).cpu().numpy()
            want = np.array([O.block_sum(mag[b * blk:(b + 1) * blk]) for b in range(nb)], np.float32)
            ok = ok and np.array_equal(bits(got), bits(want))
        o2 = int(rng.integers(0, 4))
        dm = torch.from_numpy(mag).cuda()[o2:]
        if dm.numel():
            ok = ok and np.array_equal(B.discretize(dm).cpu().numpy(), O.discretize(mag[o2:]).astype(np.uint8))
        check("ingest", ok, (n, off, blk, o2))
    elif which == 7:    # runs and vector maps
        from libredio_amd import kpn_dev as K
        n = int(rng.integers(1, 50000)); off = int(rng.integers(0, 9))
        v = np.repeat(rng.integers(0, 4, n), rng.integers(1, 6, n)).astype(np.uint8)[: n + off]
        off = min(off, len(v))
        dev, ref = K.Rle(), O.Rle()
        ok = True
        cuts = sorted(set([off, len(v)] + [int(c) for c in rng.integers(off, len(v) + 1, 2)]))
        dv = torch.from_numpy(v).cuda()
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            a, c = dev.feed(dv[lo:hi])
            ok = ok and list(zip(a.cpu().tolist(), c.cpu().tolist())) == ref.feed(v[lo:hi])
        cplx = bool(rng.integers(0, 2)); m = int(rng.integers(1, 5000)); o3 = int(rng.integers(0, 3))
        gen = O.synth_iq if cplx else O.synth_f32
        x, c = gen(int(rng.integers(1, 1 << 30)), 0, m + o3), gen(int(rng.integers(1, 1 << 30)), 0, m + 3)
        dx, dc = torch.from_numpy(x).cuda()[o3:], torch.from_numpy(c).cuda()
        ok = ok and np.array_equal(bits(K.mul_vecs(dx, dc).cpu().numpy()), bits(O.zip_vecs(x[o3:], c, add=False)))
        ok = ok and np.array_equal(bits(K.sum_vecs(dx, dc).cpu().numpy()), bits(O.zip_vecs(x[o3:], c, add=True)))
        check("runs", ok, (n, off, cplx, m, o3))
    elif which == 8:    # the host-buffer drop-ins: kiss_fft (zero-copy and copy paths) and convolve (per-thread cache)
        from libredio_amd import kissfft, dsputils
        n = int(rng.choice([int(rng.integers(1, 3000)), 2 ** int(rng.integers(0, 15)), 8192, 8193, 16384]))
        inv = int(rng.integers(0, 2))
        cfg = kissfft.Cfg(n, inv)
        ok = True
        for _ in range(3):
            x = O.synth_iq(int(rng.integers(1, 1 << 30)), 0, n)
            ok = ok and np.array_equal(bits(cfg(x)), bits(O.fft(x, n, bool(inv))))
        cfg.close()
        nu, nv = int(rng.integers(1, 100000)), int(rng.choice([1, 3, 63, 64, 127, int(rng.integers(1, 400))]))
        u, v = O.synth_f32(int(rng.integers(1, 1 << 30)), 0, nu), O.synth_f32(int(rng.integers(1, 1 << 30)), 0, nv)
        got, want = dsputils.convolve(u, v), O.convolve(u, v)
        ok = ok and got.shape == want.shape and np.array_equal(bits(got), bits(want))
        check("dropin", ok, (n, inv, nu, nv))
    elif which in (10, 11):   # carried-history streams: any plan, any cut of the stream == one stateless call on the whole stream
        kind = int(rng.integers(0, 6))
        u8 = kind >= 4            # 4, 5: the chain / channelizer streams fed with u8 I/Q bytes
        if u8:
            kind -= 3             # -> 1 (chain), 2 (pfb)
        if kind == 0:
            k = int(rng.choice([1, 2, 17, 63, 127, 128, int(rng.integers(1, 600))])); d = int(rng.choice([1, 2, 3, 5, 8, 13, int(rng.integers(1, 300))]))
            cplx, fused = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
            taps = O.synth_f32(int(rng.integers(1, 1 << 30)), 0, k)
            plan = R.Fir(taps, d, complex_input=cplx, fused=fused)
            one = lambda v: O.fir(v, taps, d, fused)
            n = int(rng.integers(0, 60000)); desc = ("fir", k, d, cplx, fused)
        elif kind == 1:
            k, d = [(127, 5), (63, 5), (127, 3), (63, 1), (31, 4), (100, 2)][int(rng.integers(0, 6))]
            nfc = int(rng.choice([1024, 1024, 256, 64])); fused = bool(rng.integers(0, 2)); cplx = True
            taps = O.synth_f32(int(rng.integers(1, 1 << 30)), 0, k)
            plan = R.Chain(taps, d, nfc, fused=fused)
            one = lambda v: O.chain_fir_fft(v, taps, d, nfc, fused=fused).reshape(-1)
            n = int(rng.integers(0, 12 * nfc * d)); desc = ("chain", k, d, nfc, fused)
        elif kind == 2:
            M = int(rng.choice([64, 32, 100, 7])); P = int(rng.choice([4, 8, 16, 5])); fused = bool(rng.integers(0, 2)); cplx = True
            h = O.synth_f32(int(rng.integers(1, 1 << 30)), 0, M * P)
            plan = R.Channelizer(h, M, P, fused=fused)
            one = lambda v: O.pfb_channelizer(v, h, M, P, fused).reshape(-1)
            n = int(rng.integers(0, M * 400)); desc = ("pfb", M, P, fused)
        else:
            nfft = int(rng.choice([64, 256, 1024, 4096, 1000, 2048])); k = int(rng.integers(1, nfft + 1)); cplx = True
            taps = O.synth_f32(int(rng.integers(1, 1 << 30)), 0, k)
            plan = R.OverlapSave(taps, nfft)
            one = lambda v: O.overlap_save(v, taps, nfft)
            n = int(rng.integers(0, nfft + 8 * (nfft - k + 1))); desc = ("ovsave", nfft, k)
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, int(rng.integers(0, 12)))]))
        if u8:
            raw = rng.integers(0, 256, 2 * n, dtype=np.uint8)
            x = O.data_to_samples(raw) if n else np.zeros(0, np.complex64)
            want = one(x) if n else np.zeros(0, x.dtype)
            st = R.Stream(plan, u8=True)
            draw = torch.from_numpy(raw).cuda() if n else torch.zeros(0, dtype=torch.uint8, device="cuda")
            outs = [st(draw[2 * lo: 2 * hi]).clone() for lo, hi in zip(cuts[:-1], cuts[1:])]
            desc = desc + ("u8",)
        else:
            x = spice((O.synth_iq if cplx else O.synth_f32)(int(rng.integers(1, 1 << 30)), 0, max(n, 1))[:n])
            want = one(x) if n else np.zeros(0, x.dtype)
            st = R.Stream(plan)
            dx = torch.from_numpy(x).cuda() if n else torch.zeros(0, dtype=torch.complex64 if cplx else torch.float32, device="cuda")
            outs = [st(dx[lo:hi]).clone() for lo, hi in zip(cuts[:-1], cuts[1:])]
        got = torch.cat(outs).cpu().numpy() if outs else np.zeros(0, x.dtype)
        check("stream", same_nan(got, np.asarray(want).reshape(-1)), desc + (n, cuts))
    elif which == 9:    # src_process drop-in (host buffers, one state, random messages)
        from libredio_amd import samplerate
        conv = int(rng.integers(0, 5)); ch = int(rng.choice([1, 1, 2, 3]))
        ratio = float(rng.choice([0.02, 0.5, 1.0, 2.0, 0.25, 1.0884, 48000 / 44100, 1.5, 0.3, float(rng.uniform(0.01, 3.0))]))
        st, ref, ok = samplerate.State(conv, ch), O.Resampler(conv, ch), True
        nmsg = int(rng.integers(1, 5)); flush = bool(rng.integers(0, 2)); sizes = []; seeds = []; detail = []
        for i in range(nmsg):
            m = int(rng.integers(1, 20000)) if rng.integers(0, 4) else int(rng.integers(1, 6))   # now and then a message of a few frames
            sd = int(rng.integers(1, 1 << 30)); seeds.append(sd)
            x = O.synth_f32(sd, 0, m * ch)
            eoi = int(flush and i + 1 == nmsg)                      # the last message may carry end_of_input: the converter drains its tail
            if HARD and i and rng.integers(0, 4):                   # FUZZ_SRC_HARD=1: most messages change the ratio, by up to 30 x, over the library's whole range
                ratio = float(np.clip(ratio * 10.0 ** rng.uniform(-1.5, 1.5), 1 / 256, 256.0))
                if ratio > 8.0: m = min(m, 400)
            elif i and not rng.integers(0, 3):                      # a new ratio now and then: the library glides to it inside the message
                ratio = float(np.clip(ratio * rng.uniform(0.5, 2.0), 0.01, 3.0))
            cap = int(ratio * m + 1.0) + (int(rng.integers(0, 6000)) if eoi else 0)
            stepped = 0
            if i and not rng.integers(0, 5):                        # src_set_ratio before the call: a step to the new ratio instead of a glide (samplerate.rs:40)
                sr = ratio if rng.integers(0, 2) else float(np.clip(ratio * rng.uniform(0.5, 2.0), 0.01, 3.0))
                stepped = 1 if (st.set_ratio(sr), ref.set_ratio(sr)) == (0, 0) else -1
            e1, a, u1 = st.process(x, ratio, cap, eoi)
            e2, b, u2 = ref.process(x, ratio, cap, bool(eoi))
            same = (e1, u1, len(a)) == (e2, u2, len(b)) and np.array_equal(bits(a), bits(b))
            if not same:  # everything needed to replay the message sequence: per message (frames, seed, ratio as hex, capacity, eoi), device / oracle (error, used, generated), first differing output
                nd = np.nonzero(bits(a[: min(len(a), len(b))]) != bits(b[: min(len(a), len(b))]))[0]
                detail.append((i, (e1, u1, len(a)), (e2, u2, len(b)), int(nd[0]) if len(nd) else -1, len(nd)))
            ok = ok and same
            ok = ok and stepped >= 0
            sizes.append((m, sd, float(ratio).hex(), cap, eoi) + ((float(sr).hex(),) if stepped else ()))
        st.close()
        check("srcdrop", ok, (conv, ch, sizes, flush, detail))
    elif which == 12:   # bitfount::trigger (bitfount.rs:36-85): quiet noise with bursts, any cut of the block stream into calls
        from libredio_amd import bitfount as B
        bl = int(rng.choice([512, 512, 512, 64, 100, 1])); nb = int(rng.integers(1, 2500 if bl >= 64 else 6000))
        amp = float(10.0 ** rng.uniform(-3, 0.5))
        blocks = (amp * rng.random((nb, bl))).astype(np.float32)
        if rng.integers(0, 4) == 0: blocks -= np.float32(amp / 2)   # sums of both signs
        for _ in range(int(rng.integers(0, 8))):
            st0 = int(rng.integers(0, nb)); ln = int(rng.choice([1, 2, 3, 10, 49, 50, 51, 60, 200]))
            blocks[st0:st0 + ln] += np.float32(amp * 10.0 ** rng.uniform(0, 2))
        cuts = sorted(set([0, nb] + [int(c) for c in rng.integers(0, nb + 1, int(rng.integers(0, 5)))]))
        dev, ref = B.Trigger(), O.Trigger()
        got, want = [], []
        dblocks = torch.from_numpy(blocks).cuda()
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if hi > lo:
                got += [g.cpu().numpy() for g in dev.feed(dblocks[lo:hi].contiguous())]
                want += ref.feed(blocks[lo:hi])
        ok = len(got) == len(want) and all(len(g) == len(w) and np.array_equal(bits(g), bits(w)) for g, w in zip(got, want))
        check("trigger", ok, (bl, nb, amp, cuts, len(got), len(want)))
    elif which == 13:   # REDIO_SRC_FAST (opt-in f32 mode) against EXACT on the device: integer decimations 2 ... 64 (25 ... 53: the round-5 kernel with its two
                        # image halves, straddling tiles and patch path), any channel count, any cut into messages: same frame counts, values inside the f32 bound
        S = int(rng.choice([50, 50, 25, 26, 32, 33, 40, 41, 48, 49, 53, int(rng.integers(2, 65)), int(rng.integers(25, 54))]))
        conv = int(rng.choice([1, 1, 2])); nch = int(rng.choice([1, 2, 3, 7, int(rng.integers(1, 40))]))
        ratio = 1.0 / S
        n = int(rng.integers(S * 40, S * 40 + 60000))
        x = np.stack([O.synth_f32(int(rng.integers(1, 1 << 30)), 0, n) for _ in range(nch)])
        tab, half, inc = O.src_table(conv)
        pos = np.arange(0.0, half, inc * ratio)
        sum_h = 2 * ratio * np.abs(np.interp(pos, np.arange(half + 2), tab.astype(np.float64))).sum()
        bound = (2 * len(pos) + 1) * 2.0 ** -24 * max(sum_h, 1.0) * float(np.abs(x).max())
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, int(rng.integers(0, 5)))]))
        exact, fast = R.Src(nch, conv), R.Src(nch, conv, mode=R.Src.FAST)
        dx = torch.from_numpy(x).cuda()
        ok, worst = True, 0.0
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if hi == lo: continue
            a, ua = exact.process(dx[:, lo:hi].contiguous(), ratio)
            b, ub = fast.process(dx[:, lo:hi].contiguous(), ratio)
            ok = ok and ua == ub and a.shape == b.shape
            if ok and a.numel():
                worst = max(worst, float((a - b).abs().max().item()))
                ok = ok and worst <= bound
        check("srcfast", ok, (S, conv, nch, n, cuts, worst, bound))
    else:               # resampler, batched, random ratio and message cuts
        nch = int(rng.choice([1, 3, 40, int(rng.integers(1, 100))])); conv = int(rng.integers(0, 5))
        ratio = float(rng.choice([0.02, 0.5, 1.0, 0.25, 0.1, 2.0, 0.0213, 1.0884, 48000 / 44100, 1.5, 0.3, 4 / 3, 0.75, 1 / 7, float(rng.uniform(0.01, 3.0)), 1 / 256, 256.0, 100.0, 0.004]))
        n = int(rng.integers(1, 40000 if ratio < 10 else 400))
        x = np.stack([O.synth_f32(int(rng.integers(1, 1 << 30)), 0, n) for _ in range(nch)])
        plan = R.Src(nch, conv, mode=int(rng.choice([0, 2])))
        refs = [O.Resampler(conv) for _ in range(nch)]
        cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, 3)] + ([min(n, int(rng.integers(0, n + 1)) + 1)] if rng.integers(0, 2) else [])))
        ok = True
        dx = torch.from_numpy(x).cuda()
        flush = bool(rng.integers(0, 2)); ratios = []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            eoi = flush and hi == n
            if HARD and lo and rng.integers(0, 2):
                ratio = float(np.clip(ratio * 10.0 ** rng.uniform(-1.0, 1.0), 1 / 256, 256.0 if n < 400 else 8.0))
            ratios.append(float(ratio).hex())
            cap = int(ratio * (hi - lo) + 1.0) + (int(rng.integers(0, 6000)) if eoi else 0)
            try:
                a, used = plan.process(dx[:, lo:hi].contiguous(), ratio, output_frames=cap, end_of_input=eoi)
                code = 0
            except R.RedioError as ex:   # a library error code (include/samplerate.h (iv)): the oracle must report the same, then the stream is over
                code = ex.code
            if code:
                ok = ok and all(refs[c].process(x[c, lo:hi], ratio, cap, eoi)[0] == code for c in range(nch))
                break
            a = a.cpu().numpy()
            for c in range(nch):
                err, want, wused = refs[c].process(x[c, lo:hi], ratio, cap, eoi)
                ok = ok and err == 0 and wused == used and a.shape[1] == len(want) and np.array_equal(bits(a[c]), bits(want))
        check("src", ok, (nch, conv, ratios, n, cuts, flush))
print("runs", runs, "failures", fails)
sys.exit(1 if fails else 0)
