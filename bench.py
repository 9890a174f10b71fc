#!/usr/bin/env python3
"""bench.py -- the north-star chain on MI355X: cf32 IQ -> 127-tap FIR, decimate by 5 -> 1024-pt FFT.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the fused chain kernel (redio_chain_enqueue) over one resident buffer of
2^28 synthetic cf32 samples per GPU (BASELINE.json configs[1]).  Inputs are generated on the device
by the integer hash of SURVEY.md 8d before the timed region.  For N > 1 the driver starts one process
per GPU (torch.distributed.run); the stream is time-sliced, every rank owns its own 2^28-sample slice
(with the 126-sample FIR halo inside the slice), there is no data-path collective, and value is the
samples of all ranks over the slowest rank's time ("weak" scaling).

Clock pre-conditioning (disclosed in the line as `precondition`; `--no-precondition` turns it off and reproduces the cold figure):
the chip raises its shader clock over the first ~100 launches of a burst (profiles/r02_clock_probe.txt), and SURVEY.md 8d asks for
"timed repetitions after warm-up", so before the W warm-up launches every rank repeats the SAME launch on the SAME buffers until
the per-launch HIP-event time has converged -- the medians of the last three windows of 20 launches within 1 % of each other -- or
400 launches have run (~0.25 s).  Same kernel, same 2^28 samples, nothing skipped in the timed region; W and K keep their meaning.

Rank 0 prints ONE JSON line.  `value` is exactly what the flags time (W warm-up launches, then K launches between
two barriers).  `roofline` is for the dominant (only) kernel, chain_v4_kernel: algorithmic bytes = 9.6 B per input
sample (8 B read + 8/5 B written) over the kernel's mean duration in the timed region, measured with HIP
events on the launch stream; `launch_ms_series` are those per-launch times.  The chip raises its clock over the first
~100 launches of a burst, so after the headline the same launch is repeated `--steady` more times (default 1000) and
reported as `steady_state` with its own kernel time and fraction -- never as `value`.  `roofline.traffic` comes from
the rocprofv3 PMC passes of a separate run (`traffic_source`); `roofline.valu` prices the same launch against the f32 vector peak
(report min(HBM, VALU)); `stages` times the FIR alone and the transform alone; for N > 1 `ranks_seen` lists the
world size and every rank's device as torch.distributed saw them.  `cpu_baseline` is the CPU oracle (oracle/, a port of
the reference's algorithm) timed on this host on a bounded sample: all cores as independent replicas (`value`), one
core, and the reference's own structure -- a thread per block with a heap Vec per message (`kpn_pipeline`).

`value_cold` is the figure rounds 1-3 reported under the same flags (launches W .. W+K-1 of the cold burst of this same run, from the
pre-conditioning's own per-launch HIP events), so that the line stays comparable round over round.  `other_configs` (rank 0, after
the timed region, `--no-other-configs` skips it, about 1 s): BASELINE.json configs[2] (256 channels x 2^22 frames, exact and the opt-in
f32 mode), configs[3] on one GPU (cf32 and u8 input), configs[4] on one GPU (65536-point blocks, 8193 taps) and the 65536-point
transform alone, each {workload, alg_bytes, kernel_ms by HIP events after >= 100 ms of warm-up, frac of 8 TB/s, binding, kernel,
cpu_baseline = the oracle's restatement of that config on this host, one core and all cores on a stated prefix}; `kpn_graph_c2` = the
chain behind the operator API (include/kpn_dev.hpp: one thread per block, messages through channels, bounded rings) against the bare
plan launches, messages of 2^13 ... 2^28 samples, with the round-5 host path beside it (`tests/_build/kpn_tests bench_c2_list`, a child
process; `.other_blocks` = dev::channelizer, dev::overlap_save, dev::fft, dev::fir the same way, `kpn_tests bench_block_list`); `dropin_calls` = microseconds per call of the host-buffer drop-ins (kiss_fft, src_process, redio_convolve_f32) with the
oracle's CPU microseconds for the same call beside them.  `--no-graph-leg` skips those two (about 35 s).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NTAPS, DECIM, NFFT, FC = 127, 5, 1024, 0.08
ALG_BYTES_PER_SAMPLE = 8.0 + 8.0 / DECIM  # cf32 in once + decimated cf32 spectra out once
HBM_PEAK_GBS = 8000.0                     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SEED = 0x5EED0002                         # 0x5EED0000 + config id


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-u8-leg", action="store_true", help="skip the from_u8_bytes leg (the same chain fed with u8 I/Q bytes, after the timed region)")
    ap.add_argument("--steady", type=int, default=1000, help="further launches after the timed region, reported as steady_state (0 = skip)")
    ap.add_argument("--log2-samples", type=int, default=28, help="input samples per GPU (default 2^28 = 2 GiB)")
    ap.add_argument("--unfused", action="store_true", help="run FIR and FFT as two kernels (12.8 B/sample)")
    ap.add_argument("--exact", action="store_true", help="reference rounding (mul+add) instead of fmaf in the FIR")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the other_configs leg (BASELINE.json configs[2..4] and the 65536-point "
                                                                    "transform, each timed with HIP events after the timed region; rank 0 only)")
    ap.add_argument("--no-graph-leg", action="store_true", help="skip other_configs.kpn_graph_c2 and other_configs.dropin_calls")
    ap.add_argument("--moved-json", default=os.path.join(ROOT, "profiles", "r06_moved_bytes.json"), help="file with the PMC-measured HBM bytes per call of the "
                                                                                                       "multi-pass configs (C5, 65536-point transform), keyed by kernel name")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                                                      "multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--cpu-log2-samples", type=int, default=23, help="slice per CPU thread")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "r06_traffic.json"), help="file with {'traffic': bytes_per_launch, 'kernel': name} from the PMC passes")
    ap.add_argument("--no-precondition", action="store_true", help="skip the clock pre-conditioning launches (the cold-burst figure of rounds 1-3)")
    ap.add_argument("--precondition-max", type=int, default=400, help="cap on the pre-conditioning launches")
    return ap.parse_args()


def cpu_baseline(log2n, min_seconds=12.0):
    """The oracle port of the reference path on the host cores of this box: one thread per core (the C calls
    release the GIL), each running the scalar strict-order FIR + kissfft restatement over its own 2^log2n-sample
    slice of the same hash-generated stream (FIR halo included) again and again for ~min_seconds of wall time.
    `value` is the aggregate; the one-thread rate is quoted in `sample`."""
    import threading
    import oracle as O
    n = 1 << log2n
    taps = O.lpf_corrected(NTAPS, FC)
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    O.chain_fir_fft(O.synth_iq(SEED, 0, 8192 * DECIM + NTAPS), taps, DECIM, NFFT)  # page in / warm
    # one thread first: the single-core rate (about a quarter of the budget)
    x0 = O.synth_iq(SEED, 0, n)
    used1, busy1 = 0, 0.0
    while busy1 < min_seconds / 4:
        t0 = time.perf_counter()
        out = O.chain_fir_fft(x0, taps, DECIM, NFFT, fused=False)
        busy1 += time.perf_counter() - t0
        used1 += out.shape[0] * NFFT * DECIM
    single = used1 / busy1 / 1e6
    if cores == 1:
        total, wall = used1, busy1
    else:
        xs = [x0] + [O.synth_iq(SEED, t * n, n) for t in range(1, cores)]  # generation is not timed
        done = [0] * cores
        start = threading.Barrier(cores + 1)
        deadline = [0.0]

        def work(t):
            start.wait()
            while time.perf_counter() < deadline[0]:
                out = O.chain_fir_fft(xs[t], taps, DECIM, NFFT, fused=False)
                done[t] += out.shape[0] * NFFT * DECIM

        th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
        for t in th:
            t.start()
        deadline[0] = time.perf_counter() + min_seconds * 0.75
        t0 = time.perf_counter()
        start.wait()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        total = sum(done)
    # the reference's own structure: one thread per block, one heap Vec per message, queue hand-off
    kdone, kmsgs, kwall = O.kpn_chain_baseline(min_seconds / 2, 22, SEED, taps, DECIM, NFFT)
    return {"value": total / wall / 1e6, "unit": "MSamples/s", "cores": cores, "kind": "port",
            "sample": f"{total} samples in {wall:.1f} s: {cores} threads (one per host core), each looping over its own 2^{log2n}-sample "
                      f"slice of the same hash-generated stream through the oracle/ C port (scalar strict-order FIR + kissfft "
                      f"restatement, gcc -O2, no FMA); one thread alone: {single:.1f} MSamples/s",
            "single_core": {"value": single, "unit": "MSamples/s", "cores": 1},
            "kpn_pipeline": {"value": kdone / kwall / 1e6, "unit": "MSamples/s", "cores": 4, "block_threads": 3,
                             "sample": f"{kmsgs} messages ({kdone} input samples) in {kwall:.1f} s through source -> [fir 127 taps, keep every "
                                       f"5th] -> [kiss_fft 1024] -> sink, one OS thread per block, one heap Vec per message, mutex/condvar "
                                       f"queues (oracle/kpn_baseline.cpp; the structure of src/kissfft/src/kissfft.rs:18-31 and "
                                       f"src/ratpak.rs:60-185), CPU restatement of the reference"}}


def _cpu_rate(make_call, units_per_call, seconds):
    """(one-core units/s, all-core units/s, cores): `make_call()` returns a closure that runs the oracle once on private data (the C calls
    release the GIL); one thread for ~seconds / 2, then one thread per host core for ~seconds / 2."""
    import threading
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 64))
    f = make_call()
    f()
    t0, n1 = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds / 2:
        f(); n1 += 1
    one = n1 * units_per_call / (time.perf_counter() - t0)
    if cores == 1:
        return one, one, 1
    calls = [f] + [make_call() for _ in range(cores - 1)]
    done = [0] * cores
    start = threading.Barrier(cores + 1)
    deadline = [0.0]

    def work(t):
        start.wait()
        while time.perf_counter() < deadline[0]:
            calls[t](); done[t] += 1

    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    for t in th:
        t.start()
    deadline[0] = time.perf_counter() + seconds / 2
    t0 = time.perf_counter()
    start.wait()
    for t in th:
        t.join()
    return one, sum(done) * units_per_call / (time.perf_counter() - t0), cores


def cpu_baseline_configs(seconds_each=3.0):
    """The oracle's restatement of BASELINE.json configs[2..4] on this host's cores (north_star: "the reference Rust+kissfft+libsamplerate path
    timed on the same host cores ... in the same run"): one core, and one thread per core as independent replicas, each on a stated prefix
    of the config's workload.  Reported beside the GPU figure, never a target."""
    import oracle as O
    out = {}

    def rec(one, allc, cores, unit, sample):
        return {"value": allc / 1e6, "unit": unit, "cores": cores, "kind": "port", "single_core": {"value": one / 1e6, "unit": unit, "cores": 1}, "sample": sample}

    # configs[2]: libsamplerate's medium sinc converter at ratio 0.02, one mono channel per state (samplerate.rs:59-87), messages of 2^16 frames
    def c3():
        st, x = O.Resampler(1), O.synth_f32(100, 0, 1 << 16)
        st.block(x, 0.02)
        return lambda: st.block(x, 0.02)
    one, allc, cores = _cpu_rate(c3, 1 << 16, seconds_each)
    out["c3"] = rec(one, allc, cores, "MSamples/s (input)", "oracle/oracle_src.c (orc_src_process: libsamplerate 0.1.8's sinc converter restated, medium quality, ratio 0.02), "
                    "one converter state per thread fed 2^16-frame messages of one channel again and again; the config has 256 such channels")

    # configs[3]: 64 channels x 16 taps per branch, 2^18 samples per call
    h4 = O.lpf_corrected(1024, 0.45 / 64)
    def c4():
        x = O.synth_iq(0x5EED0004, 0, 1 << 18)
        return lambda: O.pfb_channelizer(x, h4, 64, 16, True)
    one, allc, cores = _cpu_rate(c4, 1 << 18, seconds_each)
    out["c4"] = rec(one, allc, cores, "MSamples/s", "oracle/oracle_dsp.c (orc_pfb_channelizer: branch folds of dsputils.rs:31 + the kissfft restatement across 64 branches), "
                    "2^18-sample slices of the hash-generated stream per call")

    # configs[4]: overlap-save, 65536-point blocks, 8193 taps; 4 blocks per call
    h5 = O.lpf_corrected(8193, 0.08)
    hop = 65536 - 8193 + 1
    def c5():
        x = O.synth_iq(0x5EED0005, 0, 65536 + 3 * hop)
        return lambda: O.overlap_save(x, h5, 65536)
    one, allc, cores = _cpu_rate(c5, 4 * hop, seconds_each)
    out["c5"] = rec(one, allc, cores, "MSamples/s (output)", "oracle/oracle_dsp.c (orc_overlap_save over the kissfft restatement: 65536-point blocks, 8193 taps), 4 blocks per call")

    # the 65536-point transform alone
    def f64k():
        x = O.synth_iq(2, 0, 4 * 65536)
        return lambda: O.fft(x, 65536)
    one, allc, cores = _cpu_rate(f64k, 4 * 65536, seconds_each / 2)
    out["fft_65536"] = rec(one, allc, cores, "MSamples/s", "oracle/oracle_kiss.c (kissfft 1.3.0 restated, 4^8 butterflies), 4 transforms per call")
    return out


def kpn_graph_leg(cpu_kpn_msps, timeout=90):
    """other_configs.kpn_graph_c2: the chain behind the operator API.  Runs tests/_build/kpn_tests bench_c2_list in a child process (this
    process only holds memory meanwhile): source (4 resident messages, cycled) -> dev::fir_fft_chain -> sink, one OS thread per block, against
    the same launches made bare from one thread; see the header of bench_c2 in tests/cpp/kpn_tests.cpp."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "_build", "kpn_tests")
    if not os.path.exists(exe):
        return {"error": f"{os.path.relpath(exe, ROOT)} is missing (built by __graft_entry__.build())"}
    shipped = [(13, 12000), (16, 12000), (20, 10000), (22, 5000), (24, 2000), (26, 1000), (28, 340)]
    specs = [f"{k}:{n}:4:0:0:resident:checksum" for k, n in shipped]
    specs += [f"{k}:{n}:4:0:0:resident:drop" for k, n in ((24, 4000), (28, 340))]
    specs += [f"{k}:{n}:0:1:1:resident:checksum" for k, n in ((16, 4000), (24, 200), (28, 28))]          # the round-5 host path
    specs += ["28:200:4:0:0:synth:drop"]                                                                     # input generated per message
    specs += [f"{k}:{n}:4:0:0:carried:checksum" for k, n in ((16, 8000), (24, 2000), (28, 340))]             # the chain as a STREAM: history carried
    t0 = time.perf_counter()
    try:
        p = subprocess.run([exe, "bench_c2_list"] + specs, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired:
        return {"error": f"kpn_tests bench_c2_list did not finish in {timeout} s"}
    pts = []
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            pts.append(json.loads(line))
    if p.returncode != 0 or not pts:
        return {"error": f"kpn_tests bench_c2_list rc={p.returncode}: {p.stderr[-400:]}"}

    def slim(r):
        return {"log2_msg": r["log2_msg"], "messages": r["messages"], "graph_gsps": r["graph_gsps"], "bare_gsps": r["bare_gsps"], "frac_of_bare": r["frac_of_bare"],
                "frac_of_bare_chain_only": r["frac_of_bare_chain_only"], "graph_us_per_msg": r["graph_us_per_msg"], "bare_us_per_msg": r["bare_us_per_msg"],
                "mallocs_in_timed_region": r["mallocs_in_timed_region"]}

    # the other hot blocks through the same machinery (kpn_tests bench_block): source -> block -> checksum sink against the bare launches
    blocks = []
    bspecs = ["channelizer:24:2000", "channelizer:28:200", "ovsave:24:800", "ovsave:28:100", "fft:24:2000", "fft:28:300", "fir:24:2000", "fir:28:300"]
    try:
        pb = subprocess.run([exe, "bench_block_list"] + bspecs, capture_output=True, text=True, timeout=timeout)
        for line in pb.stdout.splitlines():
            if line.startswith("{"):
                r = json.loads(line)
                blocks.append({k: r[k] for k in ("block", "msg_samples", "messages", "graph_gsps", "bare_gsps", "frac_of_bare", "graph_us_per_msg", "bare_us_per_msg",
                                                 "mallocs_in_timed_region")})
        if pb.returncode != 0:
            blocks.append({"error": f"kpn_tests bench_block_list rc={pb.returncode}: {pb.stderr[-300:]}"})
    except subprocess.TimeoutExpired:
        blocks.append({"error": f"kpn_tests bench_block_list did not finish in {timeout} s"})

    for r in pts:
        r.setdefault("history", "per_message")
    sel = lambda **kw: [slim(r) for r in pts if all(r[k] == v for k, v in kw.items())]
    ship = sel(ring=4, host_sync=0, source="resident", sink="checksum", history="per_message")
    over = [r["log2_msg"] for r in ship if r["graph_gsps"] * 1e3 > cpu_kpn_msps] if cpu_kpn_msps else []
    return {"workload": "BASELINE.json configs[1] behind the operator API (include/kpn_dev.hpp; kpn.rs:278-291, kissfft.rs:18-31, ratpak.rs:60-185): source -> "
                        "dev::fir_fft_chain (127 taps / 5 -> 1024-point transform) -> sink, one OS thread per block, messages through channels, bounded rings of 4 "
                        "buffers per block, compute blocks on the shared graph stream; GS/s of input samples that reach a spectrum; bare = the same launches "
                        "(redio_chain_enqueue [+ redio_checksum_u32]) back to back from one thread on one stream, same buffers, same process",
            "checksum_sink": ship, "drop_sink": sel(ring=4, host_sync=0, source="resident", sink="drop"),
            "carried_history_stream_block": sel(history="carried"),
            "round5_host_path": sel(ring=0, host_sync=1),
            "synth_source_drop_sink": sel(source="synth"),
            "other_blocks": blocks,
            "overtakes_cpu_kpn_pipeline_from_log2_msg": min(over) if over else None,
            "cpu_kpn_pipeline_msps": cpu_kpn_msps,
            "note": "frac_of_bare: graph rate / bare rate for the same work (chain + sink kernel); frac_of_bare_chain_only: against the chain launches alone "
                    "(the checksum sink reads every spectrum word once more: 1.6 of 11.2 B per sample).  round5_host_path = no pool (hipMalloc + hipFree per message), "
                    "hipStreamSynchronize before every send, a stream per block.  synth_source: every message generated afresh in the source block "
                    "(8 more bytes per sample through HBM).  carried_history_stream_block: dev::fir_fft_chain_stream (redio_chain_stream_*: the unconsumed tail of "
                    "every message stays on the device, SURVEY.md 8d C2 'history carried') against redio_chain_stream_enqueue + checksum launched bare; every "
                    "sample of a message counts.  A message of 2^13 samples holds one 1024-point spectrum.  other_blocks: dev::channelizer (64 x 16, configs[3]), "
                    "dev::overlap_save (65536-point blocks, 8193 taps, configs[4]), dev::fft (1024 points), dev::fir (127 taps / 5): source -> block -> checksum "
                    "sink against the same two launches bare (four output buffers in rotation); above 1.0 where the ring's byte budget keeps a message in the "
                    "last-level cache between its producer and its reader (include/kpn_dev.hpp, profiles/r06_kpn_ring_bytes.txt).",
            "leg_seconds": time.perf_counter() - t0}


def dropin_calls_leg(R):
    """other_configs.dropin_calls: the reference's own call pattern, unmodified (kissfft.rs:26, samplerate.rs:76, dsputils.rs:30): host buffers in
    and out, synchronous, one message per call.  Microseconds per call on this box (PCIe + launch bound) and the oracle's CPU restatement of
    the same call on one core beside it."""
    import numpy as np
    import oracle as O
    from libredio_amd import dsputils, samplerate
    res = {}

    def per_call(f, reps, warm=20):
        for _ in range(warm):
            f()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        return (time.perf_counter() - t0) / reps * 1e6

    # kiss_fft, N = 1024 (kissfft.rs:19,26)
    K = R.kisslib()
    cfg = K.kiss_fft_alloc(1024, 0, None, None)
    x = O.synth_iq(3, 0, 1024); y = np.empty_like(x)
    xp, yp = x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p)
    gpu = per_call(lambda: K.kiss_fft(cfg, xp, yp), 2000)
    cpu = per_call(lambda: O.fft(x, 1024), 2000)
    res["kiss_fft_1024"] = {"gpu_us_per_call": gpu, "cpu_oracle_us_per_call": cpu, "samples_per_call": 1024, "gpu_msps": 1024 / gpu, "cpu_msps": 1024 / cpu}
    K.kiss_fft_free(cfg)
    # src_process, 4096 frames, ratio 0.02, medium converter, mono (samplerate.rs:61-76)
    st, ost = samplerate.State(1, 1), O.Resampler(1)
    xr = O.synth_f32(5, 0, 4096)
    gpu = per_call(lambda: st.block(xr, 0.02), 300, warm=10)
    cpu = per_call(lambda: ost.block(xr, 0.02), 300, warm=10)
    res["src_process_4096"] = {"gpu_us_per_call": gpu, "cpu_oracle_us_per_call": cpu, "samples_per_call": 4096, "gpu_msps": 4096 / gpu, "cpu_msps": 4096 / cpu}
    st.close()
    # redio_convolve_f32, 1024 samples x 63 taps (C1's message, dsputils.rs:30)
    u, v = O.synth_f32(1, 0, 1024), O.lpf_corrected(63, 0.1)
    gpu = per_call(lambda: dsputils.convolve(u, v), 1000)
    cpu = per_call(lambda: O.convolve(u, v), 1000)
    res["convolve_1024x63"] = {"gpu_us_per_call": gpu, "cpu_oracle_us_per_call": cpu, "samples_per_call": 1024, "gpu_msps": 1024 / gpu, "cpu_msps": 1024 / cpu}
    res["note"] = ("host clock around back-to-back calls through ctypes (about 1 us of Python per call on both sides); gpu = libkissfft.so / libsamplerate.so / "
                   "libredio.so as the reference's Rust would call them (pageable host buffers in and out, result present on return); cpu = oracle/ restatement of "
                   "the same call, one core.  At these message sizes a call is bound by PCIe + launch latency, not by the kernel: the device-resident plans "
                   "(kpn_graph_c2) are the throughput path")
    return res


def other_configs(R, lib, stream, x, n, moved_json=None):
    """Outside the timed region, rank 0 only, never `value`: the other BASELINE.json configs at SURVEY.md 8d's sizes, each timed with HIP
    events on the launch stream after >= 100 ms of back-to-back warm-up launches (the chip raises its clock over the first ~100 ms of a
    burst; tools/bench_configs.py times its lines the same way).  Every entry: workload, alg_bytes per call (SURVEY.md 8d's per-sample
    figure x the samples of the call), kernel_ms (mean per call), frac of 8 TB/s, which roofline binds, the kernels the call launches.
    x: the resident 2^28-sample cf32 stream of the headline (reused as the input of C4 / C5 / the 65536-point transform)."""
    import torch

    def timed(f, reps, warm_ms=100.0, warm_min=3):
        e0, e1 = C.c_void_p(), C.c_void_p()
        R.check(lib.redio_event_create(C.byref(e0)))
        R.check(lib.redio_event_create(C.byref(e1)))
        for _ in range(warm_min):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < warm_ms:
            for _ in range(4):
                f()
            torch.cuda.synchronize()
        R.check(lib.redio_event_record(e0, stream))
        for _ in range(reps):
            f()
        R.check(lib.redio_event_record(e1, stream))
        torch.cuda.synchronize()
        ms = C.c_float()
        R.check(lib.redio_event_elapsed_ms(e0, e1, C.byref(ms)))
        lib.redio_event_destroy(e0)
        lib.redio_event_destroy(e1)
        return ms.value / reps

    def entry(workload, alg_bytes, ms, kernel, units, unit_name, binding="hbm", **more):
        e = {"workload": workload, "alg_bytes": alg_bytes, "kernel_ms": ms, "achieved": alg_bytes / (ms * 1e-3) / 1e9, "unit": "GB/s",
             "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "binding": binding, "kernel": kernel,
             "value": units / ms / 1e3, "value_unit": unit_name}
        e.update(more)
        return e

    res = {}
    t_leg = time.perf_counter()
    moved = {}
    if moved_json and os.path.exists(moved_json):
        moved = json.load(open(moved_json))

    def moved_over_alg(key, kernel_prefix, alg_bytes):
        """PMC-measured HBM bytes per call / algorithmic bytes, from profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of
        tools/pmc_moved.sh, FETCH_SIZE doubled); refused when the file was recorded for other kernels.  Not measured by this run."""
        m = moved.get(key)
        if not m:
            return {"moved_bytes_over_alg": None, "moved_source": "no PMC record for this config in %s" % (os.path.relpath(moved_json, ROOT) if moved_json else None)}
        if not any(k.startswith(kernel_prefix) for k in m.get("kernels", [])):
            return {"moved_bytes_over_alg": None, "moved_source": "%s REFUSED: recorded for kernels %r" % (os.path.relpath(moved_json, ROOT), m.get("kernels"))}
        return {"moved_bytes_over_alg": m["bytes_per_call"] / alg_bytes, "moved_bytes_per_call": m["bytes_per_call"],
                "moved_source": "%s: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of tools/pmc_moved.sh (FETCH_SIZE doubled, the gfx950 "
                                "correction), summed over kernels %s of one call; not measured by this run" % (os.path.relpath(moved_json, ROOT), m["kernels"])}

    # --- configs[3] on one GPU: 64-channel polyphase filterbank, 16 taps per branch, the whole 2^28-sample slice (16 B per sample) ---
    h = R.dsputils.lpf_corrected(1024, 0.45 / 64)
    pfb = R.Channelizer(h)
    rows = pfb.nrows(n)
    o4 = torch.empty((rows, 64), dtype=torch.complex64, device="cuda")
    ms = timed(lambda: pfb(x, out=o4), 50)
    res["c4_channelizer_cf32"] = entry("BASELINE.json configs[3] on one GPU: 64-channel critically sampled polyphase filterbank, prototype 64 x 16 taps, "
                                       "2^%d cf32 samples per call, natural [row][channel] output" % (n.bit_length() - 1),
                                       16.0 * n, ms, "pfb64_kernel<16>", n, "MSamples/s")
    g = torch.Generator(device="cuda"); g.manual_seed(0x5EED0004)
    raw = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
    ms = timed(lambda: pfb.from_bytes(raw, out=o4), 50)
    res["c4_channelizer_u8"] = entry("the same filterbank fed with the receiver's u8 I/Q bytes (rtlsdr::data_to_samples folded into the window loads): "
                                     "2 B in + 8 B out per sample", 10.0 * n, ms, "pfb64_kernel<16, u8>", n, "MSamples/s",
                                     binding="valu (37+ vector instructions per sample; DESIGN.md 5.6)")
    del raw, o4, pfb

    # --- configs[4] on one GPU: overlap-save, 65536-point blocks, 8193 taps (17.14 B per output sample), >= 2 work-buffer chunks ---
    K5, N5 = 8193, 65536
    ovs = R.OverlapSave(R.dsputils.lpf_corrected(K5, 0.08), N5)
    o5 = torch.empty(ovs.nout(n), dtype=torch.complex64, device="cuda")
    hop = N5 - K5 + 1
    b5 = 8.0 * N5 / hop + 8.0
    ms = timed(lambda: ovs(x, out=o5), 20)
    res["c5_overlap_save_65536"] = entry("BASELINE.json configs[4] on one GPU: overlap-save FFT convolution, 65536-point blocks, 8193 taps (hop 57344), "
                                         "%d blocks = %d output samples per call (64 MiB work-buffer chunks: %d chunk steps)"
                                         % (o5.numel() // hop, o5.numel(), (o5.numel() // hop * N5 * 8 + (64 << 20) - 1) // (64 << 20)),
                                         b5 * o5.numel(), ms, "ovsave64k_step_kernel (gather / middle / last tiles interleaved per chunk step)",
                                         o5.numel(), "MSamples/s (output)",
                                         note="three passes at the L2 boundary move about 3 x the algorithmic bytes: the scheme's ceiling is 25 % at this pool's copy rate",
                                         **moved_over_alg("c5_overlap_save_65536", "ovsave64k", b5 * o5.numel()))
    del o5, ovs

    # --- the 65536-point transform alone (kissfft::fft at configs[4]'s block size), 2^26 points per call, 16 B per point ---
    m = min(n, 1 << 26)
    fft = R.Fft(65536)
    xs = x[:m]
    of = torch.empty(m, dtype=torch.complex64, device="cuda")
    ms = timed(lambda: fft(xs, out=of), 50)
    res["fft_65536"] = entry("kissfft::fft, 65536-point forward transforms, %d blocks (2^%d points) per call" % (m // 65536, m.bit_length() - 1),
                             16.0 * m, ms, "fftbig_first_kernel + fftbig_mid_kernel (two passes, 32 B moved per point)", m, "MSamples/s",
                             **moved_over_alg("fft_65536", "fftbig", 16.0 * m))
    del of, fft

    # --- configs[2]: 256 channels x 2^22 frames, ratio 48000 / 2400000 = 0.02, medium-quality sinc converter (4.08 B per input frame) ---
    nch, frames, ratio = 256, 1 << 22, 0.02
    x3 = torch.empty((nch, frames), dtype=torch.float32, device="cuda")
    for c in range(nch):
        x3[c] = R.synth_f32(100 + c, 0, frames)
    b3 = nch * frames * 4.0 * (1.0 + ratio)
    tab_half, tab_inc = 22438 - 2, 491                      # the medium converter's half table and increment (src_core.h)
    taps_per_out = 2 * int(tab_half / (tab_inc * ratio)) + 1
    nout3 = nch * frames * ratio
    for key, mode, kern, what in (("c3_resample_exact", R.Src.EXACT, "src_window_rb_kernel",
                                   "bit-identical to the oracle's libsamplerate-0.1.8 arithmetic: f32 -> f64 convert, separately rounded v_mul_f64 + v_add_f64 per tap"),
                                  ("c3_resample_fast", R.Src.FAST, "src_window_fastp2_kernel",
                                   "opt-in REDIO_SRC_FAST mode: the same filter as an f32 polyphase bank (v_pk_fma_f32), tolerance-tested, NOT bit-identical")):
        plan = R.Src(nch, 1, mode=mode)
        plan.process(x3, ratio)                             # first call: history of zeros, same work
        ms = timed(lambda: plan.process(x3, ratio), 10 if mode == R.Src.EXACT else 20)
        flops = 2.0 * taps_per_out * nout3
        if mode == R.Src.EXACT:
            # 78.6 TFLOP/s = the f64 vector peak (FMA counted as 2); unfused mul + add can reach half of it; the modelled issue bound is the
            # measured back-to-back cost of this kernel's three instructions per tap (tools/bench_configs.py c3: 6.2 / 2 + 5.0 + 4.6 cycles)
            t_issue = nout3 * taps_per_out / 64 * 12.7 / 1024 / 2.4e9
            more = {"valu": {"tflops": flops / (ms * 1e-3) / 1e12, "peak_tflops_f64": 78.6, "frac_of_f64_peak": flops / (ms * 1e-3) / 1e12 / 78.6,
                             "frac_of_modelled_issue_bound": t_issue / (ms * 1e-3), "modelled_issue_bound_ms": t_issue * 1e3}}
            bind = "valu_f64"
        else:
            more = {"valu": {"tflops": flops / (ms * 1e-3) / 1e12, "peak_tflops_f32": 157.3, "frac_of_f32_peak": flops / (ms * 1e-3) / 1e12 / 157.3}}
            bind = "valu_f32"
        res[key] = entry("BASELINE.json configs[2]: samplerate polyphase resample 2.4 MS/s -> 48 kS/s (ratio 0.02), %d channels x 2^%d frames per call, "
                         "%d taps per output; %s" % (nch, frames.bit_length() - 1, taps_per_out, what),
                         b3, ms, kern, nch * frames, "MSamples/s (input)", binding=bind, **more)
        del plan
    del x3
    torch.cuda.empty_cache()
    res["leg_seconds"] = time.perf_counter() - t_leg
    res["note"] = ("each entry: HIP events on the launch stream around `reps` back-to-back calls after >= 100 ms of warm-up calls; inputs resident in HBM; "
                   "outside the timed region, beside value, never value; every config has a full-shape bit-exactness test under tests/ (-m gpu)")
    return res


def main():
    a = parse()
    import torch
    import libredio_amd as R

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be started with torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a HIP device (libredio has no CPU path)")
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and local_rank >= ndev:
        sys.exit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} HIP devices are visible")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(a.backend)

    # who takes part: every rank's device as torch.distributed sees it.  Under the nccl backend two ranks on one device would
    # produce a "scaling curve" from ranks that shared a GPU: refuse (every rank exits non-zero, nothing is timed).
    ranks_seen = None
    if dist is not None:
        p = torch.cuda.get_device_properties(dev_index)
        me = {"rank": rank, "local_rank": local_rank, "device": dev_index, "name": p.name,
              "pci": "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), getattr(p, "pci_bus_id", -1) & 0xFF, getattr(p, "pci_device_id", 0)),
              "uuid": str(getattr(p, "uuid", ""))}
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, me)
        ranks_seen = {"world_size": dist.get_world_size(), "backend": a.backend, "devices": seen,
                      "distinct_devices": len({(d["pci"], d["uuid"]) for d in seen})}
        if a.backend == "nccl" and ranks_seen["distinct_devices"] != a.gpus:
            if rank == 0:
                print(f"bench.py --gpus {a.gpus}: the {world} ranks see only {ranks_seen['distinct_devices']} distinct devices "
                      f"({[d['pci'] for d in seen]}); refusing to time ranks that share a GPU", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(3)

    lib = R.lib()
    n = 1 << a.log2_samples
    taps = R.dsputils.lpf_corrected(NTAPS, FC)
    chain = R.Chain(taps, DECIM, NFFT, fused=not a.exact)
    if a.unfused:
        chain.set_unfused(True)
    nblk = chain.nblocks(n)
    used = nblk * NFFT * DECIM  # input samples that contribute to a spectrum
    # rank r owns the slice that starts at decimated block r*nblk of the global stream (sharding.weak_slice)
    from libredio_amd import sharding
    first, need = sharding.weak_slice(rank, nblk, NTAPS, DECIM, NFFT)
    assert need <= n
    x = R.synth_iq(SEED, first, n)
    out = torch.empty((nblk, NFFT), dtype=torch.complex64, device="cuda")
    stream = R.current_stream()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def launches_ms(count):
        """`count` launches with a HIP event after each, on the launch stream; per-launch ms (synchronises)."""
        ev = []
        for _ in range(count + 1):
            e = C.c_void_p()
            R.check(lib.redio_event_create(C.byref(e)))
            ev.append(e)
        R.check(lib.redio_event_record(ev[0], stream))
        for k in range(count):
            chain(x, out)
            R.check(lib.redio_event_record(ev[k + 1], stream))
        torch.cuda.synchronize()
        res = []
        for k in range(count):
            ms = C.c_float()
            R.check(lib.redio_event_elapsed_ms(ev[k], ev[k + 1], C.byref(ms)))
            res.append(ms.value)
        for e in ev:
            lib.redio_event_destroy(e)
        return res

    # clock pre-conditioning (docstring): the same launch on the same buffers until its time has converged, on every rank
    precondition = None
    if not a.no_precondition:
        WIN, TOL = 20, 0.01
        series, meds = [], []
        t_pc = time.perf_counter()
        while len(series) < a.precondition_max:
            w = launches_ms(min(WIN, a.precondition_max - len(series)))
            series += w
            meds.append(sorted(w)[len(w) // 2])
            if len(meds) >= 3 and max(meds[-3:]) <= (1.0 + TOL) * min(meds[-3:]):
                break
        precondition = {"launches": len(series), "ms": (time.perf_counter() - t_pc) * 1e3,
                        "criterion": f"medians of the last three windows of {WIN} launches within {TOL * 100:.0f} % of each other, at most {a.precondition_max} launches",
                        "converged": len(meds) >= 3 and max(meds[-3:]) <= (1.0 + TOL) * min(meds[-3:]),
                        "first_ms": series[0], "last_ms": series[-1], "window_medians_ms": [round(m, 4) for m in meds]}
        # the cold-burst figure of rounds 1-3 from the same run: launches W .. W+K-1 of the burst, i.e. what `--warmup W --steps K` timed
        # before the pre-conditioning existed (per-launch HIP events; the slowest rank's sum)
        cold_s = sum(series[a.warmup:a.warmup + a.steps]) * 1e-3 if len(series) >= a.warmup + a.steps else None
        if cold_s is not None and dist is not None:
            tc = torch.tensor([cold_s], dtype=torch.float64, device="cuda")
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            cold_s = float(tc.item())
        precondition["cold_seconds_for_steps"] = cold_s

    for _ in range(a.warmup):
        chain(x, out)
    # HIP events around every timed launch, on the launch stream
    evs = []
    for _ in range(a.steps + 1):
        e = C.c_void_p()
        R.check(lib.redio_event_create(C.byref(e)))
        evs.append(e)
    barrier()
    t0 = time.perf_counter()
    R.check(lib.redio_event_record(evs[0], stream))
    for k in range(a.steps):
        chain(x, out)
        R.check(lib.redio_event_record(evs[k + 1], stream))
    barrier()
    dt = time.perf_counter() - t0
    kms = []
    for k in range(a.steps):
        ms = C.c_float()
        R.check(lib.redio_event_elapsed_ms(evs[k], evs[k + 1], C.byref(ms)))
        kms.append(ms.value)
    for e in evs:
        lib.redio_event_destroy(e)

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # outside the timed region: the same launch repeated --steady more times, per-launch HIP events again
    steady = None
    if a.steady > 0:
        sev = []
        for _ in range(a.steady + 1):
            e = C.c_void_p()
            R.check(lib.redio_event_create(C.byref(e)))
            sev.append(e)
        R.check(lib.redio_event_record(sev[0], stream))
        for k in range(a.steady):
            chain(x, out)
            R.check(lib.redio_event_record(sev[k + 1], stream))
        torch.cuda.synchronize()
        sms = []
        for k in range(a.steady):
            ms = C.c_float()
            R.check(lib.redio_event_elapsed_ms(sev[k], sev[k + 1], C.byref(ms)))
            sms.append(ms.value)
        for e in sev:
            lib.redio_event_destroy(e)
        tail = sorted(sms[len(sms) // 3:])          # the last two thirds: past the clock ramp
        steady = {"launches": a.steady, "after_launches": a.warmup + a.steps + (precondition["launches"] if precondition else 0),
                  "kernel_ms_mean": sum(tail) / len(tail), "kernel_ms_median": tail[len(tail) // 2], "kernel_ms_min": tail[0],
                  "series_every_10th_ms": [round(v, 4) for v in sms[::10]]}

    # outside the timed region: the same launch with the reference's rounding (multiply and add rounded
    # separately, dsputils.rs:31) so that both arithmetic modes are on record from one run
    ref_ms = None
    if not a.exact and not a.unfused:
        ref_chain = R.Chain(taps, DECIM, NFFT, fused=False)
        e0, e1 = C.c_void_p(), C.c_void_p()
        R.check(lib.redio_event_create(C.byref(e0)))
        R.check(lib.redio_event_create(C.byref(e1)))
        for _ in range(20):
            ref_chain(x, out)
        R.check(lib.redio_event_record(e0, stream))
        for _ in range(100):
            ref_chain(x, out)
        R.check(lib.redio_event_record(e1, stream))
        torch.cuda.synchronize()
        ms = C.c_float()
        R.check(lib.redio_event_elapsed_ms(e0, e1, C.byref(ms)))
        ref_ms = ms.value / 100
        lib.redio_event_destroy(e0)
        lib.redio_event_destroy(e1)

    # outside the timed region: the same chain fed with the receiver's u8 I/Q bytes (redio_chain_enqueue_u8: data_to_samples folded
    # into the kernel's loader, 3.6 bytes per sample through HBM) -- a different input format, reported beside value, never as value
    u8_ms = None
    if not a.exact and not a.unfused and not a.no_u8_leg and n % 2 == 0:
        g = torch.Generator(device="cuda"); g.manual_seed(0x5EED0002 + rank)
        raw = torch.randint(0, 256, (2 * n,), dtype=torch.uint8, device="cuda", generator=g)
        e0, e1 = C.c_void_p(), C.c_void_p()
        R.check(lib.redio_event_create(C.byref(e0)))
        R.check(lib.redio_event_create(C.byref(e1)))
        for _ in range(20):
            chain.from_bytes(raw, out)
        R.check(lib.redio_event_record(e0, stream))
        for _ in range(100):
            chain.from_bytes(raw, out)
        R.check(lib.redio_event_record(e1, stream))
        torch.cuda.synchronize()
        ms = C.c_float()
        R.check(lib.redio_event_elapsed_ms(e0, e1, C.byref(ms)))
        u8_ms = ms.value / 100
        lib.redio_event_destroy(e0)
        lib.redio_event_destroy(e1)
        del raw

    # outside the timed region ("C2 also reported stage-by-stage"): the chain's two stages as kernels of their own --
    # the decimating FIR alone (8 + 8/5 B per input sample) and the 1024-point transform alone on the decimated blocks (8 + 8 B per
    # sample it transforms) -- 20 warm-up + 50 timed launches each
    stages = None
    if not a.exact and not a.unfused:
        def timed(f, warm=20, reps=50):
            e0, e1 = C.c_void_p(), C.c_void_p()
            R.check(lib.redio_event_create(C.byref(e0)))
            R.check(lib.redio_event_create(C.byref(e1)))
            for _ in range(warm):
                f()
            R.check(lib.redio_event_record(e0, stream))
            for _ in range(reps):
                f()
            R.check(lib.redio_event_record(e1, stream))
            torch.cuda.synchronize()
            ms = C.c_float()
            R.check(lib.redio_event_elapsed_ms(e0, e1, C.byref(ms)))
            lib.redio_event_destroy(e0)
            lib.redio_event_destroy(e1)
            return ms.value / reps
        fir = R.Fir(taps, DECIM, fused=True)
        y = torch.empty(fir.nout(n), dtype=torch.complex64, device="cuda")
        fir_ms = timed(lambda: fir(x, out=y))
        fft = R.Fft(NFFT)
        yb = y[: nblk * NFFT]
        fft_ms = timed(lambda: fft(yb, out=out.view(-1)))
        stages = {"fir_only_ms": fir_ms, "fir_only_frac": ALG_BYTES_PER_SAMPLE * n / (fir_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "fir_only_bytes_per_input_sample": ALG_BYTES_PER_SAMPLE,
                  "fft_only_ms": fft_ms, "fft_only_frac": 16.0 * nblk * NFFT / (fft_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                  "fft_only_bytes_per_transformed_sample": 16.0,
                  "note": "the two stages as kernels of their own on the same data (FIR 127 taps / 5 over the 2^%d input samples; forward 1024-point "
                          "transform of its %d decimated blocks); as separate kernels the chain moves 12.8 B per input sample, fused 9.6" % (a.log2_samples, nblk)}
        del y

    # outside the timed region, rank 0 only: the other BASELINE configs (docstring of other_configs)
    others = None
    if rank == 0 and not a.no_other_configs and not a.exact and not a.unfused and a.log2_samples >= 26:
        del out
        torch.cuda.empty_cache()
        others = other_configs(R, lib, stream, x, n, a.moved_json)
        if not a.no_cpu_baseline:
            cb = cpu_baseline_configs()
            for key, ck in (("c3_resample_exact", "c3"), ("c3_resample_fast", "c3"), ("c4_channelizer_cf32", "c4"), ("c4_channelizer_u8", "c4"),
                            ("c5_overlap_save_65536", "c5"), ("fft_65536", "fft_65536")):
                others[key]["cpu_baseline"] = cb[ck]

    cpu_rec = None
    if rank == 0:
        # rank 0 only, also at N > 1 (the other ranks have nothing left to do; they wait in destroy_process_group)
        cpu_rec = None if a.no_cpu_baseline else cpu_baseline(a.cpu_log2_samples)
        if others is not None and not a.no_graph_leg:
            del x
            torch.cuda.empty_cache()
            others["kpn_graph_c2"] = kpn_graph_leg(cpu_rec["kpn_pipeline"]["value"] if cpu_rec else None)
            others["dropin_calls"] = dropin_calls_leg(R)
    if rank == 0:
        kavg = sum(kms) / len(kms) / 1e3  # s per launch (launch-to-launch on the stream)
        alg_bytes = (12.8 if a.unfused else ALG_BYTES_PER_SAMPLE) * used
        ach = alg_bytes / kavg / 1e9
        traffic, traffic_source = None, None
        kernel_name = chain.kernel_name  # redio_chain_kernel_name: what this plan launches (None: two kernels)
        if a.traffic_json and os.path.exists(a.traffic_json) and not a.unfused and a.log2_samples == 28:
            tj = json.load(open(a.traffic_json))
            if (tj.get("kernel") or "").replace(" ", "") != (kernel_name or ""):
                # counters recorded for another kernel (a stale file after a kernel change) are not this launch's traffic
                traffic_source = (f"{os.path.relpath(a.traffic_json, ROOT)} REFUSED: recorded for kernel {tj.get('kernel')!r}, "
                                  f"this plan launches {kernel_name!r}")
            else:
                traffic = tj.get("traffic")
                traffic_source = (os.path.relpath(a.traffic_json, ROOT) + f": rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of a "
                                  f"separate run of this command (FETCH_SIZE doubled, the gfx950 correction), recorded for {tj.get('kernel')} = the kernel "
                                  f"this plan launches (redio_chain_kernel_name); not measured by this run")
        rec = {
            "metric": "MSamples/s through FIR+FFT+resample chain",
            "value": world * used * a.steps / dt / 1e6,
            "unit": "MSamples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3,
            # comparable with rounds 1-3 (no pre-conditioning then): launches W .. W+K-1 of the cold burst of this same run
            "value_cold": (world * used * a.steps / precondition["cold_seconds_for_steps"] / 1e6
                           if precondition and precondition.get("cold_seconds_for_steps") else None),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: single f32 IQ stream, 1024-pt kissfft + 127-tap FIR decimate-by-5, 1 MI355X "
                                   "(input samples/s; the 5:1 resample step is the polyphase decimation of the FIR: only every "
                                   "fifth FIR output is computed, then consecutive 1024-sample blocks are transformed)",
                       "samples_per_gpu": n, "ntaps": NTAPS, "decim": DECIM, "nfft": NFFT,
                       "kernel": "two kernels (fir_tiled + fft1k_wave)" if a.unfused else f"{kernel_name} (fused, wave per block run, halo carried in LDS)",
                       "fir_rounding": "mul+add (reference)" if a.exact else "fmaf, reference order",
                       "parallelism": f"time-sliced replicas x{world}, no collective"},
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "alg_bytes_per_launch": alg_bytes, "kernel_ms": kavg * 1e3,
                         # "report median and min" of the timed repetitions (the same per-launch HIP events)
                         "kernel_ms_median": sorted(kms)[len(kms) // 2], "kernel_ms_min": min(kms),
                         "frac_of_measured_copy_6290": ach / 6290.0,
                         # "report min(HBM, VALU) honestly": 4K/D = 101.6 (FIR, complex samples x real taps) + 5 log2(1024) / 5 = 10
                         # (transform on the decimated stream) flops per input sample against the 157.3 TFLOP/s f32 vector peak
                         "valu": {"flops_per_sample": 4.0 * NTAPS / DECIM + 5.0 * 10 / DECIM, "tflops": (4.0 * NTAPS / DECIM + 10.0) * used / kavg / 1e12,
                                  "peak_tflops": 157.3, "frac_of_157": (4.0 * NTAPS / DECIM + 10.0) * used / kavg / 1e12 / 157.3},
                         "binding": "hbm" if ach / HBM_PEAK_GBS >= (4.0 * NTAPS / DECIM + 10.0) * used / kavg / 1e12 / 157.3 else "valu"},
            "launch_ms_series": [round(v, 4) for v in (kms if len(kms) <= 128 else kms[:32] + kms[32::max(1, len(kms) // 96)])],
            "launch_ms_max_over_min": max(kms) / min(kms),
            "precondition": precondition,
        }
        if steady is not None:
            sk = steady["kernel_ms_mean"] * 1e-3
            steady.update({"achieved": alg_bytes / sk / 1e9, "unit": "GB/s", "frac": alg_bytes / sk / 1e9 / HBM_PEAK_GBS,
                           "value_per_gpu": used / sk / 1e6, "value_unit": "MSamples/s",
                           "note": "the same launch repeated after the timed region (mean of the last two thirds): what the kernel "
                                   "sustains once the clock transient of a burst is over; reported beside value, never as value"})
            rec["steady_state"] = steady
        if ref_ms is not None:
            rec["reference_rounding"] = {"kernel_ms": ref_ms, "value_per_gpu": used / ref_ms / 1e3, "unit": "MSamples/s",
                                         "frac": alg_bytes / (ref_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                         "note": "same kernel built with separately rounded multiply and add: bit-identical to the "
                                                 "reference arithmetic; the headline build uses fmaf in the reference's tap order "
                                                 "(within the stated f32 tolerance, tests/test_gpu_parity.py)"}
        if u8_ms is not None:
            rec["from_u8_bytes"] = {"kernel_ms": u8_ms, "value_per_gpu": used / u8_ms / 1e3, "unit": "MSamples/s",
                                    "alg_bytes_per_sample": 3.6, "frac": 3.6 * used / (u8_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    # the binding roofline of this leg is the vector unit, not HBM: the same 111.6 flops per sample as the headline
                                    # (the byte -> f32 conversion's divide and subtract are not counted) against the 157.3 TFLOP/s f32 vector peak
                                    "valu": {"flops_per_sample": 4.0 * NTAPS / DECIM + 10.0, "tflops": (4.0 * NTAPS / DECIM + 10.0) * used / (u8_ms * 1e-3) / 1e12,
                                             "peak_tflops": 157.3, "frac_of_157": (4.0 * NTAPS / DECIM + 10.0) * used / (u8_ms * 1e-3) / 1e12 / 157.3},
                                    "binding": "valu",
                                    "note": "the same chain from interleaved u8 I/Q bytes (rtlsdr::data_to_samples folded into the kernel's "
                                            "loader, redio_chain_enqueue_u8): 2 + 1.6 bytes per sample, VALU-bound; a different input "
                                            "format from BASELINE.json configs[1] (f32 IQ), so beside value, never as value"}
        if stages is not None:
            rec["stages"] = stages
        if others is not None:
            rec["other_configs"] = others
        if ranks_seen is not None:
            rec["ranks_seen"] = ranks_seen
        rec["cpu_baseline"] = cpu_rec
        print(json.dumps(rec))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
