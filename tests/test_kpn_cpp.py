"""The C++ kpn twin (include/kpn.hpp, include/wavio.hpp): plumbing on the CPU, and the reference-style
graphs around the three hot blocks on the MI355X, checked against the oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "kpn_tests")


@pytest.fixture(scope="module")
def exe(redio):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "-s"])
    return EXE


@pytest.fixture(params=["shared", "per_block", "ring1"])
def devenv(request):
    """The device graphs run three ways: compute blocks on the shared graph stream (the default), a stream per block (order made by events on
    demand), and rings of ONE buffer per block (every message waits for its predecessor's handle: the tightest credit)."""
    env = dict(os.environ)
    if request.param == "per_block":
        env["KPN_DEV_STREAMS"] = "per_block"
    elif request.param == "ring1":
        env["KPN_DEV_RING"] = "1"
    return env


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def write_wav_f32(path, x, rate):
    x = np.ascontiguousarray(x, dtype=np.float32)
    data = x.tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 3, 1, rate, rate * 4, 4, 32))
        f.write(b"data" + struct.pack("<I", len(data)) + data)


def read_wav_f32(path):
    b = open(path, "rb").read()
    i = b.index(b"data")
    n = struct.unpack("<I", b[i + 4:i + 8])[0]
    return np.frombuffer(b[i + 8:i + 8 + n], dtype=np.float32)


def test_plumbing_blocks_cpu(exe):
    out = subprocess.run([exe, "plumbing"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "plumbing ok" in out.stdout, out.stderr


def test_hot_blocks_fail_loudly_without_gpu(exe, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    write_wav_f32(tmp_path / "in.wav", np.zeros(2048, np.float32), 48000)
    out = subprocess.run([exe, "c1", str(tmp_path / "in.wav"), str(tmp_path / "out.wav")], capture_output=True, text=True, timeout=900)
    # convolve has no CPU path: the block dies with the library's error, nothing is computed on the host
    assert not os.path.exists(tmp_path / "out.wav") or len(read_wav_f32(tmp_path / "out.wav")) == 0


@pytest.mark.gpu
def test_config1_wav_fir_wav(exe, gpu, oracle, tmp_path):
    """BASELINE.json configs[0]: wavio WAV in -> dsputils 63-tap FIR lowpass -> WAV out on a kpn graph.
    Per-block valid mode: every 1024-sample block yields 1024-62 outputs.  2^20 samples: SURVEY.md 8d's size for C1."""
    n = 1 << 20
    x = oracle.synth_f32(0x5EED0001, 0, n + 300)  # 300 trailing samples do not fill a block: dropped by shaper
    write_wav_f32(tmp_path / "in.wav", x, 48000)
    out = subprocess.run([exe, "c1", str(tmp_path / "in.wav"), str(tmp_path / "out.wav")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    y = read_wav_f32(tmp_path / "out.wav")
    taps = oracle.lpf_corrected(63, 0.1)
    nblk = (n + 300) // 1024
    want = np.concatenate([oracle.convolve(x[b * 1024:(b + 1) * 1024], taps) for b in range(nblk)])
    assert len(y) == nblk * (1024 - 62)
    assert np.array_equal(bits(y), bits(want))


@pytest.mark.gpu
@pytest.mark.parametrize("n,inv", [(1024, 0), (64, 1)])
def test_kissfft_block_in_cpp_graph(exe, gpu, oracle, tmp_path, n, inv):
    x = oracle.synth_iq(3, 0, n * 6 + 5)
    x.tofile(tmp_path / "in.bin")
    out = subprocess.run([exe, "fft", str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(n), str(inv)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    y = np.fromfile(tmp_path / "out.bin", dtype=np.complex64)
    assert np.array_equal(bits(y), bits(oracle.fft(x[: n * 6], n, bool(inv))))


@pytest.mark.gpu
def test_resample_block_in_cpp_graph(exe, gpu, oracle, tmp_path):
    x = oracle.synth_f32(5, 0, 30000)
    x.tofile(tmp_path / "in.bin")
    out = subprocess.run([exe, "resample", str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "0.5", "7000"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    y = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    ref = oracle.Resampler(1)
    want = np.concatenate([ref.block(x[i:i + 7000], 0.5) for i in range(0, 28000, 7000)])
    assert np.array_equal(bits(y), bits(want))


@pytest.mark.gpu
def test_device_resident_graph_fork_chain_and_fir(exe, gpu, oracle, tmp_path, devenv):
    """include/kpn_dev.hpp: to_device -> fork (shares the allocation) -> {fused chain, FIR} -> to_host."""
    msg = 4 * 5120 + 126
    x = oracle.synth_iq(0x5EED0002, 0, 3 * msg)
    x.tofile(tmp_path / "in.bin")
    out = subprocess.run([exe, "devchain", str(tmp_path / "in.bin"), str(tmp_path / "spec.bin"), str(tmp_path / "fir.bin"), str(msg)],
                         capture_output=True, text=True, timeout=900, env=devenv)
    assert out.returncode == 0, out.stderr
    taps = oracle.lpf_corrected(127, 0.08)
    spec = np.fromfile(tmp_path / "spec.bin", dtype=np.complex64)
    fir = np.fromfile(tmp_path / "fir.bin", dtype=np.complex64)
    want_spec = np.concatenate([oracle.chain_fir_fft(x[i * msg:(i + 1) * msg], taps, 5, 1024, True).reshape(-1) for i in range(3)])
    want_fir = np.concatenate([oracle.fir(x[i * msg:(i + 1) * msg], taps, 5, True) for i in range(3)])
    assert np.array_equal(bits(spec), bits(want_spec)) and np.array_equal(bits(fir), bits(want_fir))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_device_stream_blocks_carry_history_across_messages(exe, gpu, oracle, tmp_path, seed, devenv):
    """include/kpn_dev.hpp stream blocks (redio_*_stream_*): a stream cut into messages of 1 ... 70000 samples gives, message
    seams or not, exactly the outputs of one stateless call on the whole stream (SURVEY.md 8d C2 "history carried")."""
    n = 300000 + 7
    x = oracle.synth_iq(0x5EED0002, 0, n)
    x.tofile(tmp_path / "in.bin")
    out = subprocess.run([exe, "devstream", str(tmp_path / "in.bin"), str(tmp_path / "spec.bin"), str(tmp_path / "fir.bin"),
                          str(tmp_path / "ovs.bin"), str(seed)], capture_output=True, text=True, timeout=900, env=devenv)
    assert out.returncode == 0, out.stderr
    taps = oracle.lpf_corrected(127, 0.08)
    spec = np.fromfile(tmp_path / "spec.bin", dtype=np.complex64)
    fir = np.fromfile(tmp_path / "fir.bin", dtype=np.complex64)
    ovs = np.fromfile(tmp_path / "ovs.bin", dtype=np.complex64)
    assert np.array_equal(bits(spec), bits(oracle.chain_fir_fft(x, taps, 5, 1024, True).reshape(-1)))
    assert np.array_equal(bits(fir), bits(oracle.fir(x, taps, 5, False)))
    assert np.array_equal(bits(ovs), bits(oracle.overlap_save(x, taps, 4096)))


@pytest.mark.gpu
def test_byte_messages_through_the_one_kernel_chain_block(exe, gpu, oracle, tmp_path, devenv):
    """dev::bytes_fir_fft_chain in a C++ kpn graph: the receiver's Vec<u8> messages (rtlsdr.rs:127-152) -> device -> data_to_samples +
    FIR + FFT in one kernel per message -> host.  Per message the spectra of the oracle's data_to_samples (rtlsdr.rs:159-162) and
    chain on that message (stateless blocks: the reference's own message semantics)."""
    msg = 2 * (5 * 1024 * 6 + 122 + 777)           # six blocks and a ragged tail per message
    raw = np.random.default_rng(3).integers(0, 256, 3 * msg + 2 * (5 * 1024 + 122), dtype=np.uint8)
    raw.tofile(tmp_path / "raw.bin")
    out = subprocess.run([exe, "devbytes", str(tmp_path / "raw.bin"), str(tmp_path / "spec.bin"), str(msg)], capture_output=True, text=True, timeout=900, env=devenv)
    assert out.returncode == 0, out.stderr
    taps = oracle.lpf_corrected(127, 0.08)
    want = np.concatenate([oracle.chain_fir_fft(oracle.data_to_samples(raw[o:o + msg]), taps, 5, 1024, True).reshape(-1)
                           for o in range(0, len(raw), msg)])
    got = np.fromfile(tmp_path / "spec.bin", dtype=np.complex64)
    assert got.shape == want.shape and np.array_equal(bits(got), bits(want))


@pytest.mark.gpu
def test_sharded_channelizer_from_one_cpp_process(exe, gpu, oracle, tmp_path):
    """BASELINE.json configs[3] with no Python in the loop: one C++ process, one channelizer thread per visible GPU, the
    regrouping through redio_comm_init_all / redio_pfb_exchange_all (RCCL) -- the reference's thread-per-block host model
    (src/ratpak.rs:60-185).  Every device's [all rows][its channels] equals the oracle's channelizer of the whole stream."""
    M, P, rows = 64, 16, 3001
    x = oracle.synth_iq(0x5EED0004, 0, M * rows)
    x.tofile(tmp_path / "in.bin")
    out = subprocess.run([exe, "devc4", str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "0"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    ndev = int(out.stdout.split("devices")[1].split()[0])
    want = oracle.pfb_channelizer(x, oracle.lpf_corrected(M * P, 0.45 / M), M, P, True)
    cpg, nout = M // ndev, rows - P + 1
    got = np.fromfile(tmp_path / "out.bin", dtype=np.complex64).reshape(ndev, nout, cpg)
    for g in range(ndev):
        assert np.array_equal(bits(got[g]), bits(np.ascontiguousarray(want[:, g * cpg:(g + 1) * cpg]))), g


def _ring_graph(exe, msg, nmsg, depth, warm, policy, host_sync):
    out = subprocess.run([exe, "devring", str(msg), str(nmsg), str(depth), str(warm), str(policy), str(host_sync)], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    f = out.stdout.split()
    return dict(zip(f[0::2], (int(v) for v in f[1::2])))


@pytest.mark.gpu
@pytest.mark.parametrize("depth,policy", [(2, 0), (1, 0), (2, 1), (1, 1)])
def test_bounded_ring_graph_never_allocates_after_warm_up(exe, gpu, oracle, depth, policy):
    """include/kpn_dev.hpp rings + queue order (SURVEY.md 8b "device-resident variant must bound memory (credit/ring)"): synth source -> fused
    chain -> checksum sink, one thread per block, 1000 messages through rings of `depth` buffers, on the shared graph stream (policy 0) and on
    a stream per block with events made on demand (policy 1).  No host synchronisation between the blocks, so a missing dependency would
    show as a wrong checksum; redio_malloc_count() must not move after message 20."""
    msg, nmsg = 2 * 5120 + 126 + 37, 1000
    got = _ring_graph(exe, msg, nmsg, depth, 20, policy, 0)
    assert got["messages"] == nmsg
    assert got["mallocs_at_end"] == got["mallocs_at_warm"], "the graph allocated device memory after warm-up"
    taps = oracle.lpf_corrected(127, 0.08)
    x = oracle.synth_iq(0x5EED0002, 0, msg * nmsg)
    want = 0
    for i in range(nmsg):
        want += int(bits(oracle.chain_fir_fft(x[i * msg:(i + 1) * msg], taps, 5, 1024, True)).astype(np.uint64).sum())
    assert got["checksum"] == want % (1 << 64)


@pytest.mark.gpu
def test_host_sync_mode_and_unpooled_ring_give_the_same_bits(exe, gpu):
    """dev::set_host_sync(true) + ring depth 0 + a stream per block (the round-5 behaviour, kept for debugging and for the before/after line of
    bench_c2) and the queue-ordered rings produce the same checksum over the same 300 messages of 2^16 + 11 samples."""
    sums = {_ring_graph(exe, (1 << 16) + 11, 300, depth, 5, policy, sync)["checksum"] for depth, policy, sync in ((4, 0, 0), (0, 1, 1), (3, 1, 0), (0, 0, 0))}
    assert len(sums) == 1 and 0 not in sums


@pytest.mark.gpu
def test_bench_c2_mode_reports_graph_against_bare_launches(exe, gpu):
    import json
    out = subprocess.run([exe, "bench_c2", "18", "300", "4", "resident", "checksum", "0", "0"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["messages"] == 300 and rec["mallocs_in_timed_region"] == 0 and rec["checksum"] != 0
    assert rec["graph_gsps"] > 0.2 * rec["bare_gsps"]          # a loose floor: the round-5 host path sat at 0.05 - 0.25


@pytest.mark.gpu
@pytest.mark.parametrize("ring_mib", ["128", "0", "1"])
def test_bench_block_mode_checksums_match_the_oracle_under_every_byte_budget(exe, gpu, oracle, ring_mib):
    # source -> dev::fft | dev::fir | dev::channelizer | dev::overlap_save -> checksum sink: the sum over the timed messages is the oracle's, whatever
    # the ring's byte budget lets out at a time (0: one message; 1 MiB: one or two of these; 128 MiB: the full depth)
    import json
    k, nmsg, R = 17, 40, 4
    msg = 1 << k
    env = dict(os.environ, KPN_DEV_RING_MIB=ring_mib)
    out = subprocess.run([exe, "bench_block_list", f"fft:{k}:{nmsg}", f"fir:{k}:{nmsg}", f"channelizer:{k}:{nmsg}", f"ovsave:{k}:{nmsg}"],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr
    recs = {r["block"]: r for r in map(json.loads, out.stdout.strip().splitlines())}
    words = lambda y: int(np.ascontiguousarray(y).view(np.uint32).astype(np.uint64).sum())
    omsg = 65536 + ((msg - 65536) // 57344) * 57344
    want = {"fft": 0, "fir": 0, "channelizer": 0, "ovsave": 0}
    x = oracle.synth_iq(0x5EED0004, 0, R * msg)
    xo = oracle.synth_iq(0x5EED0004, 0, R * omsg)
    t127, proto, t8k = oracle.lpf_corrected(127, 0.08), oracle.lpf_corrected(64 * 16, 0.45 / 64.0), oracle.lpf_corrected(8193, 0.08)
    for r in range(R):
        m = x[r * msg:(r + 1) * msg]
        want["fft"] += words(oracle.fft(m, 1024))
        want["fir"] += words(oracle.fir(m, t127, 5, fused=True))
        want["channelizer"] += words(oracle.pfb_channelizer(m, proto, 64, 16, fused=True))
        want["ovsave"] += words(oracle.overlap_save(xo[r * omsg:(r + 1) * omsg], t8k, 65536))
    for b, rec in recs.items():
        assert rec["messages"] == nmsg and rec["mallocs_in_timed_region"] == 0, rec
        assert rec["checksum"] == (want[b] * (nmsg // R)) % (1 << 64), b
    assert set(recs) == set(want)


@pytest.mark.gpu
def test_device_shaper_rechunks_views(exe, gpu, oracle, tmp_path, devenv):
    x = oracle.synth_f32(3, 0, 10000)
    x.tofile(tmp_path / "in.bin")
    out = subprocess.run([exe, "devshaper", str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), "3000", "1024"], capture_output=True, text=True, timeout=900, env=devenv)
    assert out.returncode == 0, out.stderr
    y = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    assert np.array_equal(bits(y), bits(x[: (10000 // 1024) * 1024]))   # the trailing partial chunk is dropped (kpn.rs:278-282)


@pytest.mark.gpu
def test_device_vector_maps_and_resampler_blocks(exe, gpu, oracle, tmp_path, devenv):
    # f32 stream -> dev::sum_vecs -> dev::mul_vecs -> dev::resample, all in HBM between the PCIe crossings
    msg, ratio = 4000, 0.5
    x = oracle.synth_f32(21, 0, 5 * msg + 123)          # the last message is short: zip truncates to it
    x.tofile(tmp_path / "in.bin")
    out = subprocess.run([exe, "devmix", str(tmp_path / "in.bin"), str(tmp_path / "out.bin"), str(msg), str(ratio)], capture_output=True, text=True, timeout=900, env=devenv)
    assert out.returncode == 0, out.stderr
    i = np.arange(msg)
    c = ((i % 7).astype(np.float32) * np.float32(0.25) - np.float32(0.5)).astype(np.float32)
    c2 = (np.float32(1.0) + (i % 5).astype(np.float32) * np.float32(0.125)).astype(np.float32)
    ref, want = oracle.Resampler(1), []
    for o in range(0, len(x), msg):
        m = oracle.zip_vecs(oracle.zip_vecs(x[o:o + msg], c, add=True), c2, add=False)
        want.append(ref.block(m, ratio))
    y = np.fromfile(tmp_path / "out.bin", dtype=np.float32)
    assert np.array_equal(bits(y), bits(np.concatenate(want)))


@pytest.mark.gpu
def test_device_channelizer_and_overlap_save_blocks(exe, gpu, oracle, tmp_path, devenv):
    msg = 64 * 300
    x = oracle.synth_iq(22, 0, 2 * msg)
    x.tofile(tmp_path / "in.bin")
    out = subprocess.run([exe, "devbank", str(tmp_path / "in.bin"), str(tmp_path / "pfb.bin"), str(tmp_path / "ovs.bin"), str(msg)],
                         capture_output=True, text=True, timeout=900, env=devenv)
    assert out.returncode == 0, out.stderr
    proto, taps = oracle.lpf_corrected(64 * 16, 0.45 / 64), oracle.lpf_corrected(127, 0.08)
    want_pfb = np.concatenate([oracle.pfb_channelizer(x[i * msg:(i + 1) * msg], proto, 64, 16, True).reshape(-1) for i in range(2)])
    want_ovs = np.concatenate([oracle.overlap_save(x[i * msg:(i + 1) * msg], taps, 4096) for i in range(2)])
    assert np.array_equal(bits(np.fromfile(tmp_path / "pfb.bin", dtype=np.complex64)), bits(want_pfb))
    assert np.array_equal(bits(np.fromfile(tmp_path / "ovs.bin", dtype=np.complex64)), bits(want_ovs))
