"""The C-ABI libraries load without a GPU and export every symbol include/*.h declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b((?:redio_|kiss_fft|src_)[a-z0-9_]*)\s*\(", src)
    return sorted(set(names))


def test_redio_h_symbols(redio):
    L = C.CDLL(redio.LIBREDIO)
    names = declared("redio.h")
    assert len(names) >= 35
    for n in names:
        assert hasattr(L, n), f"libredio.so does not export {n}"


def test_kiss_fft_h_symbols(redio):
    K = redio.kisslib()
    names = [n for n in declared("kiss_fft.h") if n.startswith("kiss_fft")]
    assert {"kiss_fft_alloc", "kiss_fft", "kiss_fft_cleanup"} <= set(names)  # kissfft.rs:13-15
    for n in names:
        assert hasattr(K, n), f"libkissfft.so does not export {n}"


def test_samplerate_h_symbols(redio):
    S = redio.samplerate_lib()
    names = [n for n in declared("samplerate.h") if n.startswith("src_")]
    bound = {"src_new", "src_delete", "src_process", "src_get_name", "src_get_description", "src_get_version",
             "src_set_ratio", "src_is_valid_ratio", "src_strerror"}  # samplerate.rs:34-42
    assert bound <= set(names)
    for n in names:
        assert hasattr(S, n), f"libsamplerate.so does not export {n}"
    assert S.src_is_valid_ratio(0.02) == 1 and S.src_strerror(6)


def test_src_data_layout_is_64_bytes():
    import ctypes as C
    from libredio_amd.samplerate import SRC_DATA
    assert C.sizeof(SRC_DATA) == 64  # samplerate.rs:15-24 on LP64


def test_no_cpu_fallback_without_device(redio):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    n = C.c_int(-1)
    assert redio.lib().redio_device_count(C.byref(n)) == -4 and n.value == 0
    with pytest.raises(redio.RedioError):
        redio.dsputils.convolve([1.0, 2.0, 3.0], [1.0])
    assert redio.kisslib().kiss_fft_alloc(64, 0, None, None) is None
    err = C.c_int(0)
    assert redio.samplerate_lib().src_new(1, 1, C.byref(err)) is None and err.value != 0


def test_errors_have_text(redio):
    for code in (0, -1, -2, -3, -4, -5, -1001):
        assert redio.lib().redio_strerror(code)


def test_host_tap_generators_match_oracle(redio, oracle):
    import numpy as np
    for m in (4, 63, 64, 127):
        for a, b in ((redio.dsputils.window(m), oracle.window(m)),
                     (redio.dsputils.sinc(m, 0.1), oracle.sinc(m, 0.1)),
                     (redio.dsputils.lpf(m, 0.1), oracle.lpf(m, 0.1)),
                     (redio.dsputils.hpf(m, 0.2), oracle.hpf(m, 0.2)),
                     (redio.dsputils.bsf(m, 0.1, 0.2), oracle.bsf(m, 0.1, 0.2)),
                     (redio.dsputils.bpf(m, 0.1, 0.2), oracle.bpf(m, 0.1, 0.2)),
                     (redio.dsputils.lpf_corrected(m, 0.08), oracle.lpf_corrected(m, 0.08))):
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))  # NaN-aware: compare bits
    with pytest.raises(redio.RedioError):
        redio.dsputils.sinc(8, 0.5)
    with pytest.raises(redio.RedioError):
        redio.dsputils.hpf(1, 0.1)


def test_missing_library_fails_loudly(tmp_path):
    """The product has no fallback: with libredio.so absent the loader raises instead of computing on the CPU
    (run in a fresh interpreter so that the library this suite already loaded does not mask the check)."""
    import subprocess
    import sys
    code = (
        "import os, sys\n"
        f"sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})\n"
        "import libredio_amd as R\n"
        f"R.LIBREDIO = {str(tmp_path / 'nope' / 'libredio.so')!r}\n"
        "try:\n"
        "    R.dsputils.convolve([1.0, 2.0, 3.0], [1.0])\n"
        "except ImportError as e:\n"
        "    print('LOUD', e); sys.exit(0)\n"
        "sys.exit(7)\n")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "LOUD" in out.stdout, (out.returncode, out.stdout, out.stderr)


def test_oracle_is_test_infrastructure_only():
    """The CPU restatement under oracle/ is the checker, never the product: nothing in the package, the headers or the tools may
    import, load or link it -- only tests/, __graft_entry__.smoke() / build() and bench.py's cpu_baseline leg do."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(import oracle|from oracle)\b|libredio_oracle|oracle/_build|orc_[a-z_]+\(", re.M)
    offenders = []
    for sub in ("libredio_amd", "include", "tools"):
        for dirpath, dirnames, files in os.walk(os.path.join(root, sub)):
            dirnames[:] = [d for d in dirnames if not d.startswith("_build") and d != "__pycache__" and d != "exp"]
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".sh")) or f == "Makefile":
                    text = open(os.path.join(dirpath, f), errors="replace").read()
                    if pat.search(text):
                        offenders.append(os.path.relpath(os.path.join(dirpath, f), root))
    assert not offenders, offenders
    # bench.py: only the CPU-baseline legs (the headline's, the per-config ones, the CPU side of the per-call drop-in table) import the
    # oracle -- never main(), other_configs() or anything inside the timed region
    import ast
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    users = set()
    for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if (isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names)) or \
               (isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle"):
                users.add(fn.name)
    assert users and users <= {"cpu_baseline", "cpu_baseline_configs", "dropin_calls_leg"}, users
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    assert not any("oracle" in ast.dump(n) for n in top)


def test_public_headers_are_plain_c_and_a_c_program_links(redio, tmp_path):
    """The boundary is a C ABI (plain pointers and sizes): the three public headers compile as C99 and as C++17 under -pedantic, and a C
    program that includes all of them links against the three libraries and runs the calls that need no GPU."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    for h in ("redio.h", "kiss_fft.h", "samplerate.h"):
        for cc, std, lang in (("gcc", "-std=c99", "c"), ("g++", "-std=c++17", "c++")):
            r = subprocess.run([cc, std, "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-x", lang, os.path.join(inc, h)],
                               capture_output=True, text=True, timeout=120)
            assert r.returncode == 0, (h, cc, r.stderr)
    src = tmp_path / "link.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "redio.h"
#include "kiss_fft.h"
#include "samplerate.h"
int main(void)
{
    float w[5], out[3];
    size_t n = 0;
    const float u[4] = {1.f, 2.f, 3.f, 4.f}, v[2] = {0.5f, 0.25f};
    if (!redio_version() || !redio_strerror(-1)) return 1;
    if (redio_window(4, w) != 0) return 2;                       /* host-side generators need no device */
    if (redio_fir_create(NULL, v, 2, 1, 0) >= 0) return 3;       /* argument errors come back as codes */
    if (src_is_valid_ratio(0.02) != 1 || src_is_valid_ratio(1e6) != 0) return 4;
    if (!src_strerror(6) || !src_get_name(1) || src_get_name(9)) return 5;
    if (kiss_fft_next_fast_size(17) != 18) return 6;
    if (sizeof(SRC_DATA) != 64) return 7;                        /* samplerate.rs:15-24 on LP64 */
    (void)u; (void)out; (void)n;
    puts("c link ok");
    return 0;
}
''')
    exe = tmp_path / "link"
    bdir = os.path.dirname(redio.LIBREDIO)
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", inc, str(src), "-o", str(exe), "-L", bdir, "-lredio", "-lkissfft", "-lsamplerate",
                        f"-Wl,-rpath,{bdir}"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c link ok" in r.stdout, (r.returncode, r.stdout, r.stderr)
