"""A short, seeded run of the randomised differential tool (tests/fuzz_parity.py): FIR, FFT, chain, overlap-save,
channelizer and resampler against the oracle on random shapes, lengths, alignments and message cuts."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_randomised_differential_run(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_parity.py"), "20", "7"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "failures 0" in out.stdout
