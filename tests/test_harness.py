"""The C++ harness: the reference's benchmark loop (src/main.cpp:841-898) around this library, calling it
through include/hsrans_dropin.hpp the way main.cpp's codec table would."""
import os
import subprocess

import numpy as np
import pytest

from hypersonic_rans_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "hypersonic_rans_amd", "bin", "hsrans_harness")


def _run(args, timeout=600):
    return subprocess.run([HARNESS] + args, capture_output=True, text=True, timeout=timeout)


def test_harness_is_built_and_refuses_to_run_without_gpu_or_file(tmp_path):
    assert os.path.exists(HARNESS), "run __graft_entry__.build() first"
    r = _run([])
    assert r.returncode == 1 and "Usage" in r.stdout
    r = _run([str(tmp_path / "missing.bin")])
    assert r.returncode == 1 and "Failed to read file" in r.stdout


@pytest.mark.gpu
def test_harness_validates_every_codec(tmp_path):
    f = tmp_path / "zipf.bin"
    synth.enwik8_shaped(1 << 20, seed=3).tofile(f)
    r = _run([str(f), "--runs", "1", "--decode-runs", "2", "--test"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # per codec: host buffers, runtime dispatch, host buffers + index (one launch / pipelined), device resident; + the GPU encoder line of the 12 mt_ codecs
    assert r.stdout.count("| valid") == 5 * 36 + 12 and "FAILED" not in r.stdout
    assert "All codecs validated." in r.stdout


@pytest.mark.gpu
def test_harness_headline_codec_on_nonstationary_data(tmp_path):
    f = tmp_path / "ns.bin"
    synth.nonstationary(3_000_000).tofile(f)
    r = _run([str(f), "--runs", "1", "--decode-runs", "4", "--bits", "11", "--test"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("| valid") == 5 * 6 + 2
