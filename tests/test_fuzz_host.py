"""Sanitizer fuzz of the code that handles untrusted streams and plans on the host (tools/fuzz/fuzz_host.cpp: the planner, the plan
validator and the host decoder under AddressSanitizer + UBSan; the same validator guards every GPU entry).  A short run on every
test pass; longer ones by hand: `make -C tools/fuzz && tools/fuzz/fuzz_host 600 <seed>`.  Round 3's first run found an
allocation sized by a stream header's claimed decoded length (hsrans_index_build_host): fixed, and pinned below."""
import os
import subprocess

import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sanitizer_fuzz_of_planner_validator_and_host_decoder():
    r = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tools", "fuzz")], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        pytest.skip("sanitizer build not available here: " + r.stderr[-300:])
    for seed in (11, 12):
        r = subprocess.run([os.path.join(ROOT, "tools", "fuzz", "fuzz_host"), "4", str(seed)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0 and "no sanitizer report" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


@pytest.mark.parametrize("container", (H.RAW, H.MT))
def test_index_build_is_not_sized_by_the_headers_claim(container):
    """A stream whose header claims 2^45 decoded bytes: the host index builder neither allocates that nor crashes; it returns 0
    (the walk finds the stream too short for its claim) or a plan for what is really there."""
    data = synth.enwik8_shaped(200_000, seed=3)
    stream = H.encode(container, 64, 11, data) if container == H.RAW else H.encode(container, 64, 11, data, block_size=65536)
    bad = stream.copy()
    bad[:8] = np.frombuffer(np.uint64(1 << 45).tobytes(), np.uint8)
    groups = np.array([64, 512], np.uint64)
    plan = np.zeros(1 << 20, np.uint8)
    n = api.load_library().hsrans_index_build_host(-1, 1, container, 64, 11, api._p(bad), bad.size, api._p(groups), groups.size, api._p(plan), plan.size)
    assert n == 0 or n <= plan.size
    # and the honest stream still indexes
    good = api.index_build_host(container, 64, 11, stream, groups)
    assert H.plan_chain_count(good) >= 3
