"""Build-time guard: the decode / encode kernels must not spill to scratch and the shared-table decode variants must keep
8 waves per SIMD (<= 64 VGPRs).  Reads the compiler's own resource report written by csrc/Makefile
(-Rpass-analysis=kernel-resource-usage).  A spilling variant still decodes correctly — only 3-4x slower — so nothing else
would notice."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "hypersonic_rans_amd", "csrc", "build")


def _report(name):
    path = os.path.join(BUILD, name + ".remarks")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "hypersonic_rans_amd", "csrc")])
    text = open(path).read()
    kernels = {}
    cur = None
    for line in text.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][\w /\[\]]*?): (\d+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return kernels


@pytest.mark.parametrize("unit", ("hsrans_kernels", "hsrans_encode"))
def test_no_scratch(unit):
    kernels = _report(unit)
    assert kernels, "no resource report found"
    for name, r in kernels.items():
        assert r["ScratchSize [bytes/lane]"] == 0, (name, r)


def test_shared_table_decode_occupancy():
    kernels = _report("hsrans_kernels")
    seen = 0
    for name, r in kernels.items():
        m = re.search(r"k_decodeILi(\d)ELb1E", name) or re.search(r"k_decode_directILi(\d)E", name) or re.search(r"k_decode_dualILi(\d)E", name)
        if not m or ("k_decodeI" in name and int(m.group(1)) not in (0, 1, 3, 4, 5)):
            continue
        seen += 1
        assert r["VGPRs"] <= 64 and r["Occupancy [waves/SIMD]"] == 8, (name, r)
    assert seen == 5 + 6 + 2
