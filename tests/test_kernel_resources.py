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
        if "k_decode_dual" in name:  # one 16-wave workgroup per CU (LDS-limited): 4 waves per SIMD; the asm loop pins v64-v79
            assert r["VGPRs"] <= 128 and r["Occupancy [waves/SIMD]"] >= 4, (name, r)
        else:
            assert r["VGPRs"] <= 64 and r["Occupancy [waves/SIMD]"] == 8, (name, r)
    assert seen == 5 + 6 + 2
    # the grouped launches' kernel (block_/mt_ plans with checkpoints): the 8-byte-table instantiations, lean and general
    grouped = {name: r for name, r in kernels.items() if re.search(r"k_decode_groupedILi[34]ELb[01]E", name)}  # 8-byte table and rank table
    assert len(grouped) == 6  # + the two that count their groups into a sharded decode's sub-runs (round 6: <3, true, true>, <4, true, true>)
    spread = {name: r for name, r in kernels.items() if "k_decode_spread" in name}
    assert len(spread) == 2  # k_decode_spread<3> and its sub-run-counting twin
    grouped.update(spread)
    for name, r in grouped.items():
        assert r["VGPRs"] <= 64 and r["Occupancy [waves/SIMD]"] == 8, (name, r)
    # the batch launches (round 5): the one-chain-per-wave form, its calibration twin, the grouped form
    batch = {name: r for name, r in kernels.items() if "k_decode_batch" in name or "k_calibrate_batch" in name or "k_decode_grouped_batch" in name}
    assert len(batch) == 6, sorted(batch)  # k_decode_batch<3>, k_decode_batch_pair<3>, k_decode_batch_dual<3 / 4>, k_calibrate_batch, k_decode_grouped_batch<3>
    for name, r in batch.items():
        if "k_decode_batch_dual" in name:  # k_decode_dual's shape: one 16-wave workgroup per CU, the asm loop pins v64-v79
            assert r["VGPRs"] <= 128 and r["Occupancy [waves/SIMD]"] >= 4, (name, r)
        else:
            assert r["VGPRs"] <= 64 and r["Occupancy [waves/SIMD]"] == 8, (name, r)
    persist = [r for name, r in kernels.items() if "k_decode_persist" in name]
    assert len(persist) == 2 and all(r["VGPRs"] <= 64 and r["Occupancy [waves/SIMD]"] == 8 for r in persist), persist  # 8-byte table and rank table


# ---------------------------------------------------------------------------------------------------------------
# The decode loops must not start with a wait on the whole vector-memory queue.  The chain's states come from an ordinary
# load and are first used inside the loop; left alone the compiler parks its "s_waitcnt vmcnt(0)" for that load on the loop
# header, where it runs every 4 groups and drains the previous store and every stream request in flight (measured: 61 us
# instead of 49 us for the 100 MB decode when the stream comes from HBM; ring_ready(x) in hsrans_kernels.hip is the fix).
# Checked on the ISA of the built object: extract the gfx950 code object, disassemble, find the innermost loops that hold 4+
# decode groups (v_mbcnt_hi is the group's rank instruction) and look at their first instructions.
# ---------------------------------------------------------------------------------------------------------------
LLVM = "/opt/rocm/lib/llvm/bin"


def _disassemble(tmp_path):
    obj = os.path.join(BUILD, "hsrans_kernels.o")
    if not os.path.exists(obj):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "hypersonic_rans_amd", "csrc")])
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "k.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--type=o", "--unbundle", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    return subprocess.check_output([LLVM + "/llvm-objdump", "-d", co], text=True)


@pytest.mark.skipif(not os.path.exists(LLVM + "/llvm-objdump"), reason="needs the ROCm LLVM tools")
def test_decode_loops_do_not_drain_the_memory_queue(tmp_path):
    text = _disassemble(tmp_path)
    funcs = {}  # name -> list of (address, mnemonic + operands, branch target or None)
    cur, base = None, 0
    for line in text.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            base, cur = int(m.group(1), 16), funcs.setdefault(m.group(2), [])
            continue
        m = re.match(r"^\s+(\S.*?)\s+// ([0-9A-F]+): ", line)
        if not m or cur is None:
            continue
        target = None
        t = re.search(r"<\S+\+0x([0-9a-f]+)>\s*$", line)
        if t and m.group(1).startswith(("s_branch", "s_cbranch")):
            target = base + int(t.group(1), 16)
        cur.append((int(m.group(2), 16), m.group(1), target))
    checked = 0
    for name, ins in funcs.items():
        if not re.search(r"k_decode(_direct|_dual)?ILi[0-4]E", name) and "k_decode_persist" not in name and "k_decode_single" not in name:  # (table mode 5 gathers its table from global memory: it has to wait)
            continue
        addr_index = {a: i for i, (a, _, _) in enumerate(ins)}
        loops = [(addr_index[t], i) for i, (a, _, t) in enumerate(ins) if t is not None and t < a and t in addr_index]
        groups = lambda lo, hi: sum(1 for _, s, _ in ins[lo:hi + 1] if s.startswith("v_mbcnt_hi"))
        for lo, hi in loops:
            if groups(lo, hi) < 4 or any((l2, h2) != (lo, hi) and l2 >= lo and h2 <= hi and groups(l2, h2) >= 4 for l2, h2 in loops):
                continue
            head = [s for _, s, _ in ins[lo:lo + 8]]
            assert not any(s.startswith("s_waitcnt") and "vmcnt" in s for s in head), (name, hex(ins[lo][0]), head)
            checked += 1
    assert checked >= 14, checked
