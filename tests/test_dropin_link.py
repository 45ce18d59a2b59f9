"""The naming boundary: code written against the reference's own decoder names —
`rANS32x64_16w_decode_scalar_N` (src/rANS32x64_16w.h:48), `block_rANS32x64_16w_decode_N` (src/block_rANS32x64_16w.h:19),
`mt_rANS32x64_16w_decode_N` / `…_decode_mt_N` (src/mt_rANS32x64_16w.h:20-28) — and its codec registry shape (`codec_info_t`,
`encode_no_hist_wrapper`, `decode_with_thread_pool_wrapper`: src/main.cpp:146-170) compiles and links against this library
with nothing but `#include "hsrans_dropin.hpp"` + `using namespace hsrans_hip;`, and validates like main.cpp does
(decoded size == file size && memcmp, nonzero exit on mismatch: src/main.cpp:891-897).

The caller below is written for this test (the registry's SHAPE follows main.cpp; none of its text is the reference's).
Without a GPU the mt_ entries route to the host decoder on all cores, so the same binary runs in the CPU suite and, marked gpu,
on the MI355X box (where mt_ runs the kernels).  Also: every C-linkage alias of include/hsrans_names.h is exported and the ones
that need no GPU decode correctly through ctypes."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

import hypersonic_rans_amd as H
from hypersonic_rans_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CALLER = r"""
#include "hsrans_dropin.hpp"
#include <stdio.h>
#include <string.h>
#include <vector>
using namespace hsrans_hip; // the ONLY line a caller of the reference's headers adds

template <typename F> struct func_info_t { const char *name = nullptr; F func = nullptr; };
struct codec_info_t
{
  typedef size_t (*encodeFunc)(const uint8_t *pInData, const size_t length, uint8_t *pOutData, const size_t outCapacity, const hist_t *pHist);
  typedef size_t (*decodeFunc)(const uint8_t *pInData, const size_t inLength, uint8_t *pOutData, const size_t outCapacity);
  const char *name = "?";
  uint32_t totalSymbolCountBits = 0;
  func_info_t<encodeFunc> encoders[4];
  func_info_t<decodeFunc> decoders[4];
};
template <size_t (*func)(const uint8_t *, const size_t, uint8_t *, const size_t)>
size_t encode_no_hist_wrapper(const uint8_t *i, const size_t l, uint8_t *o, const size_t c, const hist_t *) { return func(i, l, o, c); }
template <size_t (*func)(const uint8_t *, const size_t, uint8_t *, const size_t, thread_pool *)>
size_t decode_with_thread_pool_wrapper(const uint8_t *i, const size_t l, uint8_t *o, const size_t c) { return func(i, l, o, c, nullptr); }

static const codec_info_t _Codecs[] = {
  {"rANS32x64 16w (raw)", 11, {{"enc scalar", rANS32x64_16w_encode_scalar_11}}, {{"dec scalar", rANS32x64_16w_decode_scalar_11}}},
  {"rANS32x32 16w (raw)", 14, {{"enc scalar", rANS32x32_16w_encode_scalar_14}}, {{"dec scalar", rANS32x32_16w_decode_scalar_14}}},
  {"rANS32x64 16w (variable block size)", 12, {{"encode", encode_no_hist_wrapper<block_rANS32x64_16w_encode_12>}}, {{"decode", block_rANS32x64_16w_decode_12}}},
  {"rANS32x32 16w (variable block size)", 10, {{"encode", encode_no_hist_wrapper<block_rANS32x32_16w_encode_10>}}, {{"decode", block_rANS32x32_16w_decode_10}}},
  {"rANS32x64 16w (independent blocks)", 11, {{"encode", encode_no_hist_wrapper<mt_rANS32x64_16w_encode_11>}},
   {{"decode (single thread)", mt_rANS32x64_16w_decode_11}, {"decode (multi threaded)", decode_with_thread_pool_wrapper<mt_rANS32x64_16w_decode_mt_11>}}},
  {"rANS32x32 16w (independent blocks)", 15, {{"encode", encode_no_hist_wrapper<mt_rANS32x32_16w_encode_15>}}, {{"decode (single thread)", mt_rANS32x32_16w_decode_15}}},
};

int main(int argc, char **argv)
{
  const bool gpu_expected = argc > 2 && strcmp(argv[2], "gpu") == 0;
  FILE *f = fopen(argv[1], "rb");
  if (!f) return 3;
  std::vector<uint8_t> file(1 << 22);
  file.resize(fread(file.data(), 1, file.size(), f));
  fclose(f);
  if (gpu_expected && default_context() == nullptr) { puts("no gfx950 device"); return 4; }
  int bad = 0;
  for (const codec_info_t &c : _Codecs)
  {
    hist_t hist;
    make_hist(&hist, file.data(), file.size(), c.totalSymbolCountBits); // main.cpp:746
    std::vector<uint8_t> comp(mt_rANS32x64_16w_capacity(file.size()) + 4096, 0xCC), out(file.size() + 64);
    const size_t n = c.encoders[0].func(file.data(), file.size(), comp.data(), comp.size(), &hist);
    for (const auto &d : c.decoders)
    {
      if (d.func == nullptr) break;
      memset(out.data(), 0xCC, out.size()); // main.cpp:860
      const size_t m = d.func(comp.data(), n, out.data(), file.size());
      const bool ok = n != 0 && m == file.size() && memcmp(out.data(), file.data(), file.size()) == 0 && out[file.size()] == 0xCC; // main.cpp:891
      printf("%-40s %2u  %-26s %zu -> %zu  %s\n", c.name, c.totalSymbolCountBits, d.name, n, m, ok ? "ok" : "MISMATCH");
      bad += !ok;
    }
  }
  return bad ? 1 : 0;
}
"""


def _build_and_run(tmp_path, mode):
    src = tmp_path / "caller.cpp"
    src.write_text(CALLER)
    exe = tmp_path / "caller"
    libdir = os.path.dirname(H.lib_path())
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L", libdir, "-lhsrans_hip",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True, capture_output=True, text=True)
    data = tmp_path / "file.bin"
    synth.enwik8_shaped(700_001, seed=77).tofile(data)
    r = subprocess.run([str(exe), str(data), mode], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count(" ok") == 7 and "MISMATCH" not in r.stdout, r.stdout
    return r.stdout


def test_reference_named_caller_compiles_links_and_validates(tmp_path):
    import torch

    _build_and_run(tmp_path, "gpu" if torch.cuda.is_available() else "cpu")


@pytest.mark.gpu
def test_reference_named_caller_on_the_gpu(tmp_path):
    """the same binary where mt_ entries launch the gfx950 kernels (fails, exit code 4, if the library found no device)"""
    _build_and_run(tmp_path, "gpu")


def _c_alias_names():
    hdr = open(os.path.join(ROOT, "include", "hsrans_names.h")).read()
    body = re.search(r"#define HSRANS_C_DECL\(S, N\)(.*?)#define HSRANS_C_DECL_BITS", hdr, re.S).group(1)
    templates = re.findall(r"size_t (hsrans_[A-Za-z0-9_#]+)\(", body)
    names = set(re.findall(r"size_t (hsrans_[A-Za-z0-9_]+_capacity)\(", hdr))
    for t in templates:
        for S in (32, 64):
            for N in range(10, 16):
                names.add(t.replace("##S##", str(S)).replace("##N", str(N)))
    return sorted(names)


def test_c_aliases_are_exported_and_decode():
    names = _c_alias_names()
    assert len(names) == 6 + 10 * 2 * 6
    lib = ctypes.CDLL(H.lib_path())
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/hsrans_names.h but not exported"
    d = synth.enwik8_shaped(200_000, seed=9)
    sig = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    for enc, dec, cap, hist in (("hsrans_rANS32x64_16w_encode_scalar_11", "hsrans_rANS32x64_16w_decode_scalar_11", "hsrans_rANS32x64_16w_capacity", True),
                                ("hsrans_block_rANS32x32_16w_encode_13", "hsrans_block_rANS32x32_16w_decode_13", "hsrans_block_rANS32x32_16w_capacity", False)):
        fcap, fenc, fdec = getattr(lib, cap), getattr(lib, enc), getattr(lib, dec)
        fcap.restype = fenc.restype = fdec.restype = ctypes.c_size_t
        fcap.argtypes = [ctypes.c_size_t]
        fenc.argtypes = sig + ([ctypes.c_void_p] if hist else [])
        fdec.argtypes = sig
        comp = np.zeros(fcap(d.size), np.uint8)
        h = H.make_hist(d, 11)
        n = fenc(d.ctypes.data, d.size, comp.ctypes.data, comp.size, ctypes.addressof(h)) if hist else fenc(d.ctypes.data, d.size, comp.ctypes.data, comp.size)
        assert n > 0
        got = np.full(d.size, 0xCC, np.uint8)
        assert fdec(comp.ctypes.data, n, got.ctypes.data, d.size) == d.size and np.array_equal(got, d)
        assert fdec(comp.ctypes.data, n, got.ctypes.data, d.size - 1) == 0
