#!/usr/bin/env python3
"""Generates the golden vectors in this directory from the REAL reference (oracle/_ref/libhsrans_ref.so, built from
/root/reference/src by oracle/Makefile).  Run in the build container only; the vectors (data, not code) are committed.

  small_vectors.npz   for every container x states x bits: the reference-ENCODED stream of a small seeded input and what
                      the reference DECODER returns for it (+ a few "quirk" lengths where the reference's own round trip
                      is wrong, SURVEY.md §8 quirks: the bytes held here are what the reference produces)
  manifest.json       SHA-256 of stream / decoded output for 1 MiB inputs (too big to commit), raw histogram counts,
                      and the reference's capacity() values
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from hypersonic_rans_amd import synth  # noqa: E402  (pure numpy data generator)
from oracle_lib import BLOCK, MT, RAW, Ref  # noqa: E402

NAMES = {RAW: "raw", BLOCK: "block", MT: "mt"}


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    ref = Ref()
    small = {}
    manifest = {"source": "rainerzufalldererste/hypersonic-rANS @ 2024_10_08, clang++ -O3 -std=c++20 -mxsave (oracle/Makefile)", "large": [], "capacity": [],
                "hist": []}
    zipf_small = synth.enwik8_shaped(3000, seed=101)
    two = synth.two_symbol(3000, seed=5)
    big = synth.enwik8_shaped(1 << 20, seed=1)
    uni = synth.uniform_bytes(1 << 20, seed=1)
    nonstat = synth.nonstationary(1 << 20, seed=99)
    for cont in (RAW, BLOCK, MT):
        for S in (32, 64):
            for n in (0, 1, 63, 64, 65, 1000, 65536, 100_000_000):
                manifest["capacity"].append({"container": NAMES[cont], "states": S, "n": n, "capacity": int(ref.capacity(cont, S, n))})
            for bits in range(10, 16):
                for tag, data in (("zipf3000", zipf_small), ("two3000", two), ("zipf127", zipf_small[:127]), ("zipf64", zipf_small[:64])):
                    if cont != RAW and tag in ("zipf127", "zipf64") and bits != 11:
                        continue
                    s = ref.encode(cont, S, bits, data)
                    r, dec = ref.decode(cont, S, bits, s, data.size)
                    assert r == data.size and np.array_equal(dec, data), (cont, S, bits, tag)
                    key = f"{NAMES[cont]}_s{S}_b{bits}_{tag}"
                    small[key + "_in"] = data.copy()
                    small[key + "_stream"] = s
                # 1 MiB inputs: hashes only
                for tag, data in (("zipf1M_seed1", big), ("uniform1M_seed1", uni), ("nonstat1M_seed99", nonstat)):
                    if cont != RAW and bits not in (11, 14) and tag != "nonstat1M_seed99":
                        continue
                    s = ref.encode(cont, S, bits, data)
                    r, dec = ref.decode(cont, S, bits, s, data.size)
                    assert r == data.size and np.array_equal(dec, data)
                    manifest["large"].append({"container": NAMES[cont], "states": S, "bits": bits, "input": tag, "n": int(data.size),
                                              "stream_len": int(s.size), "stream_sha256": sha(s), "decoded_sha256": sha(dec)})
    # quirk lengths: reference round trip is wrong for MinBlockSize < n < MinBlockSize + S; keep what its decoder returns
    zq = synth.enwik8_shaped(65600, seed=7)
    for cont, S, bits, n in ((MT, 64, 11, 65560), (MT, 32, 11, 65550), (MT, 64, 11, 65599), (BLOCK, 64, 12, 65560)):
        d = zq[:n]
        s = ref.encode(cont, S, bits, d)
        r, dec = ref.decode(cont, S, bits, s, n)
        key = f"quirk_{NAMES[cont]}_s{S}_b{bits}_n{n}"
        small[key + "_stream"] = s
        small[key + "_decoded"] = dec.copy()
        small[key + "_in_sha"] = np.frombuffer(bytes.fromhex(sha(d)), dtype=np.uint8)
        manifest.setdefault("quirks", []).append({"key": key, "returned": int(r), "round_trip_ok": bool(np.array_equal(dec, d)),
                                                   "mismatching_bytes": int((dec != d).sum())})
    for bits in range(10, 16):
        for tag, data in (("zipf1M_seed1", big), ("uniform1M_seed1", uni), ("two3000", two)):
            counts, cumul = ref.make_hist(data, bits)
            manifest["hist"].append({"input": tag, "bits": bits, "counts": [int(c) for c in counts]})
    np.savez_compressed(os.path.join(HERE, "small_vectors.npz"), **small)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0)
    print("wrote", len(small), "arrays;", os.path.getsize(os.path.join(HERE, "small_vectors.npz")), "bytes npz")


if __name__ == "__main__":
    main()
