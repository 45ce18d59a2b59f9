"""Deterministic, integer-only synthetic inputs for the rANS decode benchmarks and tests.

The reference benchmarks on real files (enwik8, Silesia; README.md:19-27 of the reference); neither is available
offline, so SURVEY.md §8(d) defines "enwik8-shaped" data: i.i.d. bytes over 205 symbols with Zipf(1.2) weights
(order-0 entropy 5.14 bit/B -> ratio ~64.3 % at 11 bits, vs 64.48 % for real enwik8).  Everything here is integer
arithmetic on a counter-based splitmix64, so this container and the GPU box regenerate identical bytes.
"""
from __future__ import annotations

import numpy as np

# floor(2^32 / r^1.2), r = 1..205 — frozen literal table (never recomputed with floating point at run time)
ZIPF12_WEIGHTS = (
    4294967296, 1869493099, 1149249584, 813744135, 622580990, 500239936, 415759530, 354202707, 307516802, 270994115,
    241706672, 217742079, 197800706, 180969846, 166590545, 154175683, 143357743, 133854462, 125445630, 117957040,
    111249151, 105208939, 99743925, 94777744, 90246808, 86097758, 82285506, 78771700, 75523527, 72512746,
    69714935, 67108864, 64675997, 62400082, 60266810, 58263538, 56379055, 54603382, 52927616, 51343783,
    49844728, 48424005, 47075800, 45794850, 44576384, 43416065, 42309942, 41254409, 40246170, 39282204,
    38359739, 37476225, 36629316, 35816846, 35036816, 34287374, 33566807, 32873524, 32206048, 31563006,
    30943116, 30345188, 29768105, 29210829, 28672385, 28151862, 27648405, 27161213, 26689531, 26232652,
    25789910, 25360678, 24944363, 24540409, 24148287, 23767502, 23397582, 23038082, 22688581, 22348679,
    22017998, 21696178, 21382877, 21077772, 20780555, 20490932, 20208624, 19933366, 19664905, 19402998,
    19147416, 18897940, 18654358, 18416471, 18184088, 17957024, 17735105, 17518163, 17306036, 17098572,
    16895623, 16697046, 16502707, 16312474, 16126224, 15943836, 15765194, 15590188, 15418710, 15250660,
    15085936, 14924446, 14766097, 14610801, 14458473, 14309032, 14162399, 14018496, 13877253, 13738596,
    13602458, 13468773, 13337478, 13208510, 13081810, 12957320, 12834985, 12714752, 12596567, 12480380,
    12366144, 12253810, 12143332, 12034667, 11927772, 11822604, 11719124, 11617293, 11517072, 11418425,
    11321316, 11225710, 11131574, 11038876, 10947583, 10857664, 10769091, 10681833, 10595863, 10511152,
    10427675, 10345406, 10264319, 10184389, 10105593, 10027908, 9951310, 9875778, 9801291, 9727827,
    9655367, 9583890, 9513377, 9443810, 9375169, 9307438, 9240598, 9174633, 9109526, 9045262,
    8981823, 8919196, 8857364, 8796314, 8736031, 8676501, 8617711, 8559647, 8502296, 8445645,
    8389683, 8334397, 8279775, 8225806, 8172478, 8119781, 8067703, 8016235, 7965365, 7915084,
    7865382, 7816249, 7767675, 7719653, 7672171, 7625223, 7578799, 7532890, 7487488, 7442586,
    7398174, 7354247, 7310795, 7267811, 7225289,
)
assert len(ZIPF12_WEIGHTS) == 205

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_CHUNK = 1 << 24


def splitmix64(seed: int, start: int, count: int) -> np.ndarray:
    """Counter-based splitmix64: element i is mix(seed + (i+1) * golden), i in [start, start+count)."""
    with np.errstate(over="ignore"):
        z = (np.arange(start + 1, start + count + 1, dtype=np.uint64) * _GOLDEN) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _permutation(seed: int, n: int = 256) -> np.ndarray:
    """Fixed byte permutation: argsort of splitmix keys (ties impossible in practice; argsort kind is stable)."""
    keys = splitmix64(seed, 0, n)
    return np.argsort(keys, kind="stable").astype(np.uint8)


def uniform_bytes(n: int, seed: int = 1) -> np.ndarray:
    """BASELINE config 1 input: n uniform random bytes."""
    out = np.empty(n, dtype=np.uint8)
    for s in range(0, n, _CHUNK):
        c = min(_CHUNK, n - s)
        out[s:s + c] = (splitmix64(seed, s, c) >> np.uint64(56)).astype(np.uint8)
    return out


def enwik8_shaped(n: int, seed: int = 20241008, perm_seed: int = 7) -> np.ndarray:
    """i.i.d. Zipf(1.2) over 205 symbols, integer inverse-CDF sampling, rank -> byte through a fixed permutation."""
    w = np.array(ZIPF12_WEIGHTS, dtype=np.uint64)
    cdf = np.cumsum(w)  # inclusive, < 2^35
    total = cdf[-1]
    perm = _permutation(perm_seed)[:205]
    out = np.empty(n, dtype=np.uint8)
    for s in range(0, n, _CHUNK):
        c = min(_CHUNK, n - s)
        u = splitmix64(seed, s, c) >> np.uint64(40)  # 24 uniform bits; u * total < 2^59
        t = (u * total) >> np.uint64(24)
        r = np.searchsorted(cdf, t, side="right")
        out[s:s + c] = perm[r]
    return out


def nonstationary(n: int, seed: int = 99) -> np.ndarray:
    """Segments (64 KiB .. 1 MiB) of Zipf data under different byte permutations, with >=128 KiB single-byte runs
    interleaved: exercises histogram switching and the single-symbol block path of the block_/mt_ containers."""
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    k = 0
    while pos < n:
        r = int(splitmix64(seed ^ 0xABCDEF, k, 1)[0])
        seg = min(n - pos, (1 << 16) * (1 + (r & 15)))
        if k % 5 == 3:
            seg = min(n - pos, (1 << 17) + (r >> 8) % (1 << 17))
            out[pos:pos + seg] = (r >> 40) & 0xFF
        else:
            out[pos:pos + seg] = enwik8_shaped(seg, seed=seed + k, perm_seed=11 + (k % 7))
        pos += seg
        k += 1
    return out


def two_symbol(n: int, seed: int = 5) -> np.ndarray:
    """Degenerate histogram: two symbols, 15:1."""
    u = splitmix64(seed, 0, n) >> np.uint64(60)
    return np.where(u == 0, np.uint8(0x41), np.uint8(0x7A)).astype(np.uint8)


def zipf_bytes(n: int, exponent: float, symbols: int = 256, seed: int = 31, perm_seed: int = 7) -> np.ndarray:
    """i.i.d. bytes over `symbols` symbols with weights 1 / rank^exponent (exponent 0 = uniform over the symbols).  The weights come
    from floating point at run time: for sweeps that generate, encode and check in one process (tools/ratio_sweep.py), NOT for
    fixtures that have to be regenerated identically elsewhere (those use the frozen table of enwik8_shaped)."""
    w = np.floor(2.0**32 / np.arange(1, symbols + 1, dtype=np.float64) ** exponent).astype(np.uint64)
    cdf = np.cumsum(w)
    total = cdf[-1]
    perm = _permutation(perm_seed)[:symbols]
    out = np.empty(n, dtype=np.uint8)
    for s0 in range(0, n, _CHUNK):
        c = min(_CHUNK, n - s0)
        u = splitmix64(seed, s0, c) >> np.uint64(40)
        t = (u * total) >> np.uint64(24)
        out[s0:s0 + c] = perm[np.searchsorted(cdf, t, side="right")]
    return out
