"""The host-side dealing of k_decode_dealt (hsrans_dealt_shares; csrc/hsrans_kernels.hip deal_shares): pure arithmetic, no GPU.  The launch
decodes block_/mt_ plans with checkpoints in one round — the reference hands every block of a stream to its pool in one pass
(/root/reference/src/mt_rANS32x64_16w_decode.cpp:182-224) — and every workgroup may hold two decode tables: a share must never reach
into a third block, every chain must be dealt exactly once, and the shares must follow the workgroups' scheduling weights."""
import numpy as np
import pytest

import hypersonic_rans_amd as H

GRID = 512


def blocks(n_chains, per_block):
    return np.concatenate([np.arange(0, n_chains, per_block), [n_chains]]).astype(np.uint32)


@pytest.mark.parametrize("size,block,interval", (
    (100_000_000, 1 << 18, 8), (100_000_000, 1 << 18, 32), (100_000_000, 1 << 18, 128), (1 << 27, 1 << 18, 16), (1 << 27, 1 << 18, 64),
    (24_000_000, 1 << 18, 8), (40_000_000, 1 << 20, 16), (1 << 25, 1 << 17, 16), (50_000_000, 3 << 16, 4)))
def test_shares_cover_every_chain_once_and_touch_two_blocks_at_most(size, block, interval):
    groups = size // 64
    bb, c = [], 0
    for g0 in range(0, groups, block // 64):
        bb.append(c)
        c += -(-min(block // 64, groups - g0) // interval)
    n_chains = c
    bb = np.array(bb + [c], np.uint32)
    ok, begin, split = H.api.dealt_shares(bb, groups)
    assert ok, (size, block, interval)
    assert begin[0] == 0 and begin[GRID] == n_chains and (np.diff(begin.astype(np.int64)) >= 0).all()
    blk_of = np.repeat(np.arange(bb.size - 1), np.diff(bb))
    shares = np.diff(begin[: GRID + 1].astype(np.int64))
    for b in range(GRID):
        c0, c1 = int(begin[b]), int(begin[b + 1])
        if c1 == c0:
            continue
        touched = np.unique(blk_of[c0:c1])
        assert touched.size <= 2, (b, c0, c1)
        if touched.size == 2:  # split = where the second block starts inside the share
            sp = int(split[b])
            assert sp < c1 - c0 and blk_of[c0 + sp - 1] == touched[0] and blk_of[c0 + sp] == touched[1]
        else:
            assert split[b] >= c1 - c0
    # the older half of the grid gets more than the younger one, and inside a half the shares are even (the cuts' losses are re-spread)
    first, second = shares[: GRID // 2], shares[GRID // 2:]
    assert first.mean() > 1.2 * second.mean()
    for half in (first, second):
        assert half.max() <= 1.12 * half.mean() + 1, (half.max(), half.mean())


def test_plans_the_launch_does_not_suit():
    groups = 100_000_000 // 64
    # one chain per wave and fewer: nothing for the class weights to work with
    assert not H.api.dealt_shares(blocks(8192, 16), groups)[0]
    assert not H.api.dealt_shares(blocks(3000, 30), groups)[0]
    # a block and more per 8-wave workgroup slot (1,024 of them): the grouped launch's home ground (100 MB in 64 KiB blocks: 1,526)
    assert not H.api.dealt_shares(blocks(1024 * 64, 64), 1 << 24)[0]
    assert not H.api.dealt_shares(blocks(48_829, 32), groups)[0]
    # 900 blocks for 512 workgroups: the older half's shares of 2.2 blocks are cut at two, what they lose bends the rest too far
    assert not H.api.dealt_shares(blocks(900 * 40, 40), groups)[0]
    with pytest.raises(H.HsransError):
        H.api.dealt_shares(np.array([0, 10, 10, 20], np.uint32), 1000)   # an empty block
