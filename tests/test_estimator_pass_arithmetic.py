"""CPU checks of the integer arithmetic the round-4 estimator passes rely on (csrc/estimator_log.hpp: accumulate_blocks_kernel;
csrc/estimator_partition.hpp: partition_kernel).  The kernels themselves are held to the oracle on the GPU
(tests/test_estimator_pipelines.py); here the index formulas they use are restated in NumPy and checked exhaustively, so that a
reader can see WHY they are right without a GPU:

  * block sums: a trace that starts at line `a` of its tile and passes `n` lines is split into h lines before the first 8-line
    boundary, b aligned 8-line blocks and t lines after the last boundary; item j of the h + b + t items is a line or a block by the
    branch-free formulas of add_item().  Every line of [a, a + n) must be covered exactly once -- by its own item or by its block's --
    and nothing outside;
  * the flush adds a line's own accumulator and its block's: with all constants 1 the result is the number of traces over the line;
  * partition ranks: lanes of a wave with equal keys find each other by ballots over the key bits; leader + popcount-below gives every
    record a unique rank inside its bucket, and off[bucket] + rank is a permutation of the segment.
"""
import numpy as np

TILE, APRON = 2048, 256


def _decompose(a, n):
    h = np.minimum(n, (8 - (a & 7)) & 7)
    b = (n - h) >> 3
    t = (n - h) & 7
    return h, b, t


def _item(a, h, b, j):
    """add_item() of accumulate_blocks_kernel: (is_block, first line covered, lines covered)."""
    jh = (j - h) & 0xFFFFFFFF                      # unsigned wrap for j < h
    is_blk = jh < b
    line = a + j + np.where(j >= h + b, 7 * b, 0)
    blk_line = a + h + 8 * jh
    return is_blk, np.where(is_blk, blk_line, line), np.where(is_blk, 8, 1)


def test_every_line_of_a_trace_is_covered_exactly_once():
    for a in range(0, 64):                      # (the formulas depend on a only through a & 7 and additively: 64 starts cover all cases)
        for n in list(range(1, 300)) + [511, 512, 513, 1000, 2047]:
            a_, n_ = np.uint32(a + 1024), np.uint32(n)
            h, b, t = _decompose(a_, n_)
            m = int(h + b + t)
            assert m <= 7 + 7 + n // 8 + 1
            cover = np.zeros(a + 1024 + n + 16, dtype=np.int32)
            j = np.arange(m, dtype=np.int64)
            is_blk, first, cnt = _item(np.int64(a_), np.int64(h), np.int64(b), j)
            assert int(is_blk.sum()) == int(b)
            assert np.all(first[is_blk] % 8 == 0)                          # blocks are aligned (relative to the tile)
            for f, c in zip(first, cnt):
                cover[f:f + c] += 1
            assert np.all(cover[a + 1024:a + 1024 + n] == 1), (a, n)
            assert cover.sum() == n, (a, n)


def test_short_records_fit_the_pooled_path():
    """Records of up to 255 lines (ACCB_LONG) have at most 45 items: 64 of them fill at most 45 passes (ACCB_PASSES = 48), and the
    staged word a | h << 12 | b << 15 holds a < 2048, h < 8, b < 32."""
    a = np.arange(TILE, dtype=np.uint32)[:, None]
    n = np.arange(1, 256, dtype=np.uint32)[None, :]
    h, b, t = _decompose(a, n)
    assert int((h + b + t).max()) <= 45 and int(h.max()) <= 7 and int(b.max()) <= 31
    word = a | (h << 12) | (b << 15)
    assert np.array_equal(word & 0xFFF, np.broadcast_to(a, word.shape)) and np.array_equal((word >> 12) & 7, h) and np.array_equal(word >> 15, b)


def test_flush_of_line_and_block_accumulators_counts_the_traces_over_a_line():
    rng = np.random.default_rng(3)
    n_rec = 4000
    a = rng.integers(0, TILE, n_rec).astype(np.int64)
    n = np.minimum(rng.geometric(1 / 41.0, n_rec), TILE + APRON - a).astype(np.int64)   # (kept inside the tile + apron here)
    lines = np.zeros(TILE + APRON, dtype=np.int64)
    blocks = np.zeros((TILE + APRON) // 8, dtype=np.int64)
    want = np.zeros(TILE + APRON, dtype=np.int64)
    for ai, ni in zip(a, n):
        want[ai:ai + ni] += 1
        h, b, t = (int(x) for x in _decompose(np.uint32(ai), np.uint32(ni)))
        j = np.arange(h + b + t, dtype=np.int64)
        is_blk, first, _ = _item(ai, h, b, j)
        np.add.at(blocks, first[is_blk] >> 3, 1)
        np.add.at(lines, first[~is_blk], 1)
    got = lines + blocks[np.arange(TILE + APRON) >> 3]
    assert np.array_equal(got, want)
    assert 0.2 < (lines.sum() + blocks.sum()) / n.sum() < 0.45           # ~11 adds for ~41 visits


def _ranks_by_ballots(keys, valid, key_bits):
    """The rank loop of partition_kernel for one wave iteration, restated: returns (leader lane, rank in group, group size) per lane."""
    lanes = np.arange(64)
    peers = np.repeat(valid[None, :], 64, axis=0)                          # peers[l, m]: lane m is still a candidate peer of lane l
    for bit in range(key_bits):
        mine = (keys >> bit) & 1
        ballot = valid & (((keys >> bit) & 1) == 1)
        peers &= np.where(mine[:, None] == 1, ballot[None, :], ~ballot[None, :])
    leader = np.argmax(peers, axis=1)
    below = (peers & (lanes[None, :] < lanes[:, None])).sum(axis=1)
    return leader, below, peers.sum(axis=1)


def test_ballot_ranking_gives_a_permutation_of_the_segment():
    rng = np.random.default_rng(5)
    for n_buckets, key_bits in ((20, 5), (245, 10), (1024, 10), (1, 0)):
        n = 2048 - 37                                                       # a ragged last wave iteration
        keys_all = rng.integers(0, n_buckets, n)
        keys_all[:300] = keys_all[0]                                        # a crowded bucket
        hist = np.zeros(n_buckets, dtype=np.int64)
        rank = np.zeros(n, dtype=np.int64)
        for i0 in range(0, n, 64):
            idx = i0 + np.arange(64)
            valid = idx < n
            keys = np.where(valid, keys_all[np.minimum(idx, n - 1)], 0)
            leader, below, size = _ranks_by_ballots(keys, valid, key_bits)
            start = np.zeros(64, dtype=np.int64)
            for lane in np.flatnonzero(valid):
                if leader[lane] == lane:                                    # one LDS atomic per distinct bucket and wave
                    start[lane] = hist[keys[lane]]
                    hist[keys[lane]] += size[lane]
            start = start[leader]
            rank[idx[valid]] = (start + below)[valid]
            for lane in np.flatnonzero(valid):                              # peers are exactly the valid lanes with the same key
                assert size[lane] == int((valid & (keys == keys[lane])).sum())
        off = np.concatenate([[0], np.cumsum(np.bincount(keys_all, minlength=n_buckets))])[:-1]
        pos = off[keys_all] + rank
        assert np.array_equal(np.sort(pos), np.arange(n))                   # a permutation ...
        assert np.all(np.diff(keys_all[np.argsort(pos)]) >= 0)              # ... that groups the records by bucket


# ---- round 6: the dyadic hierarchy of accumulate_dyadic_kernel (csrc/estimator_log.hpp), restated
TOP = 5


def _split(a, n):
    e = a + n
    d = int(a ^ e).bit_length() - 1
    M = (e >> d) << d
    return M - a, e - M


def _nth_bit(v, j):
    """Position of the j-th lowest set bit of v (the kernel's 32-entry table)."""
    pos = [b for b in range(5) if (v >> b) & 1]
    return pos[j]


def _dyadic_item(a, up, dn, j):
    """add_item() of accumulate_dyadic_kernel: (level, first line covered)."""
    u5, d5 = up & 31, dn & 31
    pu, mid = bin(u5).count("1"), (up >> 5) + (dn >> 5)
    if j < pu:
        level = _nth_bit(u5, j)
        return level, a + (u5 & ((1 << level) - 1))
    if j - pu < mid:
        return TOP, a + u5 + ((j - pu) << TOP)
    level = _nth_bit(d5, j - pu - mid)
    return level, a + up + dn - (d5 & ((2 << level) - 1))


def test_dyadic_items_tile_a_trace_with_aligned_blocks():
    """Every line of [a, a + n) is covered exactly once, every block is aligned to its own size (relative to the tile: tiles start on
    multiples of 2048 lines), nothing outside is touched, and the item count is what the kernel's n_items() says."""
    worst = 0
    for a in list(range(0, 130)) + [1023, 1024, 2015, 2047]:
        for n in list(range(1, 300)) + [511, 512, 513, 1000, 4097]:
            up, dn = _split(a, n)
            assert up >= 1 and dn >= 0 and up + dn == n
            m = bin(up & 31).count("1") + (up >> 5) + (dn >> 5) + bin(dn & 31).count("1")
            cover = np.zeros(a + n + 64, dtype=np.int32)
            for j in range(m):
                level, pos = _dyadic_item(a, up, dn, j)
                assert 0 <= level <= TOP and pos % (1 << level) == 0, (a, n, j)
                cover[pos:pos + (1 << level)] += 1
            assert np.all(cover[a:a + n] == 1) and cover.sum() == n, (a, n)
            if n <= 255:
                worst = max(worst, m)
                assert up < 256 and dn < 256 and a < 2048 or a >= 2048  # (the staged word a | up << 11 | dn << 19 holds them: 11 + 8 + 8 bits)
    assert worst <= 18  # 64 records of <= 255 lines: <= 18 passes of 64 items (ACCD_PASSES = 20)


def test_dyadic_cells_and_flush():
    """The accumulators are laid out like a bottom-up segment tree: the block of 2^l lines at line p is cell (T + p) >> l.  T is a multiple of 2^TOP, so the
    levels do not overlap (level l fills [T >> l, 2 T >> l)), and a line's flush -- the sum of the six cells above it -- counts every trace over the line
    exactly once."""
    T = TILE + APRON
    assert T % 256 == 0
    cells = 2 * T
    for l in range(TOP + 1):
        lo, hi = T >> l, (2 * T) >> l
        assert (T + 0) >> l == lo and (T + T - 1) >> l == hi - 1
        if l < TOP:
            assert (T >> (l + 1), (2 * T) >> (l + 1)) == (lo >> 1, lo)  # the next level ends where this one begins
    rng = np.random.default_rng(3)
    acc = np.zeros(cells)
    truth = np.zeros(T)
    for _ in range(4000):
        a = int(rng.integers(0, TILE)); n = int(min(rng.geometric(1 / 40.0), T - a))
        up, dn = _split(a, n)
        m = bin(up & 31).count("1") + (up >> 5) + (dn >> 5) + bin(dn & 31).count("1")
        for j in range(m):
            level, pos = _dyadic_item(a, up, dn, j)
            assert pos % (1 << level) == 0
            acc[(T + pos) >> level] += 1.0
        truth[a:a + n] += 1.0
    k = np.arange(T)
    got = sum(acc[(T + k) >> l] for l in range(TOP + 1))
    assert np.array_equal(got, truth)
    assert not acc[:T >> TOP].any()


def test_the_greedy_walk_visits_the_items_blocks():
    """accumulate_dyadic_kernel<.., LOOP> (est_accumulate 3) has no item list: a lane walks its record from the left, at line p the block of
    min(ctz(p | 32), floor(log2(end - p))) levels.  Same blocks as the item list -- ascending to the split point, 32-line blocks across it, descending
    behind it (the list numbers those from the smallest up, the walk meets them from the largest down)."""
    for a in list(range(0, 70)) + [1023, 1024, 2015, 2047]:
        for n in range(1, 256):
            up, dn = _split(a, n)
            m = bin(up & 31).count("1") + (up >> 5) + (dn >> 5) + bin(dn & 31).count("1")
            items = [_dyadic_item(a, up, dn, j) for j in range(m)]
            walk, p, e = [], a, a + n
            while p < e:
                ctz = ((p | (1 << TOP)) & -(p | (1 << TOP))).bit_length() - 1
                lv = min(ctz, (e - p).bit_length() - 1)
                walk.append((lv, p))
                p += 1 << lv
            assert sorted(walk) == sorted(items) and len(set(walk)) == len(walk), (a, n)
