"""TEST INFRASTRUCTURE (oracle): plain-Python restatement of the black-body packet source's random streams.

Restates, with Python integers, what NumPy's Generator(PCG64) does for the three draws of
BlackBodySimpleSource.create_packets (tardis/transport/montecarlo/packet_source/base.py:195-253,
black_body.py:140-222) -- the algorithm lives in the reference's pinned dependency NumPy (>= 1.17 Generator API;
numpy/random/src/pcg64/pcg64.h, numpy/random/src/distributions/distributions.c, numpy/random/bit_generator.pyx):

  * SeedSequence(seed) -> PCG64 state/inc                                   (`seed_state`)
  * rng.choice(max_val, n) = bounded 32-bit Lemire draws with rejection      (`bounded_uint32`)
  * rng.random(k)          = (next_uint64 >> 11) * 2**-53                    (`doubles`)
  * stream positions addressed by LCG jump-ahead exactly as the HIP kernels do (`jump`)

Pinned against numpy itself in tests/test_packet_source.py (numpy is importable wherever the tests run).
Small cases only: everything is a Python loop.
"""
from __future__ import annotations

import numpy as np

M128 = (1 << 128) - 1
PCG_MULT = (2549297995355413924 << 64) | 4865540595714422341


def _u32(x):
    return x & 0xFFFFFFFF


def seed_state(seed: int):
    """(state, inc) of np.random.PCG64(seed) -- SeedSequence pool of 4, generate_state(4, uint64), pcg64_set_seed."""
    INIT_A, MULT_A, INIT_B, MULT_B, MIX_L, MIX_R = 0x43B0D7E5, 0x931E8875, 0x8B51F9DD, 0x58F38DED, 0xCA01F9DD, 0x4973F715
    entropy = [_u32(seed)] if seed < (1 << 32) else [_u32(seed), _u32(seed >> 32)]
    hc = [INIT_A]

    def hashmix(v):
        v ^= hc[0]
        hc[0] = _u32(hc[0] * MULT_A)
        v = _u32(v * hc[0])
        return v ^ (v >> 16)

    def mix(x, y):
        r = _u32(MIX_L * x - MIX_R * y)
        return r ^ (r >> 16)

    pool = [hashmix(entropy[i] if i < len(entropy) else 0) for i in range(4)]
    for i_src in range(4):
        for i_dst in range(4):
            if i_src != i_dst:
                pool[i_dst] = mix(pool[i_dst], hashmix(pool[i_src]))
    words, h = [], INIT_B
    for i in range(8):
        v = pool[i & 3] ^ h
        h = _u32(h * MULT_B)
        v = _u32(v * h)
        words.append(v ^ (v >> 16))
    st = [words[2 * i] | (words[2 * i + 1] << 32) for i in range(4)]
    initstate, initseq = (st[0] << 64) | st[1], (st[2] << 64) | st[3]
    inc = ((initseq << 1) | 1) & M128
    state = (0 * PCG_MULT + inc) & M128
    state = (state + initstate) & M128
    state = (state * PCG_MULT + inc) & M128
    return state, inc


def jump(state: int, inc: int, delta: int) -> int:
    """state after `delta` LCG steps (pcg_advance_lcg_128)."""
    acc_m, acc_p, cur_m, cur_p = 1, 0, PCG_MULT, inc
    while delta > 0:
        if delta & 1:
            acc_m = (acc_m * cur_m) & M128
            acc_p = (acc_p * cur_m + cur_p) & M128
        cur_p = ((cur_m + 1) * cur_p) & M128
        cur_m = (cur_m * cur_m) & M128
        delta >>= 1
    return (acc_m * state + acc_p) & M128


def output(state: int) -> int:
    hi, lo = state >> 64, state & 0xFFFFFFFFFFFFFFFF
    x, rot = hi ^ lo, hi >> 58
    return ((x >> rot) | (x << ((64 - rot) & 63))) & 0xFFFFFFFFFFFFFFFF


def u64_at(state, inc, q):
    return output(jump(state, inc, q + 1))


def u32_at(state, inc, pos):
    u = u64_at(state, inc, pos >> 1)
    return (u >> 32) if (pos & 1) else (u & 0xFFFFFFFF)


def bounded_uint32(state, inc, n, range_excl):
    """rng.choice(range_excl, n) for range_excl < 2**32: (values, number of u32 draws consumed)."""
    threshold = ((1 << 32) - range_excl) % range_excl
    out, pos = [], 0
    for _ in range(n):
        while True:
            m = u32_at(state, inc, pos) * range_excl
            pos += 1
            if (m & 0xFFFFFFFF) >= threshold:
                break
        out.append(m >> 32)
    return np.array(out, dtype=np.int64), pos


def doubles(state, inc, first_u64, n):
    return np.array([(u64_at(state, inc, first_u64 + i) >> 11) * (1.0 / 9007199254740992.0) for i in range(n)])


def black_body_draws(seed, n, max_seed_val=2**32 - 1):
    """(packet_seeds, xis[5, n], z[n]) of create_packets, addressed by jump-ahead."""
    state, inc = seed_state(seed)
    seeds, consumed = bounded_uint32(state, inc, n, max_seed_val)
    q0 = (consumed + 1) // 2
    xis = doubles(state, inc, q0, 5 * n).reshape(5, n)
    z = doubles(state, inc, q0 + 5 * n, n)
    return seeds, xis, z
