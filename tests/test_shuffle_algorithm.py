"""The wave-parallel Fisher-Yates of csrc/shuffle_wave.hpp, restated lane by lane in Python and checked against NumPy.

This is a model of the ALGORITHM (it runs on the CPU and calls nothing of the product): the two facts the kernel rests on are
  1. the partner sequence j(n-1) .. j(1) can be produced 64 draws at a time from the raw 32-bit stream (optimistic accept set,
     prefix counts, strike the first failing lane until none fails, cut at mask boundaries);
  2. 64 consecutive swaps can be applied together when no partner lies in the group's own range of i and no two partners are
     equal, and otherwise piece by piece, cut in front of the later swap of every conflicting pair.
The GPU tests compare the kernel itself with the oracle's shuffles (tests/test_gpu_edges.py)."""
import numpy as np
import pytest


def raw_draws32(seed, count):
    """next_uint32 stream of default_rng(seed): low half of each 64-bit output first, then the buffered high half."""
    o = np.random.PCG64(np.random.SeedSequence(seed)).random_raw(count // 2 + 1)
    out = np.empty(2 * len(o), dtype=np.int64)
    out[0::2] = (o & 0xFFFFFFFF).astype(np.int64)
    out[1::2] = (o >> 32).astype(np.int64)
    return out


def partner_stream(n, seed):
    """role C: classify 64 draws per batch; returns j(n-1), j(n-2), ..., j(1)"""
    d = raw_draws32(seed, 4 * n + 4096)
    lanes = np.arange(64)
    i, c, out = n - 1, 0, []
    mask = (1 << int(i).bit_length()) - 1
    lowpow = (mask >> 1) + 1  # steps below this index use the next smaller mask
    while i >= 1:
        v = d[c:c + 64] & mask
        bal = v <= i  # optimistic: accepted if no earlier lane of the batch had been accepted
        while True:   # settle
            rank = np.concatenate([[0], np.cumsum(bal)[:-1]])
            flip = bal & (v > i - rank)
            if not flip.any():
                break
            bal[int(np.argmax(flip))] = False  # its prefix is exact: a true reject
        il = i - rank
        if i - bal.sum() >= lowpow:
            out.extend(v[bal].tolist())
            i -= int(bal.sum())
            c += 64
        else:  # mask boundary or end of chain: only the draws of steps at or above the boundary
            low = il < lowpow
            cut = int(np.argmax(low)) if low.any() else 64
            acc = bal & (lanes < cut)
            out.extend(v[acc].tolist())
            i -= int(acc.sum())
            c += cut
            if 1 <= i < lowpow:
                mask = (1 << int(i).bit_length()) - 1
                lowpow = (mask >> 1) + 1
    return out


def apply_groups(n, js):
    """role A: 64 consecutive steps per group, lane l = step i_top - l"""
    x = list(range(n))
    i_top, k = n - 1, 0
    while i_top >= 1:
        cnt = min(64, i_top)
        v = js[k:k + cnt]
        il = [i_top - l for l in range(cnt)]
        cuts = set()
        for l in range(cnt):  # partner inside the group's own later steps: cut in front of the lane that owns that step
            if i_top - cnt < v[l] < il[l]:
                cuts.add(i_top - v[l])
        seen, dup = {}, set()
        for l in range(cnt):  # equal partners (the kernel finds them with lane-id tags): every member but the first
            if v[l] in seen:
                dup.update((seen[v[l]], l))
            else:
                seen[v[l]] = l
        cuts |= set(sorted(dup)[1:])
        piece = []
        for l in range(cnt + 1):
            if l == cnt or (l in cuts and piece):
                a = [x[il[q]] for q in piece]
                b = [x[v[q]] for q in piece]
                for q, bq in zip(piece, b):
                    x[il[q]] = bq
                for q, aq in zip(piece, a):
                    x[v[q]] = aq
                piece = []
            if l < cnt:
                piece.append(l)
        i_top -= cnt
        k += cnt
    return x


@pytest.mark.parametrize("n", list(range(2, 70)) + [127, 128, 129, 1000, 4097, 10058])
def test_partner_stream_and_grouped_apply_equal_numpy_shuffle(n):
    for seed in (0, 1, 7, 2 ** 40 + 3):
        ref = list(range(n))
        np.random.default_rng(seed).shuffle(ref)
        js = partner_stream(n, seed)
        assert len(js) == n - 1
        assert apply_groups(n, js) == ref, (n, seed)
