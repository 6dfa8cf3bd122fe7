"""CPU walk-through of csrc/sort.hip's three kernels (same tile / wave / round structure, same counters), against
numpy's stable argsort.  A design check for the index arithmetic — the kernels themselves are tested on the GPU."""
import numpy as np

RB, BINS, WAVES, ROUNDS, W = 9, 512, 4, 16, 64
SPAN, TILE = ROUNDS * W, 4 * ROUNDS * W


def digit(k, shift, nb, flip):
    return ((np.uint32(k) ^ np.uint32(flip)) >> np.uint32(shift)) & np.uint32((1 << nb) - 1)


def rank_wave(keys, first, n, shift, nb, flip, cnt):
    rk = np.zeros((ROUNDS, W), np.int64)
    for r in range(ROUNDS):
        idx = first + r * W + np.arange(W)
        valid = idx < n
        d = np.array([digit(keys[i] if v else 0, shift, nb, flip) for i, v in zip(idx, valid)])
        base = np.zeros(W, np.int64)
        for lane in range(W):           # all reads first (one instruction), then the leaders' writes
            if valid[lane]:
                base[lane] = cnt[d[lane]]
        for lane in range(W):
            if not valid[lane]:
                continue
            peers = [l for l in range(W) if valid[l] and d[l] == d[lane]]
            rank = sum(1 for l in peers if l < lane)
            if rank == 0:
                cnt[d[lane]] = base[lane] + len(peers)
            rk[r, lane] = base[lane] + rank
    return rk


def sort_pairs(keys_in, end_bit):
    n = len(keys_in)
    ntiles = -(-n // TILE)
    passes = -(-end_bit // RB)
    flip = 0x80000000 if end_bit == 32 else 0
    src_k, src_v = np.array(keys_in, np.int32), None
    for p in range(passes):
        shift = p * RB
        nb = min(RB, end_bit - shift)
        hist = np.zeros((ntiles, BINS), np.int64)
        for t in range(ntiles):
            cnt = np.zeros((WAVES, BINS), np.int64)
            for w in range(WAVES):
                rank_wave(src_k, t * TILE + w * SPAN, n, shift, nb, flip, cnt[w])
            hist[t] = cnt.sum(0)
        totals = hist.sum(0)
        hist = np.cumsum(hist, 0) - hist          # scan kernel: exclusive over tiles, per digit
        dbase = np.cumsum(totals) - totals
        dst_k, dst_v = np.full(n, -1, np.int32), np.full(n, -1, np.int32)
        for t in range(ntiles):
            cnt = np.zeros((WAVES, BINS), np.int64)
            rks = [rank_wave(src_k, t * TILE + w * SPAN, n, shift, nb, flip, cnt[w]) for w in range(WAVES)]
            run = dbase + hist[t]
            off = np.zeros_like(cnt)
            for w in range(WAVES):
                off[w] = run
                run = run + cnt[w]
            for w in range(WAVES):
                for r in range(ROUNDS):
                    for lane in range(W):
                        i = t * TILE + w * SPAN + r * W + lane
                        if i < n:
                            dst = off[w][digit(src_k[i], shift, nb, flip)] + rks[w][r, lane]
                            assert dst_k[dst] == -1 or n == 0
                            dst_k[dst] = src_k[i]
                            dst_v[dst] = i if src_v is None else src_v[i]
        src_k, src_v = dst_k, dst_v
    return src_k, src_v


rng = np.random.default_rng(0)
for n, hi, eb in [(1, 5, 3), (65, 600, 10), (4097, 2, 1), (5000, 1 << 18, 19), (9000, 1 << 27, 28)]:
    keys = rng.integers(0, hi, n).astype(np.int32)
    k, v = sort_pairs(keys, eb)
    ref = np.argsort(keys, kind="stable")
    assert (v == ref).all() and (k == keys[ref]).all(), (n, hi)
keys = rng.integers(-2 ** 31, 2 ** 31 - 1, 3000).astype(np.int32)
k, v = sort_pairs(keys, 32)
ref = np.argsort(keys, kind="stable")
assert (v == ref).all()
print("sort emulation ok")
