"""Scratch (CPU): LDS array cycles of the vote instruction (ds_add_u32, 2 groups of 32 lanes, bank = word mod 32, one
cycle per distinct address on the busiest bank of a group) for the bucket sample of tools/bank_dump.py, under the
stored order and under candidate orders.  tools/bank_sim.py dump.npz [n_samples]"""
import sys
import numpy as np

STRIDE = 31


def cycles(words, s, stride=STRIDE):
    """array cycles of every vote instruction of one hit at angle s (bins) over the bucket `words` (stored order)"""
    n = len(words)
    pad = (-n) % 256
    w = np.concatenate([words, np.zeros(pad, np.uint32)])
    valid = np.arange(n + pad) < n
    u = (w >> 11).astype(np.float64) * 30.0 / 2 ** 21
    row = (w & 0x3FF).astype(np.int64)
    b = np.floor((s - u) % 30.0).astype(np.int64)
    addr = row * stride + b
    addr[~valid] = -1
    # position p = 256*chunk + 4*lane + j -> group (chunk, j, half), member lane % 32
    a = addr.reshape(-1, 2, 32, 4).transpose(0, 3, 1, 2).reshape(-1, 32)       # (groups, 32)
    a = np.sort(a, axis=1)
    first = np.ones_like(a, bool)
    first[:, 1:] = a[:, 1:] != a[:, :-1]
    first &= a >= 0
    bank = a % 32
    cnt = np.zeros((a.shape[0], 32), np.int64)
    g = np.repeat(np.arange(a.shape[0]), 32).reshape(a.shape)
    np.add.at(cnt, (g[first], bank[first]), 1)
    mx = cnt.max(axis=1)                                 # cycles per group
    lanes = (a >= 0).sum(axis=1)
    per_instr = mx.reshape(-1, 2).sum(axis=1)            # the two halves of an instruction
    live = lanes.reshape(-1, 2).sum(axis=1) > 0
    return per_instr[live]


def spread_order(words, keyf, seg=4096):
    """the dealing of k_bucket_spread with sort key keyf(words)"""
    out = words.copy()
    for s0 in range(0, len(words), seg):
        w = words[s0:s0 + seg]
        n = len(w)
        if n < 2:
            continue
        srt = w[np.argsort(keyf(w), kind="stable")]
        # dealing order of positions: (lane & 31) first, then chunk, then (j, half)
        p = np.arange(n)
        c, q = p >> 8, p & 255
        lane, j = q >> 2, q & 3
        h, g = lane & 31, lane >> 5
        order = np.lexsort((g, j, c, h))                 # primary h, then c, then j, then g
        out[s0 + order] = srt
    return out


def key_kappa(w, stride=STRIDE):
    u = ((w >> 11).astype(np.uint64) * 30) >> 5
    row = (w & 0x3FF).astype(np.uint64)
    return (u - ((row * stride) << 16)) & ((32 << 16) - 1)


if __name__ == "__main__":
    d = np.load(sys.argv[1])
    words, offs, runs = d["words"], d["offs"], d["runs"]
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    rng = np.random.default_rng(0)
    lens = (offs[1:] - offs[:-1])[runs[:, 0]]
    wgt = runs[:, 1] * lens
    pick = rng.choice(len(runs), ns, p=wgt / wgt.sum())
    res = {}
    for name, fn in (("stored", lambda w: w), ("kappa-spread", lambda w: spread_order(w, key_kappa))):
        tot = []
        for r in pick:
            b = runs[r, 0]
            w = fn(words[offs[b]:offs[b + 1]])
            for s in rng.uniform(0, 30, 2):
                tot.append(cycles(w, s))
        c = np.concatenate(tot)
        res[name] = c
        print("%-14s array cycles/instr %.3f   max(4,.) %.3f   share >4: %.3f  lens median %d" % (
            name, c.mean(), np.maximum(c, 4).mean(), (c > 4).mean(), np.median(lens[pick])))


def spread_wrap_aware(words, seg=4096, stride=STRIDE):
    """candidate: quantile classes by kappa as now, but inside a class the entries go to the groups by their theta_u
    rank, rotated from class to class so that theta_u falls by one bank step per lane of a group: the entries a hit
    angle wraps (bin 0 -> 29) are then the first lanes of every group and meet no unwrapped entry two banks on"""
    out = words.copy()
    for s0 in range(0, len(words), seg):
        w = words[s0:s0 + seg]
        n = len(w)
        if n < 64:
            continue
        G = (n + 31) // 32
        srt = w[np.argsort(key_kappa(w, stride), kind="stable")]
        u = (srt >> 11)
        newsrt = srt.copy()
        for h in range(32):
            lo, hi = h * n // 32, (h + 1) * n // 32          # quantile class h
            cls = srt[lo:hi]
            m = len(cls)
            if m == 0:
                continue
            byu = cls[np.argsort(u[lo:hi], kind="stable")]
            rot = (h * m) // 32
            # position r in the class goes to group (r) ; it should hold u-rank (r - rot) mod m
            idx = (np.arange(m) - rot) % m
            newsrt[lo:hi] = byu[idx]
        # deal: same positions as spread_order
        p = np.arange(n)
        c, q = p >> 8, p & 255
        lane, j = q >> 2, q & 3
        h_, g = lane & 31, lane >> 5
        order = np.lexsort((g, j, c, h_))
        out[s0 + order] = newsrt
    return out


if __name__ == "__main__" and len(sys.argv) > 3:
    d = np.load(sys.argv[1])
    words, offs, runs = d["words"], d["offs"], d["runs"]
    ns = int(sys.argv[2])
    rng = np.random.default_rng(0)
    lens = (offs[1:] - offs[:-1])[runs[:, 0]]
    wgt = runs[:, 1] * lens
    pick = rng.choice(len(runs), ns, p=wgt / wgt.sum())
    for name, fn in (("wrap-aware", spread_wrap_aware),):
        tot = []
        for r in pick:
            b = runs[r, 0]
            w = fn(words[offs[b]:offs[b + 1]])
            for s in rng.uniform(0, 30, 2):
                tot.append(cycles(w, s))
        c = np.concatenate(tot)
        print("%-14s array cycles/instr %.3f   max(4,.) %.3f   share >4: %.3f" % (name, c.mean(), np.maximum(c, 4).mean(), (c > 4).mean()))


def spread_lattice(words, seg=4096, stride=STRIDE, bsel=None):
    """candidate: quantile classes by kappa (32 per segment, G entries each); a class is cut into `a` runs of b
    consecutive kappa ranks, each run ordered by theta_u and rotated from class to class; group (i, r) takes rank r of
    run i: kappa as precise as a run is narrow (1/a of a bank), theta_u falling along the group in steps of 30/b"""
    out = words.copy()
    for s0 in range(0, len(words), seg):
        w = words[s0:s0 + seg]
        n = len(w)
        if n < 64:
            continue
        srt = w[np.argsort(key_kappa(w, stride), kind="stable")]
        u = (srt >> 11)
        newsrt = srt.copy()
        for h in range(32):
            lo, hi = h * n // 32, (h + 1) * n // 32
            m = hi - lo
            if m < 2:
                continue
            b = bsel(m) if bsel else max(1, int(round(np.sqrt(m))))
            for r0 in range(0, m, b):
                r1 = min(m, r0 + b)
                mb = r1 - r0
                blk = srt[lo + r0:lo + r1]
                byu = blk[np.argsort(u[lo + r0:lo + r1], kind="stable")[::-1]]     # theta_u descending
                rot = (h * mb) // 32
                newsrt[lo + r0:lo + r1] = byu[(np.arange(mb) + rot) % mb]
        p = np.arange(n)
        c, q = p >> 8, p & 255
        lane, j = q >> 2, q & 3
        h_, g = lane & 31, lane >> 5
        order = np.lexsort((g, j, c, h_))
        out[s0 + order] = newsrt
    return out
