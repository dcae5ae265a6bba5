"""Scratch (CPU, oracle keys): the shape of the vote work of the bench workload -- bucket lengths per
(key, slice), hits / runs / items per scene reference point, chunk occupancy -- for sampled reference
points.  Drives the layout decisions of k_vote; not part of the product or of a test."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
synth = pkg.synth
from oracle import oracle as O  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
df = int(sys.argv[3]) if len(sys.argv) > 3 else 8
tau = float(sys.argv[4]) if len(sys.argv) > 4 else 0.025
n_sample = int(sys.argv[5]) if len(sys.argv) > 5 else 40
SLICE = int(os.environ.get("SLICE", "1024"))

mp, mn = synth.make_model(0, M)
d = synth.d_dist_for(mp, tau)
sp, sn, poses = synth.make_scene([0], S, 2002, instance_points=M, noise_sigma=0.1 * d)

# model table: (slice, key) -> length
keys = np.empty((M, M), np.uint32)
for r in range(M):
    keys[r] = O.ppf_row_keys(mp, mn, r, d)
sl = (np.arange(M) // SLICE).astype(np.uint64)
comb = (sl[:, None] << np.uint64(32)) | keys.astype(np.uint64)
comb = comb[keys != 0]
uk, cnt = np.unique(comb, return_counts=True)
print("model: %d pairs, %d (slice,key) buckets, %d distinct keys" % (cnt.sum(), len(uk), len(np.unique(keys[keys != 0]))))
q = [50, 90, 99, 99.9, 100]
print("bucket length percentiles", dict(zip(q, np.percentile(cnt, q))))
# which distance bins are reachable
dist = np.sqrt(((mp[:, None, :] - mp[None, :, :]) ** 2).sum(-1))
k1max = int(dist.max() / d) + 1
print("model diameter in bins", k1max)

nsl = (M + SLICE - 1) // SLICE
tot = dict(pairs=0, keep=0, hits=0, runs=0, items=0, chunks=0, steps_iters=0, votes=0, lane_slots=0)
hist_last = np.zeros(5, np.float64)     # iterations by entries-per-lane class of the chunk: <=64, <=128, <=192, <256, full
hist_R = np.zeros(65, np.int64)
per_ref = []
rng = np.random.default_rng(1)
refs = rng.choice(np.arange(0, S, df), n_sample, replace=False)
for r in refs:
    k = O.ppf_row_keys(sp, sn, int(r), d)
    dd = np.sqrt(((sp - sp[r]) ** 2).sum(-1))
    keep = (dd / d < k1max + 1)
    keep[r] = False
    tot["pairs"] += S - 1
    tot["keep"] += int(keep.sum())
    k = k[k != 0]
    allk = np.unique(uk & np.uint64(0xffffffff))
    pos = np.searchsorted(allk, k.astype(np.uint64))
    pos[pos >= len(allk)] = 0
    hk = k[allk[pos] == k.astype(np.uint64)]
    tot["hits"] += len(hk)
    dk, R = np.unique(hk, return_counts=True)
    # runs: a key's hits in pieces of <= 64
    ref_items = ref_iters = ref_votes = 0
    for s in range(nsl):
        ck = (np.uint64(s) << np.uint64(32)) | dk.astype(np.uint64)
        p = np.searchsorted(uk, ck)
        p[p >= len(uk)] = 0
        m = uk[p] == ck
        ln = cnt[p[m]]
        Rm = R[m]
        pieces = (Rm + 63) // 64
        ref_items += int(pieces.sum())
        nch = (ln + 255) // 256
        tot["chunks"] += int((nch * pieces).sum())
        iters = nch * Rm
        ref_iters += int(iters.sum())
        ref_votes += int((ln * Rm).sum())
        last = ln - (nch - 1) * 256           # entries of the last chunk (1..256)
        cls = np.minimum((last - 1) // 64, 3)
        for c in range(4):
            sel = (cls == c) & (last < 256)
            hist_last[c] += Rm[sel].sum()
        hist_last[4] += ((nch - 1) * Rm).sum() + Rm[last == 256].sum()
    np.add.at(hist_R, np.minimum(R, 64), 1)
    tot["runs"] += int(((R + 63) // 64).sum())
    tot["items"] += ref_items
    tot["steps_iters"] += ref_iters
    tot["votes"] += ref_votes
    per_ref.append((len(hk), ref_items, ref_iters, ref_votes))
n = len(refs)
print("per reference point (mean of %d): pairs %d, within reach %d (%.1f%%), hits %d (%.2f%%), runs %d, items %d, chunks(steps) %d, "
      "vote iterations %d, votes %d, lane use %.3f"
      % (n, tot["pairs"] / n, tot["keep"] / n, 100 * tot["keep"] / tot["pairs"], tot["hits"] / n, 100 * tot["hits"] / tot["pairs"],
         tot["runs"] / n, tot["items"] / n, tot["chunks"] / n, tot["steps_iters"] / n, tot["votes"] / n,
         tot["votes"] / (256.0 * tot["steps_iters"])))
print("iterations by last-chunk class (<=64, <=128, <=192, <256 entries, full chunks): ", (hist_last / hist_last.sum()).round(3))
pr = np.array(per_ref)
print("hits per ref: min %d median %d max %d; votes per ref: min %.2e median %.2e max %.2e"
      % (pr[:, 0].min(), np.median(pr[:, 0]), pr[:, 0].max(), pr[:, 3].min(), np.median(pr[:, 3]), pr[:, 3].max()))
print("R histogram (1,2,3,4,5-8,9-16,17-63,64+):", hist_R[1], hist_R[2], hist_R[3], hist_R[4], hist_R[5:9].sum(), hist_R[9:17].sum(),
      hist_R[17:64].sum(), hist_R[64])
