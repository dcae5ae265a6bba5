"""Scratch (CPU, oracle keys): what packing several short bucket tails into one vote step would save on the bench
workload.  For sampled reference points: vote iterations now (chunks x hits per (run, slice) item) against
iterations when the last chunks of items that hold <= 128 (<= 64) entries share a step two (four) at a time.
Not part of the product or of a test."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
synth = pkg.synth
from oracle import oracle as O  # noqa: E402
M, S, df, tau, n_sample = 5000, 100000, 8, 0.025, int(sys.argv[1]) if len(sys.argv) > 1 else 24
SLICE = 2046
mp, mn = synth.make_model(0, M)
d = synth.d_dist_for(mp, tau)
sp, sn, poses = synth.make_scene([0], S, 2002, instance_points=M, noise_sigma=0.1 * d)
keys = np.empty((M, M), np.uint32)
for r in range(M):
    keys[r] = O.ppf_row_keys(mp, mn, r, d)
sl = (np.arange(M) // SLICE).astype(np.uint64)
comb = ((sl[:, None] << np.uint64(32)) | keys.astype(np.uint64))[keys != 0]
uk, cnt = np.unique(comb, return_counts=True)
allk = np.unique(uk & np.uint64(0xffffffff))
nsl = (M + SLICE - 1) // SLICE
rng = np.random.default_rng(1)
refs = rng.choice(np.arange(0, S, df), n_sample, replace=False)
T = dict(iters=0, steps=0, votes=0, it_pair=0, st_pair=0, it_quad=0, st_quad=0, it_w=0, short_iters=0, short_steps=0,
         it_pair_window=0, st_pair_window=0)
for r in refs:
    k = O.ppf_row_keys(sp, sn, int(r), d)
    k = k[k != 0]
    pos = np.searchsorted(allk, k.astype(np.uint64)); pos[pos >= len(allk)] = 0
    hk = k[allk[pos] == k.astype(np.uint64)]
    dk, R = np.unique(hk, return_counts=True)
    for s in range(nsl):
        ck = (np.uint64(s) << np.uint64(32)) | dk.astype(np.uint64)
        p = np.searchsorted(uk, ck); p[p >= len(uk)] = 0
        m = uk[p] == ck
        ln, Rm = cnt[p[m]], R[m]
        nch = (ln + 255) // 256
        T["iters"] += int((nch * Rm).sum()); T["steps"] += int((nch * ((Rm + 63) // 64)).sum()); T["votes"] += int((ln * Rm).sum())
        last = ln - (nch - 1) * 256
        short = (last <= 128) & (Rm <= 64)
        Rs = np.sort(Rm[short])[::-1]
        T["short_iters"] += int(Rs.sum()); T["short_steps"] += len(Rs)
        # pairs, sorted by R (best case)
        saved_it = int(Rs[1::2].sum()); saved_st = len(Rs) // 2
        T["it_pair"] += saved_it; T["st_pair"] += saved_st
        # pairs in arrival order inside windows of 64 items (what a window generator could do without sorting)
        Ra = Rm[short]
        sv = 0
        for w0 in range(0, len(Ra), 24):
            w = Ra[w0:w0 + 24]
            sv += int(np.minimum(w[0:len(w) - 1:2], w[1::2]).sum()) if len(w) > 1 else 0
        T["it_pair_window"] += sv; T["st_pair_window"] += len(Ra) // 2
        # quads of <= 64-entry tails
        q = np.sort(Rm[(last <= 64) & (Rm <= 64)])[::-1]
        n4 = len(q) // 4 * 4
        T["it_quad"] += int(q[:n4].reshape(-1, 4)[:, 1:].sum()) if n4 else 0; T["st_quad"] += 3 * (n4 // 4)
n = len(refs)
print({k: v / n for k, v in T.items()})
print("lane use now %.3f" % (T["votes"] / (256.0 * T["iters"])))
print("short (<=128-entry last chunk) items: %.1f%% of iterations, %.1f%% of steps" % (100.0 * T["short_iters"] / T["iters"], 100.0 * T["short_steps"] / T["steps"]))
print("pairs sorted by R: -%.1f%% iterations, -%.1f%% steps" % (100.0 * T["it_pair"] / T["iters"], 100.0 * T["st_pair"] / T["steps"]))
print("pairs in arrival order: -%.1f%% iterations, -%.1f%% steps" % (100.0 * T["it_pair_window"] / T["iters"], 100.0 * T["st_pair_window"] / T["steps"]))
print("quads of <=64 tails sorted by R: -%.1f%% iterations, -%.1f%% steps" % (100.0 * T["it_quad"] / T["iters"], 100.0 * T["st_quad"] / T["steps"]))
