"""One-off fuzz: random small registrations, GPU vs oracle (cells, counters, pose), host and device
pose tails alternating.  usage: python tools/fuzz_parity.py [trials] [seed]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
ppf, synth = pkg.ppf, pkg.synth
from oracle import oracle as O
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for t in range(trials):
    kind = rng.integers(0, 4)
    M = int(rng.choice([2, 3, 7, 33, 64, 65, 200, 513, 1024, 1025, 1500, 2100]))
    S = int(rng.choice([2, 5, 63, 64, 65, 300, 1023, 1024, 1025, 2500, 6000]))
    df = int(rng.choice([1, 1, 2, 3, 7, 50]))
    thr = float(rng.choice([0.4, 0.4, 0.1, 0.9]))
    if kind == 0:      # random clouds, non-unit normals
        mp = rng.uniform(-1, 1, (M, 3)).astype(np.float32); mn = (rng.normal(size=(M, 3)) * rng.uniform(0.1, 4, (M, 1))).astype(np.float32)
        sp = rng.uniform(-1.5, 1.5, (S, 3)).astype(np.float32); sn = rng.normal(size=(S, 3)).astype(np.float32)
        d = float(rng.choice([0.05, 0.1, 0.3, 0.7]))
    elif kind == 1:    # synthetic surface + scene
        mp, mn = synth.make_model(int(rng.integers(0, 9)), M); d = synth.d_dist_for(mp, float(rng.choice([0.03, 0.05, 0.1, 0.25])))
        sp, sn, _ = synth.make_scene([0], max(S, 16), int(rng.integers(1, 1 << 20)), instance_points=max(2, min(M, S // 2)), noise_sigma=0.05 * d)
    elif kind == 2:    # planar / duplicated / zero-normal degeneracies
        mp = rng.uniform(-1, 1, (M, 3)).astype(np.float32); mp[:, 2] = 0; mn = np.tile(np.float32([[0, 0, 1]]), (M, 1))
        sp = rng.uniform(-1, 1, (S, 3)).astype(np.float32); sp[: S // 2, 2] = 0.5; sn = np.tile(np.float32([[0, 0, 1]]), (S, 1)); sn[S // 2:] = rng.normal(size=(S - S // 2, 3))
        if M > 4: mp[1] = mp[0]; mn[2] = 0
        d = float(rng.choice([0.1, 0.3]))
        M, S = min(M, 300), min(S, 700)        # planar clouds vote M*S-fold: keep the oracle within seconds
        mp, mn, sp, sn = mp[:M], mn[:M], sp[:S], sn[:S]
    else:              # the model inside the scene, exactly
        mp = rng.uniform(-1, 1, (M, 3)).astype(np.float32); mn = rng.normal(size=(M, 3)).astype(np.float32)
        k = min(M, S)
        sp = np.concatenate([mp[:k], rng.uniform(-3, 3, (S - k, 3))]).astype(np.float32); sn = np.concatenate([mn[:k], rng.normal(size=(S - k, 3))]).astype(np.float32)
        d = float(rng.choice([0.08, 0.2, 0.5]))
    print("trial %d kind %d M %d S %d df %d" % (t, kind, len(mp), len(sp), df), flush=True)
    if os.environ.get("FUZZ_ONLY") and int(os.environ["FUZZ_ONLY"]) != t:
        continue
    tail = int(os.environ.get("FUZZ_TAIL", "2" if t % 2 else "1000000000"))
    ppf.DEFAULT_OVERRIDES["pose_gpu_min"] = tail
    flags = {} if t % 5 else dict(use_l1_norm=True)
    try:
        mo = ppf.Model(mp, mn, d_dist=d, vote_count_threshold=thr, **flags)
        sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df)
        T = mo.ppf_lookup(sc, allow_no_votes=True)
        print("  gpu done", mo.stats, flush=True)
        cells, poses = mo.last_cells()
        oc, ost = O.votes_fused(mp, mn, sp, sn, df, d, thr)
        big = len(oc) > 20000          # the oracle's clustering is quadratic in crowded cells: compare the cells only
        To = T if big else O.pose_from_cells(oc, mp, mn, sp, sn, d, use_l1_norm=bool(flags))[1]
        ok = (len(cells) == len(oc) and np.array_equal(cells["code"], oc["code"]) and np.array_equal(cells["count"], oc["count"])
              and all(int(mo.stats[k]) == int(ost[k]) for k in ("num_hits", "num_votes", "num_unique_votes", "max_count"))
              and np.array_equal(T, To) and (len(oc) < 2 or np.array_equal(poses, O.trans_calc2(oc, mp, mn, sp, sn))))
    except Exception as e:
        ok = False
        print("trial", t, "exception", repr(e))
    if not ok:
        bad += 1
        print("MISMATCH trial %d kind %d M %d S %d df %d thr %.1f d %.3f tail %s" % (t, kind, M, S, df, thr, d, tail), flush=True)
    mo.close(); sc.close()
    if t % 10 == 9:
        print("trial %d done, %d mismatches so far" % (t + 1, bad), flush=True)
print("fuzz: %d trials, %d mismatches" % (trials, bad))
sys.exit(1 if bad else 0)
