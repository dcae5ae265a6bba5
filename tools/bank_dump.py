"""Scratch (GPU box): dumps, for the bench model and scene, the stored words of a sample of buckets together with the
angles of the scene hits that vote on them, for tools/bank_sim.py (LDS bank conflicts of the vote instruction under
different bucket orders, simulated on the CPU).  tools/bank_dump.py out.npz"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
ppf, synth = pkg.ppf, pkg.synth
M, S, df, tau = 5000, 100000, 8, 0.025
mp, mn = synth.make_model(0, M); d = synth.d_dist_for(mp, tau)
sp, sn, poses = synth.make_scene([0], S, 2002, instance_points=M, noise_sigma=0.1 * d)
mo = ppf.Model(mp, mn, d_dist=d)
sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df)
rng = np.random.default_rng(1)
# keys as a few scene reference points see them: the hit keys, with their multiplicity
words, offs, runs = [], [0], []
seen = {}
for r in rng.choice(np.arange(0, S, df), 6, replace=False):
    keys = sc.getHashKeys(int(r))
    uk, cnt = np.unique(keys[keys != 0], return_counts=True)
    for k, c in zip(uk, cnt):
        for sl in range(3):
            if (k, sl) not in seen:
                try:
                    w = mo.bucket_words(int(k), sl)
                except Exception:
                    w = np.zeros(0, np.uint32)
                seen[(k, sl)] = len(offs) - 1 if len(w) else -1
                if len(w):
                    words.append(w); offs.append(offs[-1] + len(w))
            if seen[(k, sl)] >= 0:
                runs.append((seen[(k, sl)], int(c)))
np.savez_compressed(sys.argv[1], words=np.concatenate(words), offs=np.array(offs), runs=np.array(runs))
print("buckets", len(offs) - 1, "words", offs[-1], "runs", len(runs))
