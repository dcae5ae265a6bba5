"""Scratch: how often do the hits of one scene reference point share a key (= a bucket)?"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
ppf, synth = pkg.ppf, pkg.synth
mp, mn = synth.make_model(0, 5000); d = synth.d_dist_for(mp, 0.025)
sp, sn, poses = synth.make_scene([0], 100000, 2002, instance_points=5000, noise_sigma=0.1 * d)
mo = ppf.Model(mp, mn, d_dist=d); sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=8)
allk = np.concatenate([mo.getHashKeys(r) for r in range(0, 5000)])
uk, cnt = np.unique(allk[allk != 0], return_counts=True)
print("model keys", len(uk), "pairs", cnt.sum())
tot_h = tot_d = tot_v = tot_vd = 0
for r in range(0, 100000, 8 * 250):
    k = sc.getHashKeys(r)
    k = k[k != 0]
    pos = np.searchsorted(uk, k); pos[pos >= len(uk)] = 0
    hit = uk[pos] == k
    hk = k[hit]; hv = cnt[pos[hit]]
    dk, first = np.unique(hk, return_index=True)
    tot_h += len(hk); tot_d += len(dk); tot_v += hv.sum(); tot_vd += hv[first].sum()
print("hits", tot_h, "distinct keys among hits", tot_d, "ratio", tot_h / max(tot_d, 1))
print("votes", tot_v, "bucket entries streamed if each distinct key's bucket is read once", tot_vd, "ratio", tot_v / max(tot_vd, 1))
