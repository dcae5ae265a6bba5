"""One-off parity run at a size the CI cannot afford: every peak cell, every counter and the pose of a
2000-point model against a 20000-point scene (5000 reference points, 10^8 scene PPFs), GPU vs the CPU
oracle (about a minute of 16 host threads).  usage: python tools/validate_midsize.py [M S df [tau_d]]"""
import importlib, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
ppf, synth = pkg.ppf, pkg.synth
from oracle import oracle as O
M, S, df = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (2000, 20000, 4)
tau = float(sys.argv[4]) if len(sys.argv) > 4 else 0.04
mp, mn = synth.make_model(0, M); d = synth.d_dist_for(mp, tau)
sp, sn, poses = synth.make_scene([0], S, 2051, instance_points=M, noise_sigma=0.1 * d)
out = {"model_points": M, "scene_points": S, "ref_point_df": df, "tau_d": tau}
for name, gpu_min in (("host tail", 1000000000), ("device tail", 2)):
    mo = ppf.Model(mp, mn, d_dist=d, params=ppf.default_params(pose_gpu_min=gpu_min)); sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df)
    T = mo.ppf_lookup(sc); cells, gposes = mo.last_cells(); st = dict(mo.stats)
    if "oracle" not in out:
        t = time.time(); ocells, ost = O.votes_fused(mp, mn, sp, sn, df, d, 0.4, threads=16); out["oracle_seconds"] = time.time() - t
        _, To = O.pose_from_cells(ocells, mp, mn, sp, sn, d)
        oposes = O.trans_calc2(ocells, mp, mn, sp, sn)
        out["oracle"] = {k: int(ost[k]) for k in ("num_scene_ppfs", "num_hits", "num_votes", "num_unique_votes", "num_top", "max_count")}
    out[name] = {"cells_equal": bool(len(cells) == len(ocells) and np.array_equal(cells["code"], ocells["code"]) and np.array_equal(cells["count"], ocells["count"])),
                 "counters_equal": all(int(st[k]) == int(ost[k]) for k in out["oracle"]),
                 "all_pose_matrices_equal": bool(np.array_equal(gposes, oposes)), "returned_pose_equal": bool(np.array_equal(T, To)),
                 "kept_cells": int(len(cells))}
print(json.dumps(out))
