"""Scratch: time model build + align at bench sizes, print stats."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
ppf, synth = pkg.ppf, pkg.synth
M = int(sys.argv[1]); S = int(sys.argv[2]); df = int(sys.argv[3]); tau = float(sys.argv[4])
mode = int(sys.argv[5]) if len(sys.argv) > 5 else 0
t = time.time(); mp, mn = synth.make_model(0, M); d = synth.d_dist_for(mp, tau)
sp, sn, poses = synth.make_scene([0], S, 2002, instance_points=M, noise_sigma=0.1 * d); print("gen", time.time() - t, "d_dist", d, flush=True)
par = ppf.default_params(vote_mode=mode)
t = time.time(); mo = ppf.Model(mp, mn, d_dist=d, params=par); print("model build", time.time() - t, flush=True)
t = time.time(); sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df, params=par); print("scene create", time.time() - t, flush=True)
for it in range(3):
    t = time.time(); T = mo.ppf_lookup(sc, allow_no_votes=True); el = time.time() - t
    st = mo.stats
    print("align %.3fs vote %.1fms ppfs/s %.3e votes/s %.3e" % (el, st["ms_vote"], st["num_scene_ppfs"] / el, st["num_votes"] / el), st, flush=True)
print(ppf.ht_dist(T, poses[0][1]))
