"""Secondary measurement: BASELINE.json configs[2] -- a 10-model database (5k points each, all tables
resident in HBM) against one 100k-point scene that holds three of them, ref_point_df = 10."""
import importlib, sys, time, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
ppf, synth = pkg.ppf, pkg.synth
ids = list(range(10))
models = [synth.make_model(k, 5000) for k in ids]
dd = [synth.d_dist_for(m[0], 0.025) for m in models]
sp, sn, poses = synth.make_scene([0, 3, 7], 100000, 2003, instance_points=5000, noise_sigma=0.1 * dd[0])
t = time.perf_counter(); M = [ppf.Model(m[0], m[1], d_dist=d) for m, d in zip(models, dd)]; t_build = time.perf_counter() - t
def frame():
    out = []
    for mo, d in zip(M, dd):
        sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=10)
        out.append((mo.ppf_lookup(sc, allow_no_votes=True), dict(mo.stats)))
        sc.close()
    return out
frame()
t = time.perf_counter(); res = frame(); el = time.perf_counter() - t
ppfs = sum(s["num_scene_ppfs"] for _, s in res); votes = sum(s["num_votes"] for _, s in res)
found = {}
for mid, T in poses:
    dt, dr = ppf.ht_dist(res[mid][0], T)
    found[mid] = bool(dr < np.radians(12) and dt < 0.1 * synth.bbox_extent(models[mid][0]))
print(json.dumps({"db_models": 10, "build_all_s": t_build, "frame_s": el, "scene_ppfs_per_s": ppfs / el,
                  "votes_per_s": votes / el, "instances_found_at_reference_criterion": found}))
