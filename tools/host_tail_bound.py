"""Scratch: what the pose tail costs when it has to stay on the host (cpu_clustering, use_averaged_clusters:
both are sequential by definition -- greedy clustering in vote order, transformation_clustering.cpp:62-94; the
in-place translation update applied in index order, kernel.cu:747-758) against the default tail on the device.
A scene that does NOT hold the model leaves the most cells above 0.4 * max."""
import importlib, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
ppf, synth = pkg.ppf, pkg.synth
out = []
for S, df, present in ((100000, 8, True), (30000, 4, False), (100000, 8, False)):
    mp, mn = synth.make_model(0, 5000)
    d = synth.d_dist_for(mp, 0.025)
    sp, sn, _ = synth.make_scene([0 if present else 3], S, 2002, instance_points=5000, noise_sigma=0.1 * d)
    sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df)
    for name, flags in (("default (device tail)", {}), ("cpu_clustering", dict(cpu_clustering=True)),
                        ("use_averaged_clusters", dict(use_averaged_clusters=True))):
        mo = ppf.Model(mp, mn, d_dist=d, **flags)
        mo.ppf_lookup(sc, allow_no_votes=True)
        mo.ppf_lookup(sc, allow_no_votes=True)
        st = mo.stats
        out.append(dict(scene_points=S, model_present=present, variant=name, cells=st["num_top"],
                        ms_tail=round(st["ms_total"] - st["ms_vote"], 2), ms_vote=round(st["ms_vote"], 2)))
        print(json.dumps(out[-1]), flush=True)
        mo.close()
