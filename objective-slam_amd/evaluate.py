"""Recall against occlusion: the evaluation protocol of the reference's analyze_mian.py
(pcl/alignment/analyze_mian.py:9-87) -- one registration per (scene, model) pair, the pose error
(|dt|, |rotation angle|) against ground truth by ht_dist, a match when dt <= 0.3 model diameters
and the angle <= 12 degrees (analyze_mian.py:49,75-77), rows sorted by occlusion and the share of
matches among the rows up to each occlusion (PercentMatchBelow, :66-74).

The reference runs it on the UWA scenes (Mian et al.), which are not in its tree and not here, so
`synthetic_rows` generates (scene, model, occlusion) pairs with objective-slam_amd.synth;
`read_occlusion_txt` / `read_alignment_log` read the reference's own file formats for whoever
has the data set and logs of `oslam_alignment` (the CLI prints the two lines the reference's
script looks for, with a two-token prefix as Boost.Log's).
"""
import itertools
import os

import numpy as np

from . import ppf, synth

TWELVEDEG = 0.209440            # analyze_mian.py:49
DIST_THRESH_FACTOR = 0.3        # analyze_mian.py:76


def read_occlusion_txt(path):
    """UWA occlusion.txt: header line, then `scene model occlusion%` rows (analyze_mian.py:9-17)."""
    rows = []
    with open(path) as f:
        for line in f.readlines()[1:]:
            t = line.split()
            if len(t) >= 3:
                rows.append([t[0], t[1], float(t[2])])
    return rows


def read_alignment_log(rows, path, scene_num):
    """Attach (dt, dr) from a log of oslam_alignment (or of the reference's alignment binary) to the
    rows of `scene_num` (analyze_mian.py:19-41: 'Transformations for <model> in <scene>:' then
    'Distance (trans, rot): a, b')."""
    alias = {"cheff": "chef", "T-rex": "trex", "parasaurolophus": "para"}
    cur = None
    with open(path) as f:
        for line in f:
            if "Transformations for" in line:
                cur = os.path.basename(line.split("Transformations for")[1].split()[0]).split("_")[0].split(".")[0]
                cur = alias.get(cur, cur)
            if "Distance" in line and cur is not None:
                vals = line.split("rot):")[1].replace(";", " ").split()
                dist = [float(v.strip(" ,")) for v in vals[:2]]
                for r in rows:
                    if r[0] == scene_num and r[1] == cur and len(r) == 3:
                        r.append(dist)


def match_within_threshold(rows, diameters, dist_thresh_factor=DIST_THRESH_FACTOR, rot_thresh=TWELVEDEG):
    """rows: [scene, model, occlusion, (dt, dr)] -> appends [dt ok, dr ok] (analyze_mian.py:51-56)."""
    for r in rows:
        dr = r[3][1] if r[3][1] <= np.pi else 2 * np.pi - r[3][1]
        r.append([r[3][0] <= dist_thresh_factor * diameters[r[1]], dr <= rot_thresh])


def percent_match_below(rows):
    """Rows sorted by occlusion -> cumulative share of matches (analyze_mian.py:66-74)."""
    m = [1 if all(r[4]) else 0 for r in rows]
    cum = list(itertools.accumulate(m))
    return [c / (i + 1) for i, c in enumerate(cum)]


def synthetic_rows(occlusions, trials_per_level=4, model_points=1500, scene_points=20000, tau_d=0.05,
                   ref_point_df=5, model_ids=(0, 2, 4, 6), seed=9000, params=None, log=None):
    """One registration per (occlusion level, trial): model k = model_ids[trial % len] voxel-gridded at
    leaf = d_dist as the reference does (alignment.cpp:282-283), a cluttered scene holding one
    instance of it with the given share of its surface cut away.  (The odd-numbered synthetic
    surfaces are nearly symmetric under a half turn and are registered flipped about half the
    time even without occlusion, a property of the shapes; the default ids avoid them.)  Returns rows
    [scene id, model name, occlusion %, (dt, dr)] and the model diameters."""
    rows, diam, models = [], {}, {}
    for k in model_ids:
        mp, mn = synth.make_model(k, model_points)
        d = synth.d_dist_for(mp, tau_d)
        mg, mgn = ppf.voxel_grid(mp, mn, leaf=d)
        models[k] = (ppf.Model(mg, mgn, d_dist=d, params=params), d)
        diam["model%d" % k] = synth.bbox_extent(mp)
    sid = 0
    for occ in occlusions:
        for t in range(trials_per_level):
            k = model_ids[t % len(model_ids)]
            mo, d = models[k]
            sp, sn, poses = synth.make_scene([k], scene_points, seed + sid, instance_points=4 * model_points,
                                             noise_sigma=0.05 * d, occlusion=occ)
            sg, sgn = ppf.voxel_grid(sp, sn, leaf=d)
            sc = ppf.Scene(sg, sgn, d_dist=d, ref_point_downsample_factor=ref_point_df, params=params)
            T = mo.ppf_lookup(sc, allow_no_votes=True)
            dt, dr = ppf.ht_dist(T, poses[0][1])
            rows.append(["s%d" % sid, "model%d" % k, 100.0 * occ, (dt, dr)])
            if log:
                log("scene %d model %d occlusion %.0f%%: dt %.3f diam, dr %.1f deg" %
                    (sid, k, 100 * occ, dt / diam["model%d" % k], np.degrees(dr)))
            sc.close()
            sid += 1
    for mo, _ in models.values():
        mo.close()
    return rows, diam


def recall_table(rows, diameters, bins=(0, 60, 70, 80, 85, 90, 101)):
    """Recall per occlusion bin and the reference's cumulative curve."""
    match_within_threshold(rows, diameters)
    rows.sort(key=lambda r: r[2])
    cum = percent_match_below(rows)
    out = []
    for lo, hi in zip(bins[:-1], bins[1:]):
        sel = [r for r in rows if lo <= r[2] < hi]
        if sel:
            out.append({"occlusion": "[%g, %g)" % (lo, hi), "pairs": len(sel),
                        "recall": sum(1 for r in sel if all(r[4])) / len(sel)})
    return out, cum
