#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/.

The reference ships no golden vectors and cannot be built in this image (see
oracle/oracle_ppf.h), so two kinds of fixture exist:
  survey_known_answers.json -- the outputs of the reference's own kernels recorded in
      SURVEY.md section 8c (FNV-1a vector; discretised PPFs and keys of a 3-point cloud).
      These pin the oracle.
  case_*.npz -- clouds plus the ORACLE's outputs for them (keys, peak cells, poses).  They
      pin nothing about the reference; they catch drift of the oracle and of the HIP path
      across machines, compilers and libm builds.
Run from the repo root:  python tests/golden/make_golden.py
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

synth = importlib.import_module("objective-slam_amd.synth")
HERE = os.path.dirname(os.path.abspath(__file__))

SURVEY = {
    "source": "SURVEY.md section 8c, 'Known-answer values recorded from that build'",
    "fnv1a": {"bytes_hex": "8001ff7f", "hash": "0x83c01da0"},
    "d_angle0": 0.20943952,
    "cloud": {"points": [[0, 0, 0], [1, 0.5, 0.25], [-0.3, 0.8, 0.1]],
              "normals": [[0, 0, 1], [0, 1, 0], [0.6, 0, 0.8]], "d_dist": 0.05, "df": 1},
    "disc_ppf": {"0,1": [1.10000002, 1.2566371, 1.04719758, 1.46607661],
                 "0,2": [0.850000024, 1.2566371, 1.67551613, 0.628318548]},
    "keys": {"0,1": "0x84b25934", "0,2": "0x2d904f82", "1,0": "0x523738cb", "1,2": "0x2a9acb13",
             "2,0": "0x2d904f82", "2,1": "0x8b81535b", "0,0": "0x0", "1,1": "0x0", "2,2": "0x0"},
}


def make_case(name, M, S, seed, df, tau_d=0.05, model_id=0):
    mp, mn = synth.make_model(model_id, M)
    d = synth.d_dist_for(mp, tau_d)
    sp, sn, poses = synth.make_scene([model_id], S, seed, instance_points=min(M, S // 2))
    _, mkeys = O.ppf_all_pairs(mp, mn, 1, d, want_ppf=False)
    skeys0 = O.ppf_row_keys(sp, sn, 0, d)
    cells, st = O.votes_fused(mp, mn, sp, sn, df, d, 0.4)
    poses_all = O.trans_calc2(cells, mp, mn, sp, sn)
    out = {}
    for tag, kw in (("gpu", {}), ("cpu", {"cpu_clustering": True})):
        rc, T = O.pose_from_cells(cells, mp, mn, sp, sn, d, **kw)
        out["T_" + tag] = T
    np.savez_compressed(os.path.join(HERE, name + ".npz"), mp=mp, mn=mn, sp=sp, sn=sn, d_dist=np.float32(d),
                        df=np.int32(df), model_keys=mkeys, scene_keys_row0=skeys0, cell_code=cells["code"],
                        cell_count=cells["count"], poses=poses_all, truth=poses[0][1].astype(np.float64),
                        stats=np.array([st[k] for k in ("num_scene_ppfs", "num_hits", "num_votes",
                                                       "num_unique_votes", "num_model_keys", "max_count",
                                                       "num_top")], np.uint64), **out)
    print(name, "cells", len(cells), "stats", st)


def make_fullsize_case(name="case_5k_100k_df8", M=5000, S=100000, df=8, tau_d=0.025, seed=2002, threads=0, chunks=50):
    """The bench workload itself (bench.py: model 0 at 5000 points, scene seed 2002 at 100000 points with one
    instance, noise 0.1 d_dist, ref_point_df 8): the oracle's whole registration -- 12 500 reference points,
    1.9e11 votes, about 10 minutes of 16 host threads.  Run in interleaved chunks of reference points (so that a
    progress line appears every few seconds and every chunk's maximum is close to the global one); the chunks' peak
    cells are supersets of what survives the global threshold (model.cu:164-170), so their union filtered with the
    global maximum and ordered (count desc, code asc) is the single run's list.
        python tests/golden/make_golden.py fullsize [threads]"""
    import hashlib
    import time
    mp, mn = synth.make_model(0, M)
    d = synth.d_dist_for(mp, tau_d)
    sp, sn, poses = synth.make_scene([0], S, seed, instance_points=M, noise_sigma=0.1 * d)
    fm = O.FusedModel(mp, mn, d)
    parts, tot = [], {k: 0 for k in ("num_scene_ppfs", "num_hits", "num_votes", "num_unique_votes")}
    gmax, t0 = 0, time.time()
    for c in range(chunks):
        cells, st = fm.votes(sp, sn, df, thresh=0.4, ref_begin=c, ref_step=chunks, threads=threads)
        parts.append(cells)
        for k in tot:
            tot[k] += st[k]
        gmax = max(gmax, st["max_count"])
        print("chunk %d/%d: %d cells, max %d, %.0f s" % (c + 1, chunks, len(cells), st["max_count"], time.time() - t0), flush=True)
    num_model_keys = st["num_model_keys"]
    fm.close()
    allc = np.concatenate(parts)
    keep = allc[allc["count"].astype(np.float32) > np.float32(0.4) * np.float32(gmax)]      # model.cu:164-167, in float
    keep = keep[np.lexsort((keep["code"], -keep["count"].astype(np.int64)))]
    out = {}
    for tag, kw in (("gpu", {}), ("cpu", {"cpu_clustering": True}), ("avg", {"use_averaged_clusters": True})):
        rc, T = O.pose_from_cells(keep, mp, mn, sp, sn, d, **kw)
        out["T_" + tag] = T
    poses_all = O.trans_calc2(keep, mp, mn, sp, sn)
    h = hashlib.sha256()
    for a in (mp, mn, sp, sn):
        h.update(np.ascontiguousarray(a, np.float32).tobytes())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), clouds_sha256=np.array(h.hexdigest()), d_dist=np.float32(d),
                        df=np.int32(df), M=np.int32(M), S=np.int32(S), tau_d=np.float32(tau_d), seed=np.int32(seed),
                        cell_code=keep["code"], cell_count=keep["count"],
                        poses_sha256=np.array(hashlib.sha256(poses_all.tobytes()).hexdigest()),
                        truth=poses[0][1].astype(np.float64),
                        stats=np.array([tot["num_scene_ppfs"], tot["num_hits"], tot["num_votes"], tot["num_unique_votes"],
                                        num_model_keys, gmax, len(keep)], np.uint64),
                        oracle_seconds=np.float32(time.time() - t0), **out)
    print(name, "cells", len(keep), "gmax", gmax, "stats", tot)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "fullsize":
        make_fullsize_case(threads=int(sys.argv[2]) if len(sys.argv) > 2 else 0)
        sys.exit(0)
    with open(os.path.join(HERE, "survey_known_answers.json"), "w") as f:
        json.dump(SURVEY, f, indent=1)
    make_case("case_m64_s128", 64, 128, 3001, 1)
    make_case("case_m200_s400_df3", 200, 400, 3002, 3)
