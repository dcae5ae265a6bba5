"""Secondary measurements on one MI355X: BASELINE.json configs[1..4] (bench.py measures the headline
configuration; these are not bench lines).  Needs a HIP device.

  python tools/bench_configs.py cfg2     single 5k model vs 50k voxel-gridded scene
  python tools/bench_configs.py cfg3     10-model database vs 100k scene, all tables resident
  python tools/bench_configs.py cfg4     a 1-GPU slice of the 100-model / 500k-scene job: this rank's
                                         share (1/8 of the reference points) of N models, ref_point_df 20
  python tools/bench_configs.py cfg5     streaming: 640x480 depth frames -> points+normals -> voxel grid
                                         -> registration against a resident model database; frames/s
One JSON line each."""
import importlib, json, os, sys, time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
ppf, synth = pkg.ppf, pkg.synth


def found_at_reference_criterion(T, truth, pts):
    dt, dr = ppf.ht_dist(T, truth)
    return bool(dr < np.radians(12) and dt < 0.1 * synth.bbox_extent(pts))


def align_all(models, dd, sp, sn, df, params=None):
    out = []
    for mo, d in zip(models, dd):
        sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df, params=params)
        out.append((mo.ppf_lookup(sc, allow_no_votes=True).copy(), dict(mo.stats)))
        sc.close()
    return out


def cfg2():
    mp, mn = synth.make_model(0, 5000)
    d = synth.d_dist_for(mp, 0.025)
    raw_p, raw_n, poses = synth.make_scene([0], 200000, 2002, instance_points=20000, noise_sigma=0.1 * d)
    t = time.perf_counter(); sp, sn = ppf.voxel_grid(raw_p, raw_n, leaf=0.6 * d); t_vox = time.perf_counter() - t
    mo = ppf.Model(mp, mn, d_dist=d)
    sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=5)
    mo.ppf_lookup(sc)
    t = time.perf_counter(); T = mo.ppf_lookup(sc); el = time.perf_counter() - t
    st = mo.stats
    return {"config": "cfg2: 5k model vs voxel-gridded scene", "scene_points_after_voxel_grid": len(sp), "voxel_grid_s": t_vox,
            "ref_point_df": 5, "align_s": el, "scene_ppfs_per_s": st["num_scene_ppfs"] / el, "votes_per_s": st["num_votes"] / el,
            "wide_workgroups": st["wide_workgroups"], "found": found_at_reference_criterion(T, poses[0][1], mp)}


def cfg3():
    ids = list(range(10))
    clouds = [synth.make_model(k, 5000) for k in ids]
    dd = [synth.d_dist_for(c[0], 0.025) for c in clouds]
    sp, sn, poses = synth.make_scene([0, 4, 8], 100000, 2003, instance_points=5000, noise_sigma=0.1 * dd[0])
    t = time.perf_counter(); models = [ppf.Model(c[0], c[1], d_dist=d) for c, d in zip(clouds, dd)]; t_build = time.perf_counter() - t
    align_all(models[:1], dd[:1], sp, sn, 10)
    t = time.perf_counter(); res = align_all(models, dd, sp, sn, 10); el = time.perf_counter() - t
    return {"config": "cfg3: 10-model database vs 100k scene", "db_models": 10, "ref_point_df": 10,
            "db_bytes_in_hbm": sum(m.table_bytes() for m in models), "build_all_s": t_build, "frame_s": el,
            "scene_ppfs_per_s": sum(s["num_scene_ppfs"] for _, s in res) / el, "votes_per_s": sum(s["num_votes"] for _, s in res) / el,
            "instances_found": {mid: found_at_reference_criterion(res[mid][0], T, clouds[mid][0]) for mid, T in poses}}


def cfg3db():
    """cfg3 through the database object, once with every model's own d_dist (ten groups of one) and once with ONE
    d_dist for the whole database (one group: one scene pass per frame for all ten models)."""
    ids = list(range(10))
    clouds = [synth.make_model(k, 5000) for k in ids]
    dd = [synth.d_dist_for(c[0], 0.025) for c in clouds]
    sp, sn, poses = synth.make_scene([0, 4, 8], 100000, 2003, instance_points=5000, noise_sigma=0.1 * dd[0])
    sc = ppf.Scene(sp, sn, d_dist=0.0, ref_point_downsample_factor=10)
    out = {"config": "cfg3 through oslam_db: 10-model database vs 100k scene", "ref_point_df": 10}
    for name, dds in (("own_d_dist", dd), ("common_d_dist", [dd[0]] * 10)):
        models = [ppf.Model(c[0], c[1], d_dist=d) for c, d in zip(clouds, dds)]
        db = ppf.Database(models)
        db.align(sc)
        t = time.perf_counter(); T, stats = db.align(sc); el = time.perf_counter() - t
        out[name] = {"groups": db.n_groups, "frame_s": el, "votes_per_s": sum(s["num_votes"] for s in stats) / el,
                     "key_kernels_ms": sum(s["ms_key_kernel"] for s in stats), "vote_kernels_ms": sum(s["ms_vote_kernel"] for s in stats),
                     "instances_found": {mid: found_at_reference_criterion(T[mid], Tt, clouds[mid][0]) for mid, Tt in poses}}
        db.close()
        for m in models:
            m.close()
    return out


def cfg4(n_models=4):
    ids = list(range(n_models))
    clouds = [synth.make_model(k, 5000) for k in ids]
    dd = [synth.d_dist_for(c[0], 0.025) for c in clouds]
    sp, sn, poses = synth.make_scene(ids[:2], 500000, 2004, instance_points=5000, noise_sigma=0.1 * dd[0])
    par = ppf.default_params(shard_rank=0, shard_world=8)          # this GPU's eighth of the reference points
    models = [ppf.Model(c[0], c[1], d_dist=d, params=par) for c, d in zip(clouds, dd)]
    align_all(models[:1], dd[:1], sp, sn, 20, params=par)         # scratch pool sized before the clock starts
    t = time.perf_counter(); res = align_all(models, dd, sp, sn, 20, params=par); el = time.perf_counter() - t
    ppfs = sum(s["num_scene_ppfs"] for _, s in res)
    return {"config": "cfg4 slice: %d of 100 models vs 500k scene, rank 0 of 8 (1/8 of the reference points), local votes + "
                      "host stage on the local peaks only" % n_models, "ref_point_df": 20,
            "ref_points_this_rank": int(res[0][1]["num_scene_ppfs"] // (len(sp) - 1)), "seconds_per_model": el / n_models,
            "projected_seconds_for_100_models_per_gpu": 100 * el / n_models, "scene_ppfs_per_s": ppfs / el,
            "votes_per_s": sum(s["num_votes"] for _, s in res) / el,
            "vote_launches_per_model": res[0][1]["vote_launches"],
            "ms_key_kernels_per_model": sum(s["ms_key_kernel"] for _, s in res) / n_models,
            "ms_vote_kernels_per_model": sum(s["ms_vote_kernel"] for _, s in res) / n_models}


def cfg5(n_models=4, frames=8):
    ids = list(range(0, 2 * n_models, 2))
    clouds = [synth.make_model(k, 1500) for k in ids]
    dd = [synth.d_dist_for(c[0], 0.05) for c in clouds]
    grids = [ppf.voxel_grid(c[0], c[1], leaf=d) for c, d in zip(clouds, dd)]
    models = [ppf.Model(g[0], g[1], d_dist=d) for g, d in zip(grids, dd)]
    dense = [synth.make_model(k, 300000)[0] for k in ids[:2]]
    rng = synth.SplitMix64(55)
    imgs, truths = [], []
    for f in range(frames):                                    # two objects drifting in front of a wall
        pts, tr = [], []
        for j, dn in enumerate(dense):
            T = np.eye(4, dtype=np.float32)
            T[:3, :3] = synth.random_rotation(rng)
            T[:3, 3] = [-2.0 + 4.0 * j + 0.05 * f, 0.3 * j - 0.1, 7.0 + 0.5 * j]
            pts.append(dn @ T[:3, :3].T + T[:3, 3])
            tr.append(T)
        imgs.append(synth.render_depth(np.concatenate(pts), background_z=10.0, splat=1))
        truths.append(tr)
    leaf = min(dd)                                                 # one scene_leaf_size for all models (alignment.cpp:265-271)
    wide = 0
    def one(img):
        t0 = time.perf_counter()
        # depth -> points + normals -> voxel grid -> scene in one call (d_dist 0: a scene for every model)
        sc = ppf.Scene.from_depth(img, 525.0, 525.0, 319.5, 239.5, leaf=leaf, d_dist=0.0, ref_point_downsample_factor=2,
                                  z_min=0.5, z_max=12.0, max_jump=0.08)
        t1 = time.perf_counter()
        nonlocal wide
        out, votes = [], 0
        for mo in models:
            out.append(mo.ppf_lookup(sc, allow_no_votes=True).copy())
            votes += mo.stats["num_votes"]
            wide += mo.stats["wide_workgroups"]
        n_scene = sc.numPoints()
        sc.close()
        t2 = time.perf_counter()
        return out, t1 - t0, 0.0, t2 - t0, 0, n_scene, t2 - t1, votes
    one(imgs[0])
    t = time.perf_counter(); res = [one(im) for im in imgs]; el = time.perf_counter() - t
    ok = sum(found_at_reference_criterion(r[0][j], truths[f][j], clouds[j][0]) for f, r in enumerate(res) for j in range(2))
    return {"config": "cfg5: 640x480 depth frames vs a %d-model database (oslam_scene_from_depth, then one align per model)" % n_models,
            "frames": frames, "frames_per_s": frames / el, "ms_per_frame": 1e3 * el / frames,
            "ms_depth_to_scene": 1e3 * np.mean([r[1] for r in res]),
            "ms_registration_all_models": 1e3 * np.mean([r[6] for r in res]), "votes_per_frame": int(np.mean([r[7] for r in res])),
            "scene_points_after_voxel_grid": int(np.mean([r[5] for r in res])), "wide_workgroups_per_frame": wide / float(frames + 1),
            "objects_found": "%d of %d" % (ok, 2 * frames)}


def _planes_and_object(n, rng, extent, obj_id=0, obj_frac=0.25):
    """A floor (z = 0), a wall (x = 0) and a table top (z = 0.8) of `extent`, with an object standing on the table: the
    kind of cloud a depth camera sees indoors.  Large planes in BOTH clouds put more than 65 535 votes into single
    accumulator cells (every in-plane pair has the same feature), which is what the 32-bit re-vote passes exist for."""
    n_obj = int(n * obj_frac)
    n_pl = (n - n_obj) // 3
    u = rng.uniform(0, extent, (3, n_pl, 2))
    floor = np.concatenate([u[0], np.zeros((n_pl, 1))], 1)
    wall = np.concatenate([np.zeros((n_pl, 1)), u[1]], 1)
    table = np.concatenate([0.3 * extent + 0.4 * u[2], np.full((n_pl, 1), 0.8)], 1)
    op, on = synth.make_model(obj_id, n - 3 * n_pl)
    op = op * (0.25 * extent / synth.bbox_extent(op)) + np.array([0.5 * extent, 0.5 * extent, 0.8 + 0.2 * extent])
    pts = np.concatenate([floor, wall, table, op]).astype(np.float32)
    nrm = np.concatenate([np.tile([0, 0, 1.0], (n_pl, 1)), np.tile([1.0, 0, 0], (n_pl, 1)), np.tile([0, 0, 1.0], (n_pl, 1)), on]).astype(np.float32)
    order = rng.permutation(len(pts))
    return np.ascontiguousarray(pts[order]), np.ascontiguousarray(nrm[order])


def planes(M=5000, S=100000, df=8):
    """The re-vote path where it matters: a plane-dominated model (object on a table in a room corner) against a
    plane-dominated scene of the same room.  Reports wide_workgroups and the time per registration."""
    rng = np.random.default_rng(7)
    mp, mn = _planes_and_object(M, rng, 3.5)
    sp0, sn0 = _planes_and_object(S, rng, 3.5)
    R = synth.random_rotation(synth.SplitMix64(5))
    sp = (sp0 @ R.T + np.float32([0.4, -0.2, 0.1])).astype(np.float32)
    sn = (sn0 @ R.T).astype(np.float32)
    sp += (0.002 * rng.normal(size=sp.shape)).astype(np.float32)
    d = synth.d_dist_for(mp, 0.025)
    mo = ppf.Model(mp, mn, d_dist=d)
    sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df)
    mo.ppf_lookup(sc, allow_no_votes=True)
    t = time.perf_counter(); T = mo.ppf_lookup(sc, allow_no_votes=True); el = time.perf_counter() - t
    st = mo.stats
    Tt = np.eye(4); Tt[:3, :3] = R; Tt[:3, 3] = [0.4, -0.2, 0.1]
    dt, dr = ppf.ht_dist(T, Tt)
    return {"config": "plane-dominated model (%d points) vs plane-dominated scene (%d points): floor + wall + table + object in both" % (M, S),
            "ref_point_df": df, "align_s": el, "ms_vote_kernels": st["ms_vote_kernel"], "ms_key_kernels": st["ms_key_kernel"],
            "votes": st["num_votes"], "max_cell": st["max_count"], "wide_workgroups": st["wide_workgroups"],
            "vote_workgroups": int(st["num_scene_ppfs"] // (S - 1)) * ((M + 2045) // 2046), "votes_per_s": st["num_votes"] / el,
            "rot_err_deg": float(np.degrees(dr)), "trans_err": float(dt)}


def db50(frames=10):
    """configs[4] at database size (tests/test_gpu_database.py measures the same): 640x480 depth frames against 50
    models with one d_dist -- all 50 on one GPU, and the 7 models one of 8 GPUs holds when the database is split by model."""
    n_models, world = 50, 8
    raw = [synth.make_model(k, 1500) for k in range(n_models)]
    d = synth.d_dist_for(raw[0][0], 0.05)
    grids = [ppf.voxel_grid(c[0], c[1], leaf=d) for c in raw]
    dense, _ = synth.make_model(0, 300000)
    rng = synth.SplitMix64(93)
    imgs = []
    for f in range(frames):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = synth.random_rotation(rng)
        T[:3, 3] = [0.5 * np.cos(0.7 * f), 0.3 * np.sin(0.7 * f), 5.5 + 0.1 * f]
        imgs.append(synth.render_depth(dense @ T[:3, :3].T + T[:3, 3], background_z=9.0, splat=1))
    out = {"config": "cfg5 at database size: 640x480 depth frames vs a 50-model database, one d_dist", "frames": frames}
    for name, ids in (("all_50_on_one_gpu", list(range(n_models))), ("shard_of_7_models", list(range(0, n_models, world)))):
        models = [ppf.Model(grids[j][0], grids[j][1], d_dist=d) for j in ids]
        db = ppf.Database(models)
        def frame(img):
            sc = ppf.Scene.from_depth(img, 525.0, 525.0, 319.5, 239.5, leaf=d, d_dist=0.0, ref_point_downsample_factor=4,
                                      z_min=0.5, z_max=12.0, max_jump=0.08)
            T, stats = db.align(sc)
            n = sc.numPoints()
            sc.close()
            return stats, n
        frame(imgs[0])
        t = time.perf_counter(); res = [frame(im) for im in imgs]; el = time.perf_counter() - t
        out[name] = {"frames_per_s": frames / el, "ms_per_frame": 1e3 * el / frames, "scene_points": res[0][1],
                     "ms_key_kernels": float(np.mean([sum(s["ms_key_kernel"] for s in r[0]) for r in res])),
                     "ms_vote_kernels": float(np.mean([sum(s["ms_vote_kernel"] for s in r[0]) for r in res])),
                     "wide_workgroups_per_frame": float(np.mean([sum(s["wide_workgroups"] for s in r[0]) for r in res])),
                     "model_points": int(np.mean([len(grids[j][0]) for j in ids]))}
        db.close()
        for m in models:
            m.close()
    return out


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
    print(json.dumps({"cfg2": cfg2, "cfg3": cfg3, "cfg3db": cfg3db, "cfg4": cfg4, "cfg5": cfg5, "planes": planes, "db50": db50}[which]()), flush=True)
