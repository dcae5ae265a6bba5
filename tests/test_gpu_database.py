"""GPU: BASELINE.json configs[3] and configs[4] at their DATABASE size -- 100 models against one rank's eighth
of a 500k-point scene, and 640x480 depth frames against a 50-model database (all 50 on one GPU, and the 7 models
one of 8 GPUs holds when the database is split by model, with the frame rate asserted).  The reference loops
scenes x models and rebuilds both every time (src/cuda/ppf.cu:57-100); here the tables stay resident
(oslam_db).  The oracle cannot run whole registrations at these sizes: sampled accumulators against it, plus
what the domain offers (every model registered, instances found at the reference's own acceptance test,
alignment.cpp:141-144)."""
import json
import time

import numpy as np
import pytest

from test_gpu_configs import ACCEPT_ROT, _sampled_accumulators

pytestmark = pytest.mark.gpu


def test_config3_hundred_model_database_one_rank_of_eight(ppf, oracle, built_lib, synth):
    """configs[3]: a database of 100 5k-point models, all tables resident (36 GB), against rank 3's eighth of the
    reference points of a 500k-point scene (ref_point_df 20): every model is registered in one oslam_db_align, the
    hit-list pool stays bounded, sampled accumulators of two models equal the oracle's, and the models that are in the
    scene are found once the peaks of all 8 shards are put together."""
    n_models, world, rank, df = 100, 8, 3, 20
    clouds = [synth.make_model(k, 5000) for k in range(n_models)]
    dd = [synth.d_dist_for(c[0], 0.025) for c in clouds]
    present = [0, 17, 42, 99, 0]
    sp, sn, poses = synth.make_scene(present, 500000, 2063, instance_points=25000, noise_sigma=0.1 * dd[0])
    ps = ppf.default_params(shard_rank=rank, shard_world=world)
    sc_r = ppf.Scene(sp, sn, d_dist=0.0, ref_point_downsample_factor=df, params=ps)
    n_all = (len(sp) + df - 1) // df
    n_mine = len(range(rank, n_all, world))
    t0 = time.perf_counter()
    models = [ppf.Model(c[0], c[1], d_dist=d, params=ps) for c, d in zip(clouds, dd)]
    t_build = time.perf_counter() - t0
    resident = sum(m.table_bytes() for m in models)
    assert resident > 30e9                                               # the whole database is in HBM
    models[0].ppf_lookup(sc_r, allow_no_votes=True)                      # maps the hit-list pool: not part of a frame
    db = ppf.Database(models)
    t0 = time.perf_counter()
    T, stats = db.align(sc_r)
    frame_s = time.perf_counter() - t0
    assert T.shape == (n_models, 4, 4)
    for j, st in enumerate(stats):
        assert st["num_scene_ppfs"] == n_mine * (len(sp) - 1), j
        assert st["num_votes"] > 0 and st["max_count"] > 0, j            # every model was registered
        assert st["scratch_bytes"] <= (5 << 30), j                       # batches inside the 4 GiB pool
    db.close()
    # one rank sees an eighth of the reference points -- about 30 on each instance: its own peaks do not decide a pose.
    # The models that are in the scene again with all 8 shards (the exchange through host buffers: local peaks,
    # global maximum, union, pose tail): found at the reference's acceptance test
    found = {}
    scenes = [ppf.Scene(sp, sn, d_dist=0.0, ref_point_downsample_factor=df, params=ppf.default_params(shard_rank=r, shard_world=world))
              for r in range(world)]
    for mid in sorted(set(present)):
        parts, lmaxes = [], []
        for r in range(world):
            loc, lmax = models[mid].align_local(scenes[r], cap=1 << 20)
            parts.append(loc)
            lmaxes.append(lmax)
        gmax = max(lmaxes)
        bound = np.float32(0.4) * np.float32(gmax)
        union = np.concatenate([q[q["count"].astype(np.float32) > bound] for q in parts])
        Tm = models[mid].align_finish(scenes[0], union, gmax)
        for pid, Tt in poses:
            if pid == mid:
                dt, dr = ppf.ht_dist(Tm, Tt)
                found[mid] = found.get(mid, False) or bool(dr < ACCEPT_ROT and dt < 0.1 * synth.bbox_extent(clouds[mid][0]))
    for sc in scenes:
        sc.close()
    assert sum(found.values()) >= 2, found
    print(json.dumps({"config": "cfg3 at database size: 100 models vs one rank's eighth of a 500k scene", "frame_s": frame_s,
                      "seconds_per_model": frame_s / n_models, "build_all_s": t_build, "db_bytes_in_hbm": resident,
                      "votes": int(sum(s["num_votes"] for s in stats)), "wide_workgroups": int(sum(s["wide_workgroups"] for s in stats)),
                      "vote_launches_per_model": stats[0]["vote_launches"]}))
    assert frame_s < 20.0                                                # 8.2 s in round 2's projection
    # two models of the database, one present and one absent, against the oracle on reference points of this rank
    for mid in (17, 64):
        mp, mn = clouds[mid]
        sc_all = ppf.Scene(sp, sn, d_dist=dd[mid], ref_point_downsample_factor=df)
        _sampled_accumulators(ppf, oracle, models[mid], sc_all, mp, mn, sp, sn, dd[mid], df, (rank, rank + world * 555))
        sc_all.close()
    for m in models:
        m.close()


def test_config4_depth_stream_against_fifty_model_database(ppf, oracle, built_lib, synth):
    """configs[4]: 640x480 depth frames -> points + normals -> voxel grid -> scene -> a 50-model database with one
    d_dist (one voxel grid for the stream, one scene pass per frame for all models).  (a) All 50 models on one GPU:
    every frame gives 50 poses, equal to the single-model registrations for sampled models.  (b) The database split
    by model over 8 GPUs: the 7 models of rank 0 (j % 8 == 0) against every frame at 30 frames per second or more,
    depth image to poses; the moving object is found."""
    n_models, world = 50, 8
    raw = [synth.make_model(k, 1500) for k in range(n_models)]
    d = synth.d_dist_for(raw[0][0], 0.05)                                # the stream's leaf size and every model's d_dist
    grids = [ppf.voxel_grid(c[0], c[1], leaf=d) for c in raw]            # alignment.cpp:282-283
    dense, _ = synth.make_model(0, 300000)
    diam = synth.bbox_extent(grids[0][0])
    rng = synth.SplitMix64(93)
    frames, truths = [], []
    for f in range(10):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = synth.random_rotation(rng)
        T[:3, 3] = [0.5 * np.cos(0.7 * f), 0.3 * np.sin(0.7 * f), 5.5 + 0.1 * f]
        frames.append(synth.render_depth(dense @ T[:3, :3].T + T[:3, 3], background_z=9.0, splat=1))
        truths.append(T)

    def scene_of(img):
        return ppf.Scene.from_depth(img, 525.0, 525.0, 319.5, 239.5, leaf=d, d_dist=0.0, ref_point_downsample_factor=4,
                                    z_min=0.5, z_max=12.0, max_jump=0.08)

    # (a) all 50 on one GPU
    models = [ppf.Model(g[0], g[1], d_dist=d) for g in grids]
    db = ppf.Database(models)
    assert db.n_groups == 1
    t_all, wide, found_all = [], 0, 0
    for f in range(3):
        t0 = time.perf_counter()
        sc = scene_of(frames[f])
        T, stats = db.align(sc)
        t_all.append(time.perf_counter() - t0)
        wide += sum(s["wide_workgroups"] for s in stats)
        dt, dr = ppf.ht_dist(T[0], truths[f])
        found_all += int(dr < ACCEPT_ROT and dt < 0.1 * diam)            # bin-accurate poses: 12 degrees is one alpha bin
        if f == 2:
            for j in (0, 13, 49):                                         # the database's pose = the model's own registration
                one = ppf.Model(grids[j][0], grids[j][1], d_dist=d)
                assert np.array_equal(one.ppf_lookup(sc, allow_no_votes=True), T[j]), j
                one.close()
        sc.close()
    assert found_all >= 2, found_all
    db.close()
    for m in models:
        m.close()
    # (b) one GPU's share of the database split by model: 7 models, the frame rate of the whole loop
    mine = list(range(0, n_models, world))
    assert len(mine) == 7
    models = [ppf.Model(grids[j][0], grids[j][1], d_dist=d) for j in mine]
    db = ppf.Database(models)
    scene_of(frames[0]).close()
    db.align(scene_of(frames[1]))                                         # warm: pool and pose-tail work space
    found, wide7 = 0, []
    t0 = time.perf_counter()
    for f, img in enumerate(frames):
        sc = scene_of(img)
        T, stats = db.align(sc)
        sc.close()
        wide7.append(int(sum(s["wide_workgroups"] for s in stats)))
        dt, dr = ppf.ht_dist(T[0], truths[f])
        found += int(dr < ACCEPT_ROT and dt < 0.1 * diam)
    fps = len(frames) / (time.perf_counter() - t0)
    print(json.dumps({"config": "cfg4/5 at database size: depth stream vs 50 models", "all_50_on_one_gpu_frames_per_s": 1.0 / float(np.median(t_all)),
                      "shard_of_7_frames_per_s": fps, "wide_workgroups_all_50": int(wide), "wide_workgroups_per_frame_shard": wide7,
                      "object_found_frames": "%d of %d" % (found, len(frames))}))
    assert found >= len(frames) - 2, found
    assert fps >= 30.0, fps                                               # the real-time loop closes on one GPU's share
    db.close()
    for m in models:
        m.close()
