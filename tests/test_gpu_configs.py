"""GPU: every configuration BASELINE.json names, at its size, through the C-ABI -- the ones the
round-1 suite only ran reduced or not at all (configs[1], [2], [3], [4]; configs[0] and the metric
configuration 5k x 100k live in test_gpu_parity.py) -- plus the multi-GPU host-buffer exchange with
buffers that are too small, and the recall-vs-occlusion protocol on the device.

At these sizes the oracle (CPU) cannot run a whole registration in test time, so the checks are the
ones the domain offers: dense accumulators of sampled reference points against the oracle's
per-reference counters and largest cells, unions of shards against the single run, one-call chains
against their separate steps, poses against ground truth at the reference's own acceptance test
(12 degrees / 0.1 diameters, alignment.cpp:141-144)."""
import importlib

import numpy as np
import pytest

from conftest import cells_equal

pytestmark = pytest.mark.gpu

ACCEPT_ROT = np.deg2rad(12.0)


def _sampled_accumulators(ppf, oracle, mo, sc, mp, mn, sp, sn, d, df, ref_ordinals):
    """Dense accumulator of a few reference points against the oracle: votes, non-empty cells, maximum,
    the ten largest cells."""
    fm = oracle.FusedModel(mp, mn, d)
    try:
        for k in ref_ordinals:
            ocells, st = fm.votes(sp, sn, df, thresh=0.0, ref_begin=k, ref_limit=1)
            acc = mo.vote_accumulator(sc, df * k)
            assert int(acc.sum()) == st["num_votes"], k
            assert int(np.count_nonzero(acc)) == st["num_unique_votes"], k
            assert int(acc.max()) == st["max_count"], k
            top = ocells[:10]
            got = acc[((top["code"] & 0xFFFFFFFF) >> 6).astype(int), (top["code"] & 63).astype(int)]
            assert np.array_equal(got, top["count"]), k
    finally:
        fm.close()


def test_config1_5k_model_vs_50k_voxel_gridded_scene(ppf, oracle, built_lib, synth):
    """BASELINE.json configs[1]: one 5k-point model against a scene voxel-gridded to about 50k points
    (alignment.cpp:265-288: the scene goes through voxelGridDownsample before the path)."""
    mp, mn = synth.make_model(0, 5000)
    d = synth.d_dist_for(mp, 0.025)
    raw_p, raw_n, poses = synth.make_scene([0], 85000, 2061, instance_points=6000, noise_sigma=0.1 * d)
    sp, sn = ppf.voxel_grid(raw_p, raw_n, leaf=d)
    op, on = oracle.voxel_grid(raw_p, raw_n, d)
    assert 45000 <= len(sp) <= 55000
    assert np.array_equal(sp, op) and np.array_equal(sn, on)          # the voxel grid equals its statement, bit for bit
    df = 5
    sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df)
    mo = ppf.Model(mp, mn, d_dist=d)
    n_ref = (len(sp) + df - 1) // df
    _sampled_accumulators(ppf, oracle, mo, sc, mp, mn, sp, sn, d, df, (0, 33, n_ref // 2, n_ref - 1))
    T1 = mo.ppf_lookup(sc)
    cells1, st1 = mo.last_cells()[0], dict(mo.stats)
    assert st1["num_scene_ppfs"] == n_ref * (len(sp) - 1)
    assert st1["num_hits"] <= st1["num_pairs_probed"] <= st1["num_scene_ppfs"]
    assert st1["vote_launches"] == 1                                  # hit lists sized by demand: one launch
    T2 = mo.ppf_lookup(sc)                                            # idempotence
    assert np.array_equal(T1, T2) and cells_equal(cells1, mo.last_cells()[0])
    rc, To = oracle.pose_from_cells(cells1, mp, mn, sp, sn, d)        # pose tail of these cells = the oracle's
    assert np.array_equal(T1, To)
    # two shards partition the votes; their union is the single run
    parts, gmax, votes = [], 0, 0
    for rank in range(2):
        ps = ppf.default_params(shard_rank=rank, shard_world=2)
        scs = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df, params=ps)
        n_loc, lmax = mo.align_local(scs)
        votes += mo.stats["num_votes"]
        parts.append(mo.local_peaks(lmax))
        assert len(parts[-1]) == n_loc
        gmax = max(gmax, lmax)
    assert (votes, gmax) == (st1["num_votes"], st1["max_count"])
    T3 = mo.align_finish(sc, np.concatenate(parts), gmax)
    assert np.array_equal(T3, T1) and cells_equal(mo.last_cells()[0], cells1)
    dt, dr = ppf.ht_dist(T1, poses[0][1])
    assert dr < ACCEPT_ROT and dt < 0.1 * synth.bbox_extent(mp)


def test_config2_ten_model_database_vs_100k_scene(ppf, oracle, built_lib, synth):
    """BASELINE.json configs[2] at full size: a database of ten 5k-point models (all tables resident in
    HBM) against one 100k-point scene that holds instances of three of them, through ppf_registration."""
    ids = list(range(10))
    models = [synth.make_model(k, 5000) for k in ids]
    dd = [synth.d_dist_for(m[0], 0.025) for m in models]
    present = [0, 4, 8]
    sp, sn, poses = synth.make_scene(present, 100000, 2003, instance_points=5000, noise_sigma=0.1 * dd[0])
    df = 10
    res = ppf.ppf_registration([(sp, sn)], models, dd, ref_point_downsample_factor=df)
    assert res.shape == (1, 10, 4, 4)
    found = 0
    for mid, T in poses:
        dt, dr = ppf.ht_dist(res[0, mid], T)
        found += int(dr < ACCEPT_ROT and dt < 0.1 * synth.bbox_extent(models[mid][0]))
    assert found == len(present), found
    # two of the models again, one present and one absent, as objects: the entry point's pose is theirs,
    # and sampled accumulators equal the oracle's
    sc = ppf.Scene(sp, sn, d_dist=0.0, ref_point_downsample_factor=df)      # one scene for every d_dist
    for mid in (4, 5):
        mp, mn = models[mid]
        mo = ppf.Model(mp, mn, d_dist=dd[mid])
        T = mo.ppf_lookup(sc, allow_no_votes=True)
        assert np.array_equal(T, res[0, mid])
        _sampled_accumulators(ppf, oracle, mo, sc, mp, mn, sp, sn, dd[mid], df, (7, 4321))
        assert mo.table_bytes() > 100e6                                   # 25 M pair entries resident
        mo.close()


def test_config3_one_rank_of_eight_500k_scene(ppf, oracle, built_lib, synth):
    """BASELINE.json configs[3], one GPU's part: a 500k-point scene, ref_point_df 20, reference points
    dealt to 8 ranks, two models of the database.  (a) rank 3's accumulators of sampled reference points
    equal the oracle's; (b) on a subsampled reference set (df 160) the union of the 8 shards' peaks above
    the global threshold equals the single run's, and so does the pose."""
    ids = [0, 1]
    models = [synth.make_model(k, 5000) for k in ids]
    dd = [synth.d_dist_for(m[0], 0.025) for m in models]
    sp, sn, poses = synth.make_scene([0, 1, 0, 1, 0], 500000, 2063, instance_points=5000, noise_sigma=0.1 * dd[0])
    world, rank, df = 8, 3, 20
    ps = ppf.default_params(shard_rank=rank, shard_world=world)
    sc_r = ppf.Scene(sp, sn, d_dist=0.0, ref_point_downsample_factor=df, params=ps)
    n_all = (len(sp) + df - 1) // df
    n_mine = len(range(rank, n_all, world))
    for mid in ids:
        mp, mn = models[mid]
        mo = ppf.Model(mp, mn, d_dist=dd[mid], params=ps)
        n_loc, lmax = mo.align_local(sc_r)
        st = mo.stats
        assert st["num_scene_ppfs"] == n_mine * (len(sp) - 1)
        assert st["scratch_bytes"] <= (5 << 30)                           # no 32 GiB pool, no OSLAM_SCRATCH_GIB
        assert n_loc > 0 and lmax > 0
        if mid == 0:
            # reference ordinals of rank 3: rank + world * t
            sc_all = ppf.Scene(sp, sn, d_dist=dd[mid], ref_point_downsample_factor=df)
            _sampled_accumulators(ppf, oracle, mo, sc_all, mp, mn, sp, sn, dd[mid], df, (rank, rank + world * 777))
            sc_all.close()
        mo.close()
    # (b) 8 shards against the single run
    df2 = 160
    mp, mn = models[0]
    mo = ppf.Model(mp, mn, d_dist=dd[0])
    sc1 = ppf.Scene(sp, sn, d_dist=dd[0], ref_point_downsample_factor=df2)
    T1 = mo.ppf_lookup(sc1)
    cells1, st1 = mo.last_cells()[0], dict(mo.stats)
    parts, lmaxes, votes, hits = [], [], 0, 0
    for r in range(world):
        # the shard lives in the scene (oslam_scene_create); one resident model table serves all of them
        pr = ppf.default_params(shard_rank=r, shard_world=world)
        scr = ppf.Scene(sp, sn, d_dist=dd[0], ref_point_downsample_factor=df2, params=pr)
        loc, lmax = mo.align_local(scr, cap=1 << 20)          # records above the LOCAL threshold: a superset
        votes += mo.stats["num_votes"]
        hits += mo.stats["num_hits"]
        parts.append(loc)
        lmaxes.append(lmax)
        scr.close()
    gmax = max(lmaxes)
    bound = np.float32(0.4) * np.float32(gmax)                 # model.cu:164-167, in float
    union = np.concatenate([p[p["count"].astype(np.float32) > bound] for p in parts])
    assert (votes, hits, gmax) == (st1["num_votes"], st1["num_hits"], st1["max_count"])
    T8 = mo.align_finish(sc1, union, gmax)
    assert np.array_equal(T8, T1) and cells_equal(mo.last_cells()[0], cells1)


def test_config4_depth_stream_against_database(ppf, oracle, built_lib, synth):
    """BASELINE.json configs[4], one GPU's part of the loop: 640x480 depth frames -> points + normals ->
    voxel grid -> scene (Scene.from_depth, one call, the full-resolution cloud staying in HBM) -> match
    against a resident multi-model database.  Eight frames of an object moving in front of a wall: every
    frame's poses equal the chain of separate calls (depth_to_cloud, voxel_grid, Scene) cell for cell, and the
    object that is there is found in most frames."""
    ids = [0, 2, 4, 6]
    models, dds, mos = [], [], []
    for k in ids:
        mp, mn = synth.make_model(k, 1500)
        d = synth.d_dist_for(mp, 0.05)
        mg, mgn = ppf.voxel_grid(mp, mn, leaf=d)                          # alignment.cpp:282-283
        models.append((mg, mgn))
        dds.append(d)
        mos.append(ppf.Model(mg, mgn, d_dist=d))
    leaf = dds[0]
    dense, _ = synth.make_model(0, 300000)
    diam = synth.bbox_extent(models[0][0])
    rng = synth.SplitMix64(91)
    found = 0
    n_frames = 8
    for f in range(n_frames):
        T = np.eye(4, dtype=np.float32)
        T[:3, :3] = synth.random_rotation(rng)
        T[:3, 3] = [0.5 * np.cos(0.7 * f), 0.3 * np.sin(0.7 * f), 5.5 + 0.1 * f]
        img = synth.render_depth(dense @ T[:3, :3].T + T[:3, 3], background_z=9.0, splat=1)
        sc = ppf.Scene.from_depth(img, 525.0, 525.0, 319.5, 239.5, leaf=leaf, d_dist=0.0, ref_point_downsample_factor=4,
                                  z_min=0.5, z_max=12.0, max_jump=0.08)
        # the chain in separate calls
        cp, cn = ppf.depth_to_cloud(img, 525.0, 525.0, 319.5, 239.5, z_min=0.5, z_max=12.0, max_jump=0.08)
        vp, vn = ppf.voxel_grid(cp, cn, leaf=leaf)
        assert sc.numPoints() == len(vp)
        sc2 = ppf.Scene(vp, vn, d_dist=0.0, ref_point_downsample_factor=4)
        for j, mo in enumerate(mos):
            Ta = mo.ppf_lookup(sc, allow_no_votes=True).copy()
            ca = mo.last_cells()[0]
            Tb = mo.ppf_lookup(sc2, allow_no_votes=True)
            assert np.array_equal(Ta, Tb) and cells_equal(ca, mo.last_cells()[0]), (f, j)
            if j == 0:
                dt, dr = ppf.ht_dist(Ta, T)
                found += int(dr < ACCEPT_ROT and dt < 0.1 * diam)
        sc.close()
        sc2.close()
    assert found >= n_frames - 2, found


def test_local_peaks_are_never_cut_silently(ppf, oracle, built_lib, case_small):
    """The multi-GPU exchange through host buffers with buffers that are too small: oslam_align_local says
    so (OSLAM_E_LIMIT and the number needed) instead of dropping records, oslam_local_peaks hands out
    the records above the global threshold, and the union of three emulated shards equals the
    single-GPU cells and pose."""
    c = case_small
    world, df = 3, 2
    mo1 = ppf.Model(c["mp"], c["mn"], d_dist=c["d"])
    sc1 = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=df)
    T1 = mo1.ppf_lookup(sc1)
    cells1 = mo1.last_cells()[0]
    shards, lmaxes = [], []
    for rank in range(world):
        par = ppf.default_params(shard_rank=rank, shard_world=world)
        sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=df, params=par)
        mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], params=par)
        n_loc, lmax = mo.align_local(sc)                 # numbers only
        assert n_loc > 4
        with pytest.raises(ppf.OslamError) as e:         # a buffer for 4 records: refused, loudly
            mo.align_local(sc, cap=4)
        assert e.value.code == ppf.OSLAM_E_LIMIT
        shards.append(mo)
        lmaxes.append(lmax)
    gmax = max(lmaxes)
    import ctypes as C
    n = C.c_size_t(0)
    small = np.zeros(1, ppf.CELL_DTYPE)
    rc = ppf.lib().oslam_local_peaks(shards[0]._h, gmax, small.ctypes.data_as(C.c_void_p), 1, C.byref(n))
    assert rc == ppf.OSLAM_E_LIMIT and n.value > 1       # the number needed comes back
    union = np.concatenate([m.local_peaks(gmax) for m in shards])
    T = mo1.align_finish(sc1, union, gmax)
    assert np.array_equal(T, T1) and cells_equal(mo1.last_cells()[0], cells1)
    ocells, _ = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], df, c["d"], 0.4)
    assert cells_equal(cells1, ocells)


def test_recall_vs_occlusion_protocol_on_device(ppf, built_lib, synth):
    """analyze_mian.py's protocol (evaluate.py) on synthetic occluded scenes, two levels x two trials: the
    unoccluded pairs are matches by the reference's rule (0.3 diameters, 12 degrees), the table and the
    cumulative curve have the reference's shape."""
    ev = importlib.import_module("objective-slam_amd.evaluate")
    rows, diam = ev.synthetic_rows((0.0, 0.85), trials_per_level=2, model_points=1500, scene_points=20000,
                                   model_ids=(0, 2))
    assert len(rows) == 4 and all(len(r) == 4 for r in rows)
    table, cum = ev.recall_table(rows, diam, bins=(0, 50, 101))
    assert [t["pairs"] for t in table] == [2, 2]
    assert table[0]["recall"] == 1.0                      # nothing cut away: both found
    assert len(cum) == 4 and cum[0] == 1.0 and cum[1] == 1.0
    assert table[1]["recall"] <= table[0]["recall"]


@pytest.mark.parametrize("M,S,seed,noise", [(150, 360, 2071, 0.02), (300, 700, 2072, 0.1)])
def test_matlab_argmax_set_on_device(ppf, oracle, built_lib, synth, M, S, seed, noise):
    """SURVEY.md 8 row a23 on the device: the HIP accumulator of every reference point, read the way the MATLAB
    prototype reads its own (voting_scheme.m:83-94: first maximum in column order per reference point, then the
    reference points above 0.9 of the largest), against the double-precision restatement of the prototype
    (oracle/oracle_matlab.c).  Slices without a bin-edge case are identical; where the margin exceeds the
    counted edge cases the argmax is the same; the selected set is the same."""
    mp, mn = synth.make_model(0, M)
    d = synth.d_dist_for(mp, 0.05)
    sp, sn, _ = synth.make_scene([0], S, seed, instance_points=M, noise_sigma=noise * d)
    skip = 5
    R = oracle.matlab_voting_scheme(mp, mn, sp, sn, skip, d)
    sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=skip)
    mo = ppf.Model(mp, mn, d_dist=d)
    maxima, decided, identical = [], 0, 0
    for t, r in enumerate(range(0, S, skip)):
        a = mo.vote_accumulator(sc, r)
        f = a[:, :30].astype(np.int64)
        f[:, 29] += a[:, 30]                                  # voting_scheme.m:74 puts alpha + pi == 2 pi into the last bin
        ev = int(R["edge_votes"][t])
        assert int(np.abs(f - R["acc"][t].astype(np.int64)).sum()) <= 2 * ev, t
        identical += int(ev == 0)
        row, col, mx = oracle.matlab_argmax(f)
        maxima.append(mx)
        top2 = np.sort(f.ravel())[-2:]
        if top2[1] - top2[0] > 2 * ev:
            decided += 1
            assert (row, col, mx) == (R["argmax_row"][t], R["argmax_col"][t], R["max_tots"][t]), t
    assert decided >= 10 and identical >= 1          # slices without any edge case are identical (l1 <= 0 above)
    maxima = np.float64(maxima)
    selected = maxima / maxima.max() > 0.9                    # voting_scheme.m:90-92
    # the selection can only differ where a maximum sits within the edge cases' reach of the 0.9 line
    near = np.abs(maxima - 0.9 * maxima.max()) <= 2 * R["edge_votes"].astype(np.float64) + 1
    assert np.array_equal(selected[~near], R["selected"][~near])
    assert selected.any()


def test_model_database_shares_the_scene_pass(ppf, oracle, built_lib, synth):
    """oslam_db: models that share d_dist vote from ONE scene pass per frame (one union table for the group); the
    poses, peak cells and vote counters of every model equal its own single-model registration, for a group of
    three, a model with a d_dist of its own next to it, and after the database is gone again."""
    ids = [0, 2, 4, 6]
    clouds = [synth.make_model(k, 900) for k in ids]
    d_common = synth.d_dist_for(clouds[0][0], 0.05)
    dd = [d_common, d_common, d_common, synth.d_dist_for(clouds[3][0], 0.04)]
    sp, sn, poses = synth.make_scene(ids, 9000, 2081, instance_points=900, noise_sigma=0.05 * d_common)
    sc = ppf.Scene(sp, sn, d_dist=0.0, ref_point_downsample_factor=3)
    models = [ppf.Model(c[0], c[1], d_dist=d) for c, d in zip(clouds, dd)]
    single = []
    for mo in models:
        T = mo.ppf_lookup(sc, allow_no_votes=True).copy()
        single.append((T, mo.last_cells()[0], mo.stats["num_votes"], mo.stats["max_count"], mo.stats["num_unique_votes"]))
    db = ppf.Database(models)
    assert db.n_groups == 2
    for _ in range(2):                                          # twice: nothing is left over between frames
        Ts, stats = db.align(sc)
        for j, mo in enumerate(models):
            T, cells, votes, mx, nz = single[j]
            assert np.array_equal(Ts[j], T), j
            assert cells_equal(mo.last_cells()[0], cells), j
            assert (stats[j]["num_votes"], stats[j]["max_count"], stats[j]["num_unique_votes"]) == (votes, mx, nz), j
        # the group's pass probes the union of its members' keys: at least every member's own hits
        assert stats[0]["num_hits"] == stats[1]["num_hits"] == stats[2]["num_hits"]
    # one member against the oracle, through the database
    ocells, _ = oracle.votes_fused(clouds[1][0], clouds[1][1], sp, sn, 3, dd[1], 0.4)
    assert cells_equal(models[1].last_cells()[0], ocells)
    db.close()
    for j, mo in enumerate(models):                             # members are whole again
        assert np.array_equal(mo.ppf_lookup(sc, allow_no_votes=True), single[j][0]), j
        assert cells_equal(mo.last_cells()[0], single[j][1])


@pytest.mark.parametrize("df", [4, 8])
def test_model_database_group_of_two_slice_models(ppf, built_lib, synth, df):
    """A group whose members have more than one table slice (2046 model points per slice) and different table layouts:
    every member addresses its own slices' entries, voted member after member (df 4: 3 504 workgroups each) or in one
    grid (df 8: 1 760 each, below the bar of run_votes_group).  Poses, peak cells and counters equal the members' own
    registrations."""
    ids = [1, 3, 5]
    clouds = [synth.make_model(k, n) for k, n in zip(ids, (2300, 2100, 2500))]
    d = synth.d_dist_for(clouds[0][0], 0.05)
    sp, sn, _ = synth.make_scene(ids, 7000, 2093, instance_points=900, noise_sigma=0.05 * d)
    sc = ppf.Scene(sp, sn, d_dist=0.0, ref_point_downsample_factor=df)
    models = [ppf.Model(c[0], c[1], d_dist=d) for c in clouds]
    single = []
    for mo in models:
        T = mo.ppf_lookup(sc, allow_no_votes=True).copy()
        single.append((T, mo.last_cells()[0], mo.stats["num_votes"], mo.stats["max_count"], mo.stats["num_unique_votes"]))
    db = ppf.Database(models)
    assert db.n_groups == 1
    for _ in range(2):
        Ts, stats = db.align(sc)
        for j, mo in enumerate(models):
            T, cells, votes, mx, nz = single[j]
            assert (stats[j]["num_votes"], stats[j]["max_count"], stats[j]["num_unique_votes"]) == (votes, mx, nz), j
            assert cells_equal(mo.last_cells()[0], cells), j
            assert np.array_equal(Ts[j], T), j
    db.close()
    for m in models:
        m.close()


def test_align_multi_on_a_communicator_of_one(ppf, oracle, built_lib, case_small, case_two_slices):
    """oslam_align_multi through RCCL with a world of one rank (all this box has): the all-reduce of the maximum, the
    device-side filter with the global threshold, the all-gather with exact sizes and the pose tail on the union
    give the single-GPU registration -- cells, counters and pose -- with the host tail and with the device tail."""
    comm = ppf.Comm(ppf.Comm.unique_id(), 0, 1, 0)
    try:
        for c, df in ((case_small, 1), (case_two_slices, 10)):
            for tail_min in (2, 1000000):
                mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], params=ppf.default_params(pose_gpu_min=tail_min))
                sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=df)
                T1 = mo.ppf_lookup(sc).copy()
                cells1, st1 = mo.last_cells()[0], dict(mo.stats)
                Tm = mo.align_multi(sc, comm)
                assert np.array_equal(Tm, T1) and cells_equal(mo.last_cells()[0], cells1)
                for k in ("num_votes", "num_hits", "max_count", "num_top"):
                    assert mo.stats[k] == st1[k], k
                ocells, _ = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], df, c["d"], 0.4)
                assert cells_equal(cells1, ocells)
                mo.close()
    finally:
        comm.close()
    # a scene whose shard is not the communicator's rank is refused
    par = ppf.default_params(shard_rank=1, shard_world=2)
    comm = ppf.Comm(ppf.Comm.unique_id(), 0, 1, 0)
    mo = ppf.Model(case_small["mp"], case_small["mn"], d_dist=case_small["d"])
    with pytest.raises(ppf.OslamError):
        mo.align_multi(ppf.Scene(case_small["sp"], case_small["sn"], d_dist=case_small["d"], params=par), comm)
    comm.close()


def _arc_cloud(n, rng, spread=0.15):
    """A point at the origin and n points on a short arc of radius 1 around it, all normals +z: every (origin, arc)
    pair has the same key and nearly the same in-plane angle."""
    a = rng.uniform(-spread, spread, n)
    p = np.zeros((n + 1, 3), np.float32)
    p[1:, 0] = np.cos(a)
    p[1:, 1] = np.sin(a)
    p[1:, 2] = rng.uniform(-1e-3, 1e-3, n)
    return p, np.tile(np.float32([0, 0, 1]), (n + 1, 1))


@pytest.mark.parametrize("filler", [0, 1040])
def test_counters_beyond_16_bits(ppf, oracle, built_lib, filler):
    """The accumulator keeps two 16-bit counters per word; a workgroup whose counters overflow notices (the sum of
    its counters falls short of the votes it cast) and is voted again with 32-bit counters, one half at a time.
    Arc clouds put ~2.8e5 votes into single cells (more than four times what 16 bits hold): the dense accumulator,
    the peak cells, the statistics and the pose still equal the oracle's, and the statistics say that the wide passes
    ran.  With filler points in front the overflowing rows lie in the upper half of a word."""
    rng = np.random.default_rng(5)
    mp, mn = _arc_cloud(159, rng)
    sp, sn = _arc_cloud(2999, rng)
    if filler:
        fp = rng.uniform(5, 6, (filler, 3)).astype(np.float32)
        fn = rng.normal(size=(filler, 3)).astype(np.float32)
        fn /= np.linalg.norm(fn, axis=1, keepdims=True)
        mp, mn = np.concatenate([fp, mp]), np.concatenate([fn, mn])
    d = 0.3
    sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=1500)
    mo = ppf.Model(mp, mn, d_dist=d)
    acc = mo.vote_accumulator(sc, 0)
    assert acc.max() > 3 * 65535
    assert np.array_equal(acc, oracle.accumulator_for_ref(mp, mn, sp, sn, 0, d))
    T = mo.ppf_lookup(sc)
    assert mo.stats["wide_workgroups"] > 0 and mo.stats["max_count"] > 4 * 65535
    ocells, ost = oracle.votes_fused(mp, mn, sp, sn, 1500, d, 0.4)
    assert cells_equal(mo.last_cells()[0], ocells)
    for k in ("num_votes", "num_unique_votes", "max_count", "num_hits"):
        assert mo.stats[k] == ost[k], k
    _, To = oracle.pose_from_cells(ocells, mp, mn, sp, sn, d)
    assert np.array_equal(T, To)


def test_counters_beyond_16_bits_in_a_database_group(ppf, built_lib):
    """The members of a group of small models vote in one grid, and the re-vote with 32-bit counters is queued
    afterwards for the members whose workgroups overflowed: two arc models (cells of 2.8e5 votes) and one ordinary
    model in one group give what each gives alone -- peak cells, counters, pose -- and say that wide passes ran."""
    rng = np.random.default_rng(5)
    mp, mn = _arc_cloud(159, rng)
    sp, sn = _arc_cloud(2999, rng)
    mp2, mn2 = _arc_cloud(149, rng)
    op = rng.uniform(-1, 1, (140, 3)).astype(np.float32)
    on = rng.normal(size=(140, 3)).astype(np.float32)
    on /= np.linalg.norm(on, axis=1, keepdims=True)
    d = 0.3
    sc = ppf.Scene(sp, sn, d_dist=0.0, ref_point_downsample_factor=1500)
    models = [ppf.Model(a, b, d_dist=d) for a, b in ((mp, mn), (op, on), (mp2, mn2))]
    single = []
    for mo in models:
        T = mo.ppf_lookup(sc, allow_no_votes=True).copy()
        single.append((T, mo.last_cells()[0], {k: mo.stats[k] for k in ("num_votes", "num_unique_votes", "max_count", "wide_workgroups")}))
    assert single[0][2]["wide_workgroups"] > 0 and single[0][2]["max_count"] > 4 * 65535
    db = ppf.Database(models)
    assert db.n_groups == 1
    for _ in range(2):
        Ts, stats = db.align(sc)
        for j, mo in enumerate(models):
            T, cells, st = single[j]
            assert {k: stats[j][k] for k in st} == st, j
            assert cells_equal(mo.last_cells()[0], cells), j
            assert np.array_equal(Ts[j], T), j
    db.close()
    for m in models:
        m.close()


def test_distances_on_bin_edges(ppf, oracle, built_lib):
    """The scene-key kernels find a pair's distance bin from the hardware's approximate square root and fall
    back to the exact sequence near a bin edge.  Lattice clouds whose spacing IS d_dist (and a third and 1.5 times
    d_dist) put most pair distances exactly on, one ulp below or one ulp above an edge: hit counts, peak cells and
    pose equal the oracle's, so every such pair went to the reference's bin."""
    rng = np.random.default_rng(11)
    for h, mult in ((np.float32(0.05), 1.0), (np.float32(0.1), 1.0 / 3.0), (np.float32(0.037), 1.5)):
        g = np.stack(np.meshgrid(np.arange(7), np.arange(7), np.arange(5), indexing="ij"), -1).reshape(-1, 3)
        mp = (g[rng.permutation(len(g))[:150]] * h).astype(np.float32)
        mn = rng.normal(size=(len(mp), 3)).astype(np.float32)
        g2 = np.stack(np.meshgrid(np.arange(12), np.arange(12), np.arange(8), indexing="ij"), -1).reshape(-1, 3)
        sp = (g2[rng.permutation(len(g2))[:900]] * h).astype(np.float32)
        sn = rng.normal(size=(len(sp), 3)).astype(np.float32)
        near = {tuple(v): i for i, v in enumerate(np.round(mp / h).astype(int))}
        for j, v in enumerate(np.round(sp / h).astype(int) - 2):       # the model sits in the scene, shifted by 2 cells
            if tuple(v) in near:
                sn[j] = mn[near[tuple(v)]]
        d = float(np.float32(h * np.float32(mult)))
        sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=3)
        mo = ppf.Model(mp, mn, d_dist=d)
        T = mo.ppf_lookup(sc, allow_no_votes=True)
        ocells, ost = oracle.votes_fused(mp, mn, sp, sn, 3, d, 0.4)
        assert cells_equal(mo.last_cells()[0], ocells), float(h)
        for key in ("num_scene_ppfs", "num_hits", "num_votes", "num_unique_votes", "max_count"):
            assert mo.stats[key] == ost[key], (float(h), key)
        if len(ocells):
            assert np.array_equal(T, oracle.pose_from_cells(ocells, mp, mn, sp, sn, d)[1])


@pytest.mark.parametrize("bins", [3000, 40000])
def test_more_distance_bins_than_the_key_map_covers(ppf, oracle, built_lib, bins):
    """The scene-key kernel looks a pair up in the key map for distance bins below 2048 and hashes and probes beyond
    it; beyond 16384 bins the reach bitset ends and every pair is keyed.  A d_dist so small that the clouds span
    3000 (40000) bins exercises both seams: counters, peak cells and pose equal the oracle's."""
    rng = np.random.default_rng(77)
    M, S = 90, 260
    mp = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
    mn = rng.normal(size=(M, 3)).astype(np.float32)
    R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
    sp = rng.uniform(-2, 2, (S, 3)).astype(np.float32)
    sn = rng.normal(size=(S, 3)).astype(np.float32)
    sp[:M] = (mp @ R.T + np.float32([0.3, -0.2, 0.1])).astype(np.float32)
    sn[:M] = (mn @ R.T).astype(np.float32)
    d = float(np.float32(7.0 / bins))
    sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=1)
    mo = ppf.Model(mp, mn, d_dist=d)
    for r in (0, 5, M - 1):
        assert np.array_equal(sc.getHashKeys(r), oracle.ppf_row_keys(sp, sn, r, d))
    T = mo.ppf_lookup(sc, allow_no_votes=True)
    ocells, ost = oracle.votes_fused(mp, mn, sp, sn, 1, d, 0.4)
    assert cells_equal(mo.last_cells()[0], ocells)
    for key in ("num_scene_ppfs", "num_hits", "num_votes", "num_unique_votes", "max_count"):
        assert mo.stats[key] == ost[key], key
    if len(ocells):
        assert np.array_equal(T, oracle.pose_from_cells(ocells, mp, mn, sp, sn, d)[1])
