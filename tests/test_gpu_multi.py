"""GPU: the multi-GPU exchange of oslam_align_multi with MORE THAN ONE RANK on a one-GPU box.

The RCCL path of the library and this test run the same C function (exchange_peaks in oslam_host.c) over a
table of collective operations; here the table is the in-process loopback (oslam_comm_create_loopback): N
emulated ranks share the device, one thread per rank, each with its own model handle, scene shard and
communicator end.  What is covered: N = 2, 3, 8; different survivor counts per rank; ranks without any
survivor; the record buffer growing on some ranks only; the union above the device pose tail's threshold;
a failure injected on one rank at each stage between two collectives (every rank must return, none may
hang, the communicator stays usable); and the database split by model (oslam_db_align_multi).
The reference has no multi-GPU code (src/cuda/ppf.cu:45 picks one device): the expected result of every
test is the single-GPU registration, which other tests pin to the oracle."""
import threading

import numpy as np
import pytest

from conftest import cells_equal

pytestmark = pytest.mark.gpu


def _run_ranks(fn, world):
    """fn(rank) on one thread per rank (ctypes releases the GIL inside the library); returns the results,
    exceptions included."""
    out = [None] * world

    def body(r):
        try:
            out[r] = fn(r)
        except Exception as e:                       # noqa: BLE001 -- handed to the caller
            out[r] = e

    ts = [threading.Thread(target=body, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a rank is still inside the exchange: it hangs"
    return out


def _single(ppf, c, df, **par):
    p = ppf.default_params(**par)
    mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], vote_count_threshold=p.vote_count_threshold, params=p)
    sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=df)
    T = mo.ppf_lookup(sc, allow_no_votes=True).copy()
    cells, st = mo.last_cells()[0], dict(mo.stats)
    mo.close()
    sc.close()
    return T, cells, st


def _shards(ppf, c, df, world, per_rank=None, **par):
    models, scenes = [], []
    for r in range(world):
        kw = dict(par)
        kw.update((per_rank or {}).get(r, {}))
        p = ppf.default_params(shard_rank=r, shard_world=world, **kw)
        models.append(ppf.Model(c["mp"], c["mn"], d_dist=c["d"], vote_count_threshold=p.vote_count_threshold, params=p))
        scenes.append(ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=df, params=p))
    return models, scenes


@pytest.mark.parametrize("world", [2, 3, 8])
def test_align_multi_on_emulated_ranks(ppf, oracle, built_lib, case_small, case_two_slices, world):
    """Every rank returns the single-GPU pose and cells; host tail and device tail; survivors differ per rank."""
    comms = ppf.Comm.loopback(world)
    try:
        for c, df in ((case_small, 1), (case_two_slices, 10)):
            for tail_min in (2, 1000000):
                T1, cells1, st1 = _single(ppf, c, df, pose_gpu_min=tail_min)
                models, scenes = _shards(ppf, c, df, world, pose_gpu_min=tail_min)
                res = _run_ranks(lambda r: models[r].align_multi(scenes[r], comms[r], allow_no_votes=True).copy(), world)
                emitted = set()
                for r in range(world):
                    assert not isinstance(res[r], Exception), res[r]
                    assert np.array_equal(res[r], T1), (world, r)
                    assert cells_equal(models[r].last_cells()[0], cells1), (world, r)
                    assert models[r].stats["num_top"] == st1["num_top"] and models[r].stats["max_count"] == st1["max_count"]
                    emitted.add(models[r].stats["num_emitted"])
                assert len(emitted) == 1                                  # everybody saw the same union
                assert sum(m.stats["num_votes"] for m in models) == st1["num_votes"]
                assert sum(m.stats["num_hits"] for m in models) == st1["num_hits"]
                for m, s in zip(models, scenes):
                    m.close()
                    s.close()
        # the expectation itself against the oracle, once
        T1, cells1, _ = _single(ppf, case_small, 1)
        ocells, _ = oracle.votes_fused(case_small["mp"], case_small["mn"], case_small["sp"], case_small["sn"], 1, case_small["d"], 0.4)
        assert cells_equal(cells1, ocells)
        assert np.array_equal(T1, oracle.pose_from_cells(ocells, case_small["mp"], case_small["mn"], case_small["sp"], case_small["sn"], case_small["d"])[1])
    finally:
        for cm in comms:
            cm.close()


def test_ranks_without_survivors_and_buffers_that_grow_on_some_ranks(ppf, built_lib, case_small):
    """(a) A threshold of 0.97: the cells above it belong to one or two reference points, so most of the 8 ranks
    contribute nothing to the union.  (b) A threshold of 0.1 and a record buffer of 8 cells on ranks 0 and 2 only:
    their vote stage repeats with the exact threshold, and the union does not fit their buffers, which grow while
    the other ranks wait at the extra collective."""
    c, df, world = case_small, 1, 8
    comms = ppf.Comm.loopback(world)
    try:
        for thr, per_rank in ((0.97, None), (0.1, {0: dict(max_cells=8), 2: dict(max_cells=8)})):
            T1, cells1, st1 = _single(ppf, c, df, vote_count_threshold=thr)
            models, scenes = _shards(ppf, c, df, world, per_rank=per_rank, vote_count_threshold=thr)
            res = _run_ranks(lambda r: models[r].align_multi(scenes[r], comms[r], allow_no_votes=True).copy(), world)
            for r in range(world):
                assert not isinstance(res[r], Exception), res[r]
                assert np.array_equal(res[r], T1) and cells_equal(models[r].last_cells()[0], cells1), (thr, r)
            if thr > 0.9:
                assert len(cells1) < world                               # fewer surviving cells than ranks
            else:
                assert len(cells1) > 8                                   # the union is larger than the small buffers
            for m, s in zip(models, scenes):
                m.close()
                s.close()
    finally:
        for cm in comms:
            cm.close()


@pytest.mark.parametrize("stage", [1, 2, 3])
def test_a_failure_on_one_rank_releases_every_rank(ppf, built_lib, case_small, stage):
    """A rank that fails between two collectives (here: injected, as an allocation failure would) reports it with
    the next collective: it returns its own error, every other rank OSLAM_E_PEER, nobody hangs, and the same
    communicator then carries a registration that succeeds."""
    c, df, world, bad = case_small, 2, 3, 1
    # stage 3 (growing the record buffer) is only reached when the union does not fit somewhere
    # (every rank's own buffer grows to its own peaks during the votes: a threshold low enough that the union of three
    # ranks is larger still)
    par = dict(vote_count_threshold=0.02, max_cells=8) if stage == 3 else {}
    T1, cells1, _ = _single(ppf, c, df, **{k: v for k, v in par.items() if k != "max_cells"})
    comms = ppf.Comm.loopback(world)
    try:
        models, scenes = _shards(ppf, c, df, world, **par)
        comms[bad].inject_failure(stage)
        res = _run_ranks(lambda r: models[r].align_multi(scenes[r], comms[r], allow_no_votes=True).copy(), world)
        for r in range(world):
            assert isinstance(res[r], ppf.OslamError), (r, res[r])
            assert res[r].code == (ppf.OSLAM_E_DEVICE if r == bad else ppf.OSLAM_E_PEER), (r, res[r])
        assert not any(cm.broken for cm in comms)
        res = _run_ranks(lambda r: models[r].align_multi(scenes[r], comms[r], allow_no_votes=True).copy(), world)
        for r in range(world):
            assert not isinstance(res[r], Exception), res[r]
            assert np.array_equal(res[r], T1) and cells_equal(models[r].last_cells()[0], cells1)
    finally:
        for cm in comms:
            cm.close()


def test_an_aborted_communicator_releases_its_peers_and_refuses_further_calls(ppf, built_lib, case_small):
    """One of two ranks gives its end up (oslam_comm_abort: ncclCommAbort for the RCCL transport) instead of entering
    the exchange: the other rank's call comes back with an error instead of waiting for ever, and both handles refuse
    further calls."""
    c, world = case_small, 2
    comms = ppf.Comm.loopback(world)
    try:
        models, scenes = _shards(ppf, c, 2, world)
        comms[1].abort()
        res = _run_ranks(lambda r: models[r].align_multi(scenes[r], comms[r], allow_no_votes=True), 1)   # rank 0 alone
        assert isinstance(res[0], ppf.OslamError) and res[0].code == ppf.OSLAM_E_DEVICE
        assert comms[0].broken and comms[1].broken
        for r in range(world):
            with pytest.raises(ppf.OslamError):
                models[r].align_multi(scenes[r], comms[r])
    finally:
        for cm in comms:
            cm.close()


@pytest.mark.parametrize("world", [2, 3])
def test_database_split_by_model(ppf, built_lib, synth, world):
    """oslam_db_align_multi: five models dealt to the ranks (model j on rank j % world), every rank registers its own
    against the WHOLE scene and one all-gather brings every pose to every rank: equal to one database on one GPU."""
    ids = [0, 2, 4, 6, 8]
    clouds = [synth.make_model(k, 700) for k in ids]
    d = synth.d_dist_for(clouds[0][0], 0.05)
    sp, sn, _ = synth.make_scene(ids[:3], 7000, 2093, instance_points=700, noise_sigma=0.05 * d)
    sc = ppf.Scene(sp, sn, d_dist=0.0, ref_point_downsample_factor=3)
    ref_models = [ppf.Model(c[0], c[1], d_dist=d) for c in clouds]
    db1 = ppf.Database(ref_models)
    T_ref, _ = db1.align(sc)
    db1.close()
    comms = ppf.Comm.loopback(world)
    try:
        dbs = []
        for r in range(world):
            mine = [ppf.Model(clouds[j][0], clouds[j][1], d_dist=d) for j in range(r, len(ids), world)]
            dbs.append(ppf.Database(mine) if mine else None)
        res = _run_ranks(lambda r: ppf.db_align_multi(dbs[r], sc, comms[r], len(ids)), world)
        for r in range(world):
            assert not isinstance(res[r], Exception), res[r]
            T, found = res[r]
            assert np.array_equal(T, T_ref), r
            assert np.array_equal(found, np.any(T_ref.reshape(len(ids), 16) != 0, axis=1).astype(np.int32))
        for db in dbs:
            if db is not None:
                db.close()
        if world == 3:
            # more ranks than models: the rank without a database takes part in the exchange all the same
            dbs = [ppf.Database([ppf.Model(clouds[j][0], clouds[j][1], d_dist=d)]) for j in range(2)] + [None]
            res = _run_ranks(lambda r: ppf.db_align_multi(dbs[r], sc, comms[r], 2), world)
            for r in range(world):
                assert not isinstance(res[r], Exception), res[r]
                assert np.array_equal(res[r][0], T_ref[:2]), r
            for db in dbs[:2]:
                db.close()
    finally:
        for cm in comms:
            cm.close()
