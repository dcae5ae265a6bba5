"""The bench workload itself -- the 5k-point model against the 100k-point scene, ref_point_df 8: 12 500 reference
points, 1.9e11 votes -- against the oracle's WHOLE registration of it, committed as
tests/golden/case_5k_100k_df8.npz (tests/golden/make_golden.py fullsize: 50 minutes of 6 host threads; kept cells in
order, counters, poses).  It is an oracle output, like the other case_*.npz: it pins nothing about the reference,
but it makes full-size parity something every run of the suite observes (model.cu:95-171,269-306)."""
import hashlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "case_5k_100k_df8.npz")


def _workload(synth, z):
    M, S = int(z["M"]), int(z["S"])
    mp, mn = synth.make_model(0, M)
    d = synth.d_dist_for(mp, float(z["tau_d"]))
    sp, sn, poses = synth.make_scene([0], S, int(z["seed"]), instance_points=M, noise_sigma=0.1 * d)
    h = hashlib.sha256()
    for a in (mp, mn, sp, sn):
        h.update(np.ascontiguousarray(a, np.float32).tobytes())
    assert h.hexdigest() == str(z["clouds_sha256"]), "the synthetic clouds differ from the ones the fixture was made with"
    assert np.float32(d) == z["d_dist"]
    return mp, mn, sp, sn, d


def test_fixture_is_consistent_with_the_oracle(oracle, synth):
    """CPU: the committed cells give the committed poses through the oracle's pose tail (the votes themselves take 50
    minutes and are not repeated here)."""
    z = np.load(FIXTURE)
    mp, mn, sp, sn, d = _workload(synth, z)
    cells = np.zeros(len(z["cell_code"]), oracle.CELL_DTYPE)
    cells["code"], cells["count"] = z["cell_code"], z["cell_count"]
    assert len(cells) == int(z["stats"][6]) and int(cells["count"][0]) == int(z["stats"][5])
    assert np.all(np.diff(cells["count"].astype(np.int64)) <= 0)                   # count descending
    for tag, kw in (("gpu", {}), ("cpu", dict(cpu_clustering=True)), ("avg", dict(use_averaged_clusters=True))):
        rc, T = oracle.pose_from_cells(cells, mp, mn, sp, sn, d, **kw)
        assert np.array_equal(T, z["T_" + tag]), tag
    poses = oracle.trans_calc2(cells, mp, mn, sp, sn)
    assert hashlib.sha256(poses.tobytes()).hexdigest() == str(z["poses_sha256"])


@pytest.mark.gpu
def test_bench_registration_equals_the_oracle_cell_for_cell(ppf, built_lib, synth):
    """GPU: the whole registration bench.py times -- every kept cell in order, every counter, every pose matrix and the
    returned pose -- equals the oracle's, with the pose tail on the host and on the device, and with the two host-only
    variants."""
    z = np.load(FIXTURE)
    mp, mn, sp, sn, d = _workload(synth, z)
    df = int(z["df"])
    keys = ("num_scene_ppfs", "num_hits", "num_votes", "num_unique_votes", "num_model_keys", "max_count", "num_top")
    sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df)
    for tag, par, flags in (("gpu", dict(pose_gpu_min=1000000000), {}), ("gpu", dict(pose_gpu_min=2), {}),
                            ("cpu", {}, dict(cpu_clustering=True)), ("avg", {}, dict(use_averaged_clusters=True))):
        mo = ppf.Model(mp, mn, d_dist=d, params=ppf.default_params(**par), **flags)
        T = mo.ppf_lookup(sc)
        cells, poses = mo.last_cells()
        assert np.array_equal(cells["code"], z["cell_code"]) and np.array_equal(cells["count"], z["cell_count"]), (tag, par)
        for k, v in zip(keys, z["stats"]):
            assert int(mo.stats[k]) == int(v), (tag, k)
        assert hashlib.sha256(np.ascontiguousarray(poses, np.float32).tobytes()).hexdigest() == str(z["poses_sha256"]), (tag, par)
        assert np.array_equal(T, z["T_" + tag]), (tag, par)
        mo.close()


@pytest.mark.gpu
def test_bench_registration_in_batches_equals_the_single_launch(ppf, built_lib, synth):
    """GPU: the same registration with the hit-list pool limited to 2 GiB (oslam_params.scratch_gib): the 12 500
    reference points no longer fit one launch (they need 4.3 GB) and are voted in batches -- same cells, same counters,
    same pose as the oracle's, and the pool keeps its size from one registration to the next."""
    z = np.load(FIXTURE)
    mp, mn, sp, sn, d = _workload(synth, z)
    ppf.release_scratch(0)          # whatever earlier registrations left: the pool only ever grows
    sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=int(z["df"]))
    mo = ppf.Model(mp, mn, d_dist=d, params=ppf.default_params(scratch_gib=2))
    pool = []
    for _ in range(2):
        T = mo.ppf_lookup(sc)
        cells, _ = mo.last_cells()
        assert mo.stats["vote_launches"] >= 2
        assert np.array_equal(cells["code"], z["cell_code"]) and np.array_equal(cells["count"], z["cell_count"])
        assert int(mo.stats["num_votes"]) == int(z["stats"][2]) and int(mo.stats["num_hits"]) == int(z["stats"][1])
        assert np.array_equal(T, z["T_gpu"])
        pool.append(mo.stats["scratch_bytes"])
    assert pool[0] == pool[1] <= (2 << 30) + (1 << 20)
    mo.close()
    ppf.release_scratch(0)          # the next test gets the default pool again
