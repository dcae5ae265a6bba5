"""CPU: the oracle against every recorded output of the reference, the
committed fixtures, its own two back-ends, and the geometric properties the reference's
check scripts test (matlab/utils/transform_check.m, translation_vector_processing_check.m)."""
import json
import os

import numpy as np
import pytest

from conftest import cells_equal, make_case

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_survey_known_answers(oracle):
    ka = json.load(open(os.path.join(GOLD, "survey_known_answers.json")))
    assert oracle.hash_bytes(bytes.fromhex(ka["fnv1a"]["bytes_hex"])) == int(ka["fnv1a"]["hash"], 16)
    # textbook (unsigned-byte) FNV-1a gives 0xd86e2da0 here: the sign extension of kernel.cu:24-26 matters
    assert oracle.hash_bytes(bytes.fromhex(ka["fnv1a"]["bytes_hex"])) != 0xD86E2DA0
    assert np.float32(oracle.d_angle0()) == np.float32(ka["d_angle0"])
    c = ka["cloud"]
    ppf, keys = oracle.ppf_all_pairs(np.float32(c["points"]), np.float32(c["normals"]), c["df"], c["d_dist"])
    for ij, want in ka["keys"].items():
        i, j = map(int, ij.split(","))
        assert int(keys[i, j]) == int(want, 16), ij
    for ij, want in ka["disc_ppf"].items():
        i, j = map(int, ij.split(","))
        assert np.array_equal(ppf[i, j], np.float32(want)), ij


@pytest.mark.parametrize("name", ["case_m64_s128", "case_m200_s400_df3"])
def test_golden_fixtures(oracle, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    d, df = float(g["d_dist"]), int(g["df"])
    _, mkeys = oracle.ppf_all_pairs(g["mp"], g["mn"], 1, d, want_ppf=False)
    assert np.array_equal(mkeys, g["model_keys"])
    assert np.array_equal(oracle.ppf_row_keys(g["sp"], g["sn"], 0, d), g["scene_keys_row0"])
    cells, st = oracle.votes_fused(g["mp"], g["mn"], g["sp"], g["sn"], df, d, 0.4)
    assert np.array_equal(cells["code"], g["cell_code"]) and np.array_equal(cells["count"], g["cell_count"])
    want = dict(zip(("num_scene_ppfs", "num_hits", "num_votes", "num_unique_votes", "num_model_keys",
                     "max_count", "num_top"), map(int, g["stats"])))
    assert {k: st[k] for k in want} == want
    # poses go through libm sinf/cosf, whose x86 variants (FMA or not) may differ in the last bit
    np.testing.assert_allclose(oracle.trans_calc2(cells, g["mp"], g["mn"], g["sp"], g["sn"]), g["poses"],
                               rtol=0, atol=2e-5)
    for tag, kw in (("gpu", {}), ("cpu", {"cpu_clustering": True})):
        rc, T = oracle.pose_from_cells(cells, g["mp"], g["mn"], g["sp"], g["sn"], d, **kw)
        np.testing.assert_allclose(T, g["T_" + tag], rtol=0, atol=2e-5)


def test_literal_and_fused_backends_agree(oracle, synth):
    c = make_case(synth, 120, 260, 2010)
    lit, st_l, allc = oracle.votes_literal(c["mp"], c["mn"], c["sp"], c["sn"], 2, c["d"], want_all=True)
    fus, st_f = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], 2, c["d"])
    assert cells_equal(lit, fus) and st_l == st_f
    # the dense accumulator of one reference point equals the literal cells of that point
    r = 4
    acc = oracle.accumulator_for_ref(c["mp"], c["mn"], c["sp"], c["sn"], r, c["d"])
    mine = allc[(allc["code"] >> 32) == r]
    dense = np.zeros_like(acc)
    dense[((mine["code"] & 0xFFFFFFFF) >> 6).astype(int), (mine["code"] & 63).astype(int)] = mine["count"]
    assert np.array_equal(acc, dense)
    # sharded reference points partition the work
    parts = [oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], 2, c["d"], ref_begin=k, ref_step=3)[1]
             for k in range(3)]
    assert sum(p["num_votes"] for p in parts) == st_f["num_votes"]


def _rigid(rng):
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    return R, rng.uniform(-2, 2, 3)


def test_transform_check_property(oracle):
    """transform_check.m: for a model pair moved by a rigid G, inv(T_s_g) * rotx(alpha) * T_m_g
    reproduces G -- here up to the alpha bin (the reference keeps the lower bin edge,
    kernel.cu:341-343): reference point and normal map exactly, the rest within one bin."""
    rng = np.random.default_rng(11)
    D = oracle.d_angle0()
    for _ in range(200):
        m_r, m_i = rng.uniform(-1, 1, 3), rng.uniform(-1, 1, 3)
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        R, t = _rigid(rng)
        s_r, s_i, n_s = R @ m_r + t, R @ m_i + t, R @ n
        idx = oracle.trans_model_scene(m_r, n, m_i, s_r, n_s, s_i)
        assert 0 <= idx <= 30
        # the pose of that accumulator cell, as K5 computes it (kernel.cu:372-401; the T that
        # trans_model_scene itself builds at :343-348 uses alpha + pi and is discarded there)
        cells = np.zeros(2, oracle.CELL_DTYPE)
        cells["code"] = (1 << 32) | (1 << 6) | idx     # reference points at index 1: code (0,0,0) is skipped (:628)
        cells["count"] = 1
        mp2, mn2 = np.float32([m_i, m_r]), np.float32([n, n])
        sp2, sn2 = np.float32([s_i, s_r]), np.float32([n_s, n_s])
        T = oracle.trans_calc2(cells, mp2, mn2, sp2, sn2)[0].reshape(4, 4).astype(np.float64)
        assert np.allclose(T[:3, :3] @ m_r + T[:3, 3], s_r, atol=2e-5)       # reference point
        assert np.allclose(T[:3, :3] @ n, n_s, atol=2e-5)                    # its normal
        G = np.eye(4)
        G[:3, :3], G[:3, 3] = R, t
        dt, dr = oracle.ht_dist(T, G)
        assert dr < D + 1e-4


def test_pose_recovery_whole_scene(oracle, synth):
    """translation_vector_processing_check.m: the scene is a rigid copy of the model."""
    mp, mn = synth.make_model(0, 150)
    rng = np.random.default_rng(3)
    R, t = _rigid(rng)
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, t
    sp, sn = synth.transform_cloud(mp, mn, T)
    d = synth.d_dist_for(mp, 0.05)
    Tr, cells, st = oracle.align(mp, mn, sp, sn, 1, d)
    dt, dr = oracle.ht_dist(Tr, T)
    assert dr < np.deg2rad(12) and dt < 0.1 * synth.bbox_extent(mp)


def test_reference_quirks_are_kept(oracle, synth):
    c = make_case(synth, 60, 120, 2011)
    cells, _ = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], 1, c["d"])
    one = cells[:1]
    # a single surviving cell: every kernel returns early for count <= 1 (kernel.cu:609,651,667,712)
    rc, T = oracle.pose_from_cells(one, c["mp"], c["mn"], c["sp"], c["sn"], c["d"])
    assert rc == 0 and np.all(T == 0)
    rc, T = oracle.pose_from_cells(cells[:0], c["mp"], c["mn"], c["sp"], c["sn"], c["d"])
    assert rc == 1 and np.all(T == 0)
    # own votes count as 1 and the centre cell is skipped (kernel.cu:684-689,722)
    Tall = oracle.trans_calc2(cells, c["mp"], c["mn"], c["sp"], c["sn"])
    tr, qu = oracle.mat2transquat(Tall)
    best, scores, _ = oracle.cluster_gpu_style(cells, tr, qu, c["d"])
    assert scores.min() >= 1.0 and scores[best] == scores.max()
    # quaternion normalised by |q|^(1/2), not |q| (kernel.cu:138-142): unit quaternions stay unit
    assert np.allclose(np.linalg.norm(qu, axis=1), 1.0, atol=1e-4)


def test_ht_dist(oracle):
    I = np.eye(4)
    assert oracle.ht_dist(I, I) == (0.0, 0.0)
    a = 0.3
    Rz = np.array([[np.cos(a), -np.sin(a), 0, 1], [np.sin(a), np.cos(a), 0, 2], [0, 0, 1, 2], [0, 0, 0, 1.0]])
    dt, dr = oracle.ht_dist(I, Rz)
    assert abs(dt - 3.0) < 1e-6 and abs(dr - a) < 1e-6
