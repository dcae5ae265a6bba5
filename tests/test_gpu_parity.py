"""GPU parity: the HIP path through the C-ABI against the CPU oracle on the same
seeded clouds.  Integer results (keys, buckets, accumulators, peak cells) must be
identical; the pose is produced by the same host stage from identical cells, so
it is compared exactly too."""
import numpy as np
import pytest

from conftest import cells_equal, make_case

pytestmark = pytest.mark.gpu


def test_device_float_path_matches_host(ppf, built_lib):
    # pm_acosf / pm_atan2f / quantisation / alpha bin: 4M pseudo-random inputs, GPU vs host bits
    assert ppf.selftest_math(1 << 22, seed=7) == 0


def test_scene_and_model_keys_bit_exact(ppf, oracle, built_lib, case_small):
    c = case_small
    sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"])
    mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"])
    for r in (0, 1, 17, len(c["sp"]) - 1):
        assert np.array_equal(sc.getHashKeys(r), oracle.ppf_row_keys(c["sp"], c["sn"], r, c["d"]))
    for r in (0, 5, len(c["mp"]) - 1):
        assert np.array_equal(mo.getHashKeys(r), oracle.ppf_row_keys(c["mp"], c["mn"], r, c["d"]))


def test_keys_with_degenerate_geometry(ppf, oracle, built_lib):
    # duplicate points (|d| = 0 -> NaN angles), parallel/antiparallel and identical
    # non-unit normals (acosf argument rounds past 1 -> NaN bytes hashed), zero normal
    rng = np.random.default_rng(5)
    p = rng.uniform(-1, 1, (96, 3)).astype(np.float32)
    n = rng.normal(size=(96, 3)).astype(np.float32)
    p[10] = p[3]
    n[20:40] = np.float32([0.6, 0.0, 0.8])
    n[40:50] = np.float32([0.3, 0.1, 0.7])
    n[50] = 0
    p[60:70, 2] = 0
    n[60:70] = np.float32([0, 0, 1])
    p[70] = p[71] + np.float32([0, 0, 1e-7])
    d = 0.07
    sc = ppf.Scene(p, n, d_dist=d)
    for r in range(0, 96, 5):
        assert np.array_equal(sc.getHashKeys(r), oracle.ppf_row_keys(p, n, r, d)), r


def test_model_table_buckets(ppf, oracle, built_lib, case_small):
    c = case_small
    mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"])
    _, keys = oracle.ppf_all_pairs(c["mp"], c["mn"], 1, c["d"], want_ppf=False)
    flat = keys.reshape(-1)
    uniq, counts = np.unique(flat, return_counts=True)
    rng = np.random.default_rng(0)
    pick = rng.choice(len(uniq), size=40, replace=False)
    for u in pick:
        k = int(uniq[u])
        if k == 0:
            continue
        got, n = mo.bucket(k)
        want = np.nonzero(flat == k)[0].astype(np.uint32)
        assert n == counts[u] and np.array_equal(got, want)
    assert mo.bucket(0x12345)[1] == int((flat == 0x12345).sum())


def test_accumulator_exact(ppf, oracle, built_lib, case_small):
    c = case_small
    sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"])
    mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"])
    for r in (0, 3, 250, 499):
        got = mo.vote_accumulator(sc, r)
        want = oracle.accumulator_for_ref(c["mp"], c["mn"], c["sp"], c["sn"], r, c["d"])
        assert np.array_equal(got, want), r


def test_accumulator_long_hit_lists(ppf, oracle, built_lib, synth):
    # scene = a dense sampling of the model's own surface, coarse d_dist: a large share of the scene
    # pairs hit, > 16384 hits per reference point, so the hit list is sorted in segments and a key has
    # runs in several segments; > 64 equal keys in a row also exercises the run split at 64 hits
    mp, mn = synth.make_model(0, 150)
    d = synth.d_dist_for(mp, 0.25)
    sp, sn = synth.make_model(0, 60000)
    sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=20000)
    mo = ppf.Model(mp, mn, d_dist=d)
    mo.ppf_lookup(sc, allow_no_votes=True)
    assert mo.stats["num_hits"] > 3 * 16384 * 1.2      # 3 reference points, each past one segment
    for r in (0, 40000):
        got = mo.vote_accumulator(sc, r)
        want = oracle.accumulator_for_ref(mp, mn, sp, sn, r, d)
        assert np.array_equal(got, want), r


def test_model_database_round_trip(ppf, oracle, built_lib, case_two_slices, tmp_path):
    # a table written with Model.save and mapped back with Model.load votes exactly like the one that
    # was built (cells, counters and pose identical); a flipped byte in the file is refused
    c = case_two_slices
    sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=10)
    built = ppf.Model(c["mp"], c["mn"], d_dist=c["d"])
    T0 = built.ppf_lookup(sc).copy()
    cells0, _ = built.last_cells()
    f = str(tmp_path / "m.oslam")
    built.save(f)
    loaded = ppf.Model.load(f)
    assert loaded.numPoints() == built.numPoints() and loaded.d_dist == np.float32(c["d"])
    assert loaded.table_bytes() == built.table_bytes()
    T1 = loaded.ppf_lookup(sc)
    cells1, _ = loaded.last_cells()
    assert np.array_equal(T0, T1) and cells_equal(cells0, cells1)
    for k in ("num_hits", "num_votes", "num_unique_votes", "num_model_keys", "max_count"):
        assert loaded.stats[k] == built.stats[k], k
    raw = bytearray(open(f, "rb").read())
    raw[len(raw) // 2] ^= 0x40
    g = str(tmp_path / "damaged.oslam")
    open(g, "wb").write(bytes(raw))
    with pytest.raises(ppf.OslamError) as e:
        ppf.Model.load(g)
    assert "checksum" in str(e.value)
    open(g, "wb").write(bytes(raw[: len(raw) // 3]))
    with pytest.raises(ppf.OslamError) as e:
        ppf.Model.load(g)
    assert "truncated" in str(e.value)
    # header fields a kernel indexes with (db_header: magic u64, version, vote_mode, n_points, n_slices, cap,
    # shift, ucap, ushift, n_entries, has_uv): a stale shift, a bucket table that is not a power of two, an exact
    # table without its exact entries and a flipped bit in an unchecked-looking field are all refused before
    # anything reaches the GPU (the checksum covers the header as well)
    good = bytearray(open(f, "rb").read())
    for off, val, what in ((28, 7, "inconsistent"), (24, 3000, "inconsistent"), (44, 0, "inconsistent"), (48, None, "checksum")):
        bad = bytearray(good)
        if val is None:
            bad[off] ^= 1                      # num_model_keys: consistent with everything, but not what was written
        else:
            bad[off:off + 4] = int(val).to_bytes(4, "little")
        open(g, "wb").write(bytes(bad))
        with pytest.raises(ppf.OslamError) as e:
            ppf.Model.load(g)
        assert what in str(e.value), (off, str(e.value))


def _align_and_compare(ppf, oracle, c, df=1, **flags):
    par = ppf.default_params()
    sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=df, params=par)
    mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], params=par, **flags)
    T = mo.ppf_lookup(sc)
    cells, poses = mo.last_cells()
    ocells, ost = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], df, c["d"], 0.4)
    assert cells_equal(cells, ocells)
    st = mo.stats
    for k in ("num_scene_ppfs", "num_hits", "num_votes", "num_unique_votes", "num_model_keys", "max_count"):
        assert st[k] == ost[k], k
    assert st["num_top"] == ost["num_top"]
    rc, To = oracle.pose_from_cells(ocells, c["mp"], c["mn"], c["sp"], c["sn"], c["d"],
                                    cpu_clustering=flags.get("cpu_clustering", False),
                                    use_l1_norm=flags.get("use_l1_norm", False),
                                    use_averaged_clusters=flags.get("use_averaged_clusters", False))
    assert np.array_equal(T, To)
    return T


def test_align_matches_oracle(ppf, oracle, built_lib, case_small):
    T = _align_and_compare(ppf, oracle, case_small)
    dt, dr = ppf.ht_dist(T, case_small["truth"])
    # the reference itself is bin-accurate only (D_ANGLE0 = 12 deg, d_dist): its own
    # acceptance test is 12 deg / 0.1 diameter (alignment.cpp:141-144)
    assert dr < np.deg2rad(12) and dt < 0.1 * 3.8


@pytest.mark.parametrize("flags", [dict(cpu_clustering=True), dict(use_l1_norm=True),
                                   dict(use_averaged_clusters=True)])
def test_align_flags(ppf, oracle, built_lib, case_small, flags):
    _align_and_compare(ppf, oracle, case_small, **flags)


def test_align_df_and_ragged_sizes(ppf, oracle, built_lib, synth):
    # S not a multiple of the 1024-thread tile, df = 3 (last reference index ragged)
    c = make_case(synth, 150, 1031, 2003)
    _align_and_compare(ppf, oracle, c, df=3)


def test_align_two_slices(ppf, oracle, built_lib, case_two_slices):
    # M = 1300 -> two model slices; cells from both slices must be merged exactly
    _align_and_compare(ppf, oracle, case_two_slices, df=10)


def test_sharded_align_equals_single(ppf, oracle, built_lib, case_small):
    c = case_small
    world = 3
    all_cells, gmax = [], 0
    mo = None
    for rank in range(world):
        par = ppf.default_params(shard_rank=rank, shard_world=world)
        sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=2, params=par)
        mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], params=par)
        cells, lmax = mo.align_local(sc, cap=1 << 16)
        all_cells.append(cells)
        gmax = max(gmax, lmax)
    T = mo.align_finish(sc, np.concatenate(all_cells), gmax)
    ocells, _ = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], 2, c["d"], 0.4)
    rc, To = oracle.pose_from_cells(ocells, c["mp"], c["mn"], c["sp"], c["sn"], c["d"])
    assert cells_equal(mo.last_cells()[0], ocells)
    assert np.array_equal(T, To)


def test_no_match_and_bad_arguments(ppf, built_lib, synth):
    mp, mn = synth.make_model(0, 64)
    d = synth.d_dist_for(mp, 0.05)
    # a scene far smaller than one distance bin of interest: two points 100 diameters apart
    sp = np.float32([[0, 0, 0], [400, 0, 0], [0, 400, 0]])
    sn = np.float32([[0, 0, 1], [0, 1, 0], [1, 0, 0]])
    mo = ppf.Model(mp, mn, d_dist=d)
    sc = ppf.Scene(sp, sn, d_dist=d)
    T = mo.ppf_lookup(sc, allow_no_votes=True)
    assert np.all(T == 0) and mo.stats["num_votes"] == 0
    with pytest.raises(ppf.OslamError):
        mo.ppf_lookup(sc)                       # OSLAM_E_NO_VOTES is reported, not swallowed
    with pytest.raises(ppf.OslamError):
        ppf.Model(mp[:1], mn[:1], d_dist=d)     # n < 2
    # d_dist 0 makes a scene for models of any d_dist; a negative one is a bad argument
    assert np.all(mo.ppf_lookup(ppf.Scene(sp, sn, d_dist=0.0), allow_no_votes=True) == 0)
    with pytest.raises(ppf.OslamError):
        ppf.Scene(sp, sn, d_dist=-1.0)
    sc2 = ppf.Scene(sp, sn, d_dist=2 * d)
    with pytest.raises(ppf.OslamError):
        mo.ppf_lookup(sc2)                      # d_dist mismatch (ppf.cu:64-67)


def test_ppf_registration_entry_point(ppf, oracle, built_lib, synth):
    a = make_case(synth, 120, 400, 2004)
    b = make_case(synth, 140, 400, 2005, model_id=1)
    res = ppf.ppf_registration([(a["sp"], a["sn"])], [(a["mp"], a["mn"]), (b["mp"], b["mn"])],
                               [a["d"], b["d"]], ref_point_downsample_factor=2)
    for j, c in enumerate((a, b)):
        ocells, _ = oracle.votes_fused(c["mp"], c["mn"], a["sp"], a["sn"], 2, c["d"], 0.4)
        rc, To = oracle.pose_from_cells(ocells, c["mp"], c["mn"], a["sp"], a["sn"], c["d"])
        assert np.array_equal(res[0, j], To)


# ---------------------------------------------------------------------------
# committed fixtures and BASELINE.json-size properties
# ---------------------------------------------------------------------------
import os  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", ["case_m64_s128", "case_m200_s400_df3"])
def test_golden_fixtures_on_gpu(ppf, built_lib, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    d, df = float(g["d_dist"]), int(g["df"])
    sc = ppf.Scene(g["sp"], g["sn"], d_dist=d, ref_point_downsample_factor=df)
    mo = ppf.Model(g["mp"], g["mn"], d_dist=d)
    for r in (0, 7, len(g["mp"]) - 1):
        assert np.array_equal(mo.getHashKeys(r), g["model_keys"][r])
    assert np.array_equal(sc.getHashKeys(0), g["scene_keys_row0"])
    T = mo.ppf_lookup(sc)
    cells, poses = mo.last_cells()
    assert np.array_equal(cells["code"], g["cell_code"]) and np.array_equal(cells["count"], g["cell_count"])
    # poses involve libm sinf/cosf of the host that wrote the fixture: last-bit tolerance
    np.testing.assert_allclose(poses, g["poses"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(T, g["T_gpu"], rtol=0, atol=2e-5)
    mo2 = ppf.Model(g["mp"], g["mn"], d_dist=d, cpu_clustering=True)
    np.testing.assert_allclose(mo2.ppf_lookup(sc), g["T_cpu"], rtol=0, atol=2e-5)


def test_record_buffer_overflow_takes_the_second_pass(ppf, oracle, built_lib, case_small):
    c = case_small
    par = ppf.default_params(max_cells=8)          # far fewer than the cells the first pass emits
    sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], params=par)
    mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], params=par)
    ocells, _ = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], 1, c["d"], 0.4)
    assert len(ocells) > 8
    T = mo.ppf_lookup(sc)                          # even the exact set does not fit: the buffer grows, third launch
    assert cells_equal(mo.last_cells()[0], ocells) and mo.stats["vote_launches"] == 3
    assert np.array_equal(T, oracle.pose_from_cells(ocells, c["mp"], c["mn"], c["sp"], c["sn"], c["d"])[1])
    par = ppf.default_params(max_cells=max(16, 2 * len(ocells)))
    mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], params=par)
    mo.ppf_lookup(sc)
    assert cells_equal(mo.last_cells()[0], ocells) and mo.stats["vote_launches"] >= 1


@pytest.fixture(scope="module")
def full_size(synth):
    """BASELINE.json metric configuration: 5k-point model, 100k-point scene."""
    mp, mn = synth.make_model(0, 5000)
    d = synth.d_dist_for(mp, 0.025)
    sp, sn, poses = synth.make_scene([0], 100000, 2002, instance_points=5000, noise_sigma=0.1 * d)
    return dict(mp=mp, mn=mn, sp=sp, sn=sn, d=d, truth=poses[0][1], diam=synth.bbox_extent(mp))


def test_full_size_sample_against_oracle(ppf, oracle, built_lib, full_size):
    """Five reference points of the 5k x 100k workload: dense accumulators against the oracle's
    per-reference counters (sum = votes, non-empty cells, maximum) and the 10 largest cells."""
    c = full_size
    sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=8)
    mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"])
    fm = oracle.FusedModel(c["mp"], c["mn"], c["d"])
    assert mo.bucket(int(mo.getHashKeys(0)[1]))[1] > 0
    for k in (0, 17, 402, 6000, 12499):
        ocells, st = fm.votes(c["sp"], c["sn"], 8, thresh=0.0, ref_begin=k, ref_limit=1)
        acc = mo.vote_accumulator(sc, 8 * k)
        assert int(acc.sum()) == st["num_votes"]
        assert int(np.count_nonzero(acc)) == st["num_unique_votes"]
        assert int(acc.max()) == st["max_count"]
        top = ocells[:10]
        got = acc[((top["code"] & 0xFFFFFFFF) >> 6).astype(int), (top["code"] & 63).astype(int)]
        assert np.array_equal(got, top["count"])
    fm.close()


def test_full_size_properties(ppf, oracle, built_lib, full_size):
    c = full_size
    par = ppf.default_params()
    sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=8, params=par)
    mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], params=par)
    T1 = mo.ppf_lookup(sc)
    cells1, st1 = mo.last_cells()[0], dict(mo.stats)
    T2 = mo.ppf_lookup(sc)                                           # determinism / idempotence
    assert np.array_equal(T1, T2) and cells_equal(cells1, mo.last_cells()[0])
    assert st1["num_scene_ppfs"] == 12500 * 99999
    # 4350 cells: the clustering scores come from the GPU kernel (>= 2048 cells); the oracle's
    # serial host-order loop must give the same pose, for both clustering norms
    assert len(cells1) >= 2048
    rc, To = oracle.pose_from_cells(cells1, c["mp"], c["mn"], c["sp"], c["sn"], c["d"])
    assert np.array_equal(T1, To)
    ml1 = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], use_l1_norm=True)
    rc, To = oracle.pose_from_cells(cells1, c["mp"], c["mn"], c["sp"], c["sn"], c["d"], use_l1_norm=True)
    assert np.array_equal(ml1.ppf_lookup(sc), To)
    # shards partition the votes and their union reproduces the single-GPU cells and pose
    tot_votes = tot_hits = 0
    parts, gmax = [], 0
    for rank in range(2):
        ps = ppf.default_params(shard_rank=rank, shard_world=2)
        scs = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=8, params=ps)
        loc, lmax = mo.align_local(scs, cap=1 << 16)
        tot_votes += mo.stats["num_votes"]
        tot_hits += mo.stats["num_hits"]
        parts.append(loc)
        gmax = max(gmax, lmax)
    assert (tot_votes, tot_hits, gmax) == (st1["num_votes"], st1["num_hits"], st1["max_count"])
    T3 = mo.align_finish(sc, np.concatenate(parts), gmax)
    assert np.array_equal(T3, T1) and cells_equal(mo.last_cells()[0], cells1)
    # pose against ground truth at the reference's own acceptance test (alignment.cpp:141-144,317-323)
    dt, dr = ppf.ht_dist(T1, c["truth"])
    assert dr < np.deg2rad(12) and dt < 0.1 * c["diam"]
    # fast mode (no re-evaluation near bin edges): pose within 1 deg / 1 % of the diameter of exact
    pf = ppf.default_params(vote_mode=ppf.VOTE_FAST)
    mf = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], params=pf)
    Tf = mf.ppf_lookup(ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=8, params=pf))
    dtf, drf = ppf.ht_dist(Tf, T1)
    assert drf < np.deg2rad(1.0) and dtf < 0.01 * c["diam"]
    assert mf.stats["num_votes"] == st1["num_votes"] and mf.stats["max_count"] >= 0.99 * st1["max_count"]


def test_result_fields_of_the_reference_model(ppf, oracle, built_lib, case_small):
    """transformation_trans / transformation_rots / vote_counts_out / max_idx (model.h:100-113) of a lookup equal
    the oracle's K7 quaternions and translations and its clustering scores; the pose is extracted from them as
    ppf.cu:74-93 does."""
    c = case_small
    sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"])
    for averaged in (False, True):
        mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], use_averaged_clusters=averaged)
        T = mo.ppf_lookup(sc)
        cells, poses = mo.last_cells()
        tr, ro, scores, best = mo.last_result(sc)
        otr, oqu = oracle.mat2transquat(oracle.trans_calc2(cells, c["mp"], c["mn"], c["sp"], c["sn"]))
        obest, oscores, otr2 = oracle.cluster_gpu_style(cells, otr, oqu, c["d"], use_averaged_clusters=averaged)
        assert np.array_equal(ro, oqu) and np.array_equal(tr, otr2) and np.array_equal(scores, oscores) and best == obest
        Tx = poses[best].reshape(4, 4).copy()
        Tx[:3, 3] = tr[best]
        assert np.array_equal(Tx, T)


def test_cpp_adaptor_runs(ppf, built_lib):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "build", "pcl_adaptor_check_gpu")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    libdir = os.path.dirname(ppf.LIB_PATH)
    subprocess.run(["g++", "-std=c++14", "-I", os.path.join(root, "include"),
                    os.path.join(root, "tests", "native", "pcl_adaptor_check.cpp"), "-o", out,
                    "-L", libdir, "-loslam_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    r = subprocess.run([out, "run"], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "fields ok" in r.stdout          # model.h:92-113 result fields give the pose ppf.cu:74-93 extracts
    t = [float(x) for x in r.stdout.split("=")[1].split()[:3]]
    # the scene is the model shifted by (2, -1, 0); bin-accurate recovery (d_dist = 0.05)
    assert abs(t[0] - 2.0) < 0.3 and abs(t[1] + 1.0) < 0.3 and abs(t[2]) < 0.3


def test_accumulator_with_marker_paths(ppf, oracle, built_lib, synth):
    """Coordinates of magnitude 2^45: every stored angle carries the 'always re-evaluate' marker
    (ppf_core.h: vectors outside 2^-40..2^40), so all votes take the queued exact path; and a
    tiny cloud (2^-45) for the other side of the range."""
    for scale in (2.0 ** 45, 2.0 ** -45):
        c = make_case(synth, 90, 200, 2040)
        mp, sp = (c["mp"] * np.float32(scale)), (c["sp"] * np.float32(scale))
        d = float(np.float32(c["d"]) * np.float32(scale))
        sc = ppf.Scene(sp, c["sn"], d_dist=d)
        mo = ppf.Model(mp, c["mn"], d_dist=d)
        for r in (0, 57, 199):
            assert np.array_equal(mo.vote_accumulator(sc, r),
                                  oracle.accumulator_for_ref(mp, c["mn"], sp, c["sn"], r, d)), (scale, r)
        mo.ppf_lookup(sc, allow_no_votes=True)
        ocells, _ = oracle.votes_fused(mp, c["mn"], sp, c["sn"], 1, d, 0.4)
        assert cells_equal(mo.last_cells()[0], ocells)


def test_baseline_config0_1k_model_5k_scene(ppf, oracle, built_lib, synth):
    """BASELINE.json configs[0] (1k-point model vs 5k-point scene, the reference's own CPU-runnable
    case; no bunny.ply exists in the reference, so model 0 stands in): every peak cell, every
    counter and the pose against the oracle."""
    c = make_case(synth, 1000, 5000, 2050, instance_points=1000)
    _align_and_compare(ppf, oracle, c, df=5)


def test_baseline_config2_model_database(ppf, oracle, built_lib, synth):
    """BASELINE.json configs[2] shape, reduced: a 4-model database against one scene that holds an
    instance of each; ppf_registration keeps all tables resident.  Each pose is checked against
    ground truth at the reference's acceptance test, one model against the oracle exactly."""
    ids = [0, 1, 2, 3]
    models = [synth.make_model(k, 700) for k in ids]
    dd = [synth.d_dist_for(m[0], 0.05) for m in models]
    sp, sn, poses = synth.make_scene(ids, 12000, 2051, instance_points=700)
    res = ppf.ppf_registration([(sp, sn)], models, dd, ref_point_downsample_factor=4)
    ok = 0
    for j, (mid, T) in enumerate(poses):
        dt, dr = ppf.ht_dist(res[0, j], T)
        ok += int(dr < np.deg2rad(12) and dt < 0.1 * synth.bbox_extent(models[j][0]))
    # recall here is a property of the reference algorithm on this cluttered scene (global 0.4*max
    # threshold, 12-degree bins), not of this build: the exact check is the oracle comparison below
    assert ok >= 2, ok
    ocells, _ = oracle.votes_fused(models[1][0], models[1][1], sp, sn, 4, dd[1], 0.4)
    rc, To = oracle.pose_from_cells(ocells, models[1][0], models[1][1], sp, sn, dd[1])
    assert np.array_equal(res[0, 1], To)


def test_voxel_grid_equals_oracle_statement(ppf, oracle, built_lib, synth):
    """voxelGridDownsample (alignment.cpp:79-87).  PCL is not available, so parity with PCL is
    unpinned; the GPU stage must equal the oracle's statement of PCL's algorithm exactly."""
    mp, mn = synth.make_model(0, 30000)
    sp, sn, _ = synth.make_scene([0], 60000, 2060, instance_points=8000, noise_sigma=0.002)
    for pts, nrm, leaf in ((mp, mn, 0.05), (mp, mn, 0.19), (sp, sn, 0.25), (sp, sn, 3.0)):
        go, gn = ppf.voxel_grid(pts, nrm, leaf=leaf)
        oo, on = oracle.voxel_grid(pts, nrm, leaf)
        assert len(go) == len(oo) and 0 < len(go) <= len(pts)
        assert np.array_equal(go, oo) and np.array_equal(gn, on)
    # non-finite points are ignored; a leaf that makes the voxel count overflow int32 is an error
    bad = mp.copy()
    bad[5] = np.nan
    go, gn = ppf.voxel_grid(bad, mn, leaf=0.1)
    oo, on = oracle.voxel_grid(bad, mn, 0.1)
    assert np.array_equal(go, oo) and np.array_equal(gn, on)
    with pytest.raises(ppf.OslamError):
        ppf.voxel_grid(sp, sn, leaf=1e-5)
    # the reference's pipeline: model voxel-gridded at leaf = d_dist, then registered
    d = synth.d_dist_for(mp, 0.05)
    mg, mgn = ppf.voxel_grid(mp, mn, leaf=d)
    sg, sgn = ppf.voxel_grid(sp, sn, leaf=0.5 * d)
    T = ppf.Model(mg, mgn, d_dist=d).ppf_lookup(ppf.Scene(sg, sgn, d_dist=d, ref_point_downsample_factor=5))
    assert T.shape == (4, 4)


def test_cli_front_end(ppf, built_lib, synth, tmp_path):
    """objective-slam_amd/oslam_alignment: the reference's `alignment` flags and flow
    (alignment.cpp:191-335): PLY in, d_dist rule, voxel grids, registration, validation 0/1."""
    import subprocess
    exe = os.path.join(os.path.dirname(ppf.LIB_PATH), "oslam_alignment")
    mp, mn = synth.make_model(0, 6000)
    sp, sn, poses = synth.make_scene([0], 20000, 2070, instance_points=6000)
    ppf.ply_write(str(tmp_path / "model.ply"), mp, mn, binary=True)
    ppf.ply_write(str(tmp_path / "scene.ply"), sp, sn, binary=False)
    np.savetxt(str(tmp_path / "truth.txt"), poses[0][1])
    tau, leaf = 0.05, 0.12
    r = subprocess.run([exe, "--scene_files", str(tmp_path / "scene.ply"), "--model_files", str(tmp_path / "model.ply"),
                        "--tau_d", str(tau), "--scene_leaf_size", str(leaf), "--ref_point_df", "2", "--dev", "0",
                        "--validation_files", str(tmp_path / "truth.txt"), "--visualize", "false"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == "1", r.stderr
    # the same flow through the library gives the same matrix
    d = ppf.d_dist_from_cloud(mp, tau)
    mg, mgn = ppf.voxel_grid(mp, mn, leaf=d)
    sp2, sn2 = ppf.ply_read(str(tmp_path / "scene.ply"))          # ascii: 9 significant digits
    sg, sgn = ppf.voxel_grid(sp2, sn2, leaf=leaf)
    T = ppf.Model(mg, mgn, d_dist=d).ppf_lookup(ppf.Scene(sg, sgn, d_dist=d, ref_point_downsample_factor=2))
    lines = r.stderr.split("Transformations for")[1].splitlines()[1:5]
    Tcli = np.array([[float(x) for x in ln.split()] for ln in lines], np.float32)
    np.testing.assert_allclose(Tcli, T, atol=2e-6)


def test_random_small_clouds_match_oracle(ppf, oracle, built_lib):
    """Seeded sweep over ragged sizes (down to 2 points), tau_d, df, normals that are not unit
    length, duplicated points and planar patches: peak cells, counters and pose against the oracle."""
    rng = np.random.default_rng(20260)
    for trial in range(24):
        M = int(rng.choice([2, 3, 5, 17, 64, 130, 333, 1025, 1100]))
        S = int(rng.choice([2, 3, 9, 65, 257, 700, 1024, 1500]))
        df = int(rng.choice([1, 1, 2, 7]))
        mp = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
        mn = (rng.normal(size=(M, 3)) * rng.uniform(0.2, 3.0, (M, 1))).astype(np.float32)
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        k = min(M, S)
        sp = rng.uniform(-2, 2, (S, 3)).astype(np.float32)
        sn = rng.normal(size=(S, 3)).astype(np.float32)
        sp[:k] = (mp[:k] @ R.T + rng.uniform(-1, 1, 3)).astype(np.float32)      # part of the model, moved
        sn[:k] = (mn[:k] @ R.T).astype(np.float32)
        if S > 20:
            sp[k // 2] = sp[0]                                                   # duplicate point
            sp[-8:, 2] = 0.5                                                     # planar patch, equal normals
            sn[-8:] = np.float32([0, 0, 2])
        d = float(np.float32(rng.choice([0.05, 0.11, 0.3]) * 2.0))
        par = ppf.default_params()
        sc = ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=df, params=par)
        mo = ppf.Model(mp, mn, d_dist=d, params=par)
        T = mo.ppf_lookup(sc, allow_no_votes=True)
        ocells, ost = oracle.votes_fused(mp, mn, sp, sn, df, d, 0.4)
        cells = mo.last_cells()[0]
        assert cells_equal(cells, ocells), (trial, M, S, df, d)
        for key in ("num_scene_ppfs", "num_hits", "num_votes", "num_unique_votes", "max_count"):
            assert mo.stats[key] == ost[key], (trial, key)
        if len(ocells):
            rc, To = oracle.pose_from_cells(ocells, mp, mn, sp, sn, d)
            assert np.array_equal(T, To), (trial, M, S)


def test_depth_front_end_equals_its_statement(ppf, oracle, built_lib, synth):
    """Depth image -> points + normals (oslam_depth_to_cloud) against oracle/oracle_depth.c, bit for
    bit, on uint16 and float images with holes, depth edges and a ragged size; a tilted plane gives
    the plane's normal at every pixel.  The reference has no such step: parity unpinned."""
    rng = np.random.default_rng(11)
    h, w = 61, 83
    v, u = np.mgrid[0:h, 0:w]
    z = 1.2 + 0.002 * u + 0.004 * v + 0.05 * np.sin(u / 5.0) * np.cos(v / 7.0)
    z[20:30, 40:55] = 0.6                      # an object in front: depth edges around it
    z[rng.random((h, w)) < 0.05] = 0           # holes
    d16 = np.rint(z * 1000).astype(np.uint16)
    cam = dict(fx=80.0, fy=82.0, cx=41.3, cy=30.1)
    for img, scale in ((d16, 0.001), (z.astype(np.float32), 1.0)):
        gp, gn = ppf.depth_to_cloud(img, depth_scale=scale, z_min=0.3, z_max=5.0, max_jump=0.03, **cam)
        op, on = oracle.depth_to_cloud(img, depth_scale=scale, z_min=0.3, z_max=5.0, max_jump=0.03, **cam)
        assert 0 < len(gp) < h * w and gp.shape == op.shape
        assert np.array_equal(gp, op) and np.array_equal(gn, on)
        assert np.allclose(np.linalg.norm(gn, axis=1), 1, atol=1e-5) and np.all(np.einsum("ij,ij->i", gn, gp) <= 0)
    # plane a*x + b*y + c*z = d seen by the camera: every normal is -(a, b, c) normalised
    a, b, c, dd = 0.2, -0.1, 1.0, 2.0
    x, y = (u - cam["cx"]) / cam["fx"], (v - cam["cy"]) / cam["fy"]
    zp = (dd / (a * x + b * y + c)).astype(np.float32)
    gp, gn = ppf.depth_to_cloud(zp, depth_scale=1.0, z_min=0.3, z_max=9.0, max_jump=0.5, **cam)
    want = -np.float32([a, b, c]) / np.linalg.norm([a, b, c])
    assert len(gp) == (h - 2) * (w - 2) and np.abs(gn - want).max() < 2e-3
    with pytest.raises(ppf.OslamError):
        ppf.depth_to_cloud(d16[:2], **cam)


def test_depth_frame_to_pose(ppf, built_lib, synth):
    """The streaming chain on one frame: a model in front of a wall rendered into a 640x480 depth
    image -> depth_to_cloud -> voxel grid at leaf = d_dist -> PPF registration recovers the model's
    pose within the reference's own acceptance test (12 degrees, 0.1 diameters, alignment.cpp:141-144)."""
    mp, mn = synth.make_model(0, 1500)
    d = synth.d_dist_for(mp, 0.05)
    diam = synth.bbox_extent(mp)
    dense, _ = synth.make_model(0, 400000)
    rng = synth.SplitMix64(77)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = synth.random_rotation(rng)
    T[:3, 3] = [0.2, -0.1, 6.0]
    cam_pts = dense @ T[:3, :3].T + T[:3, 3]
    img = synth.render_depth(cam_pts, background_z=9.0, splat=1)
    sp, sn = ppf.depth_to_cloud(img, 525.0, 525.0, 319.5, 239.5, z_min=0.5, z_max=12.0, max_jump=0.08)
    assert len(sp) > 50000
    sp, sn = ppf.voxel_grid(sp, sn, leaf=d)
    mg, mgn = ppf.voxel_grid(mp, mn, leaf=d)
    mo = ppf.Model(mg, mgn, d_dist=d)
    Tm = mo.ppf_lookup(ppf.Scene(sp, sn, d_dist=d, ref_point_downsample_factor=2)).copy()
    cells_a, _ = mo.last_cells()
    dt, dr = ppf.ht_dist(Tm, T)
    assert dr < np.deg2rad(12) and dt < 0.1 * diam, (np.degrees(dr), dt / diam)
    # the same chain in one call, the full-resolution cloud staying in HBM: the same scene, cell for cell
    sc = ppf.Scene.from_depth(img, 525.0, 525.0, 319.5, 239.5, leaf=d, d_dist=d, ref_point_downsample_factor=2,
                              z_min=0.5, z_max=12.0, max_jump=0.08)
    assert sc.numPoints() == len(sp)
    Tc = mo.ppf_lookup(sc)
    assert np.array_equal(Tc, Tm) and cells_equal(mo.last_cells()[0], cells_a)
    with pytest.raises(ppf.OslamError):
        ppf.Scene.from_depth(np.zeros((480, 640), np.uint16), 525.0, 525.0, 319.5, 239.5, leaf=d)   # no valid pixel


@pytest.fixture
def pose_tail_on_device(monkeypatch):
    # the device pose tail (oslam_posegpu.hip) normally starts at 4096 peak records; run it from 2
    import importlib
    monkeypatch.setitem(importlib.import_module("objective-slam_amd").ppf.DEFAULT_OVERRIDES, "pose_gpu_min", 2)


def test_device_pose_tail_two_sorts_equals_oracle(ppf, oracle, built_lib, case_small, case_two_slices, pose_tail_on_device, monkeypatch):
    """The order of the kept cells (count descending, code ascending) normally comes from one sort of packed keys; clouds
    whose fields do not fit 64 bits take two stable sorts of (code, count) pairs instead -- forced here: same cells in the
    same order, same poses, same winner."""
    import importlib
    monkeypatch.setitem(importlib.import_module("objective-slam_amd").ppf.DEFAULT_OVERRIDES, "pose_two_sorts", 1)
    for c, df, flags in ((case_small, 1, {}), (case_two_slices, 10, {}), (case_small, 3, dict(use_l1_norm=True))):
        _align_and_compare(ppf, oracle, c, df=df, **flags)
        sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=df)
        mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], **flags)
        mo.ppf_lookup(sc)
        ocells, _ = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], df, c["d"], 0.4)
        cells, poses = mo.last_cells()
        assert cells_equal(cells, ocells)
        assert np.array_equal(poses, oracle.trans_calc2(ocells, c["mp"], c["mn"], c["sp"], c["sn"]))


def test_device_pose_tail_equals_oracle(ppf, oracle, built_lib, case_small, case_two_slices, synth, pose_tail_on_device):
    """Filter, order, K5..K9 and the winner on the GPU: kept cells (in order), every pose matrix and the
    returned pose equal the oracle's, as the host tail's do; also with the l1 flag, with point
    weights, with two model slices, through the sharded align_finish, and when only one cell survives
    (which the device path hands back to the host)."""
    for c, df, flags in ((case_small, 1, {}), (case_small, 3, dict(use_l1_norm=True)), (case_two_slices, 10, {})):
        _align_and_compare(ppf, oracle, c, df=df, **flags)
    c = case_small
    sc = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"])
    mo = ppf.Model(c["mp"], c["mn"], d_dist=c["d"])
    mo.ppf_lookup(sc)
    cells, poses = mo.last_cells()
    assert len(cells) > 50
    ocells, _ = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], 1, c["d"], 0.4)
    assert cells_equal(cells, ocells)
    assert np.array_equal(poses, oracle.trans_calc2(ocells, c["mp"], c["mn"], c["sp"], c["sn"]))
    # weights change the clustering scores (kernel.cu:777)
    w = np.linspace(0.1, 3.0, len(c["mp"])).astype(np.float32)
    mo.SetModelPointVoteWeights(w)
    Tw = mo.ppf_lookup(sc)
    _, Tow = oracle.pose_from_cells(ocells, c["mp"], c["mn"], c["sp"], c["sn"], c["d"], weights=w)   # kernel.cu:766-782
    assert np.array_equal(Tw, Tow)
    Th, _ = ppf.pose_stage(ocells, c["mp"], c["mn"], c["sp"], c["sn"], c["d"], weights=w)           # the host tail too
    assert np.array_equal(Th, Tow)
    # sharded: three local peak lists -> union -> align_finish on the device
    all_cells, gmax = [], 0
    for rank in range(3):
        par = ppf.default_params(shard_rank=rank, shard_world=3)
        scr = ppf.Scene(c["sp"], c["sn"], d_dist=c["d"], ref_point_downsample_factor=2, params=par)
        mor = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], params=par)
        cl, lmax = mor.align_local(scr, cap=1 << 16)
        all_cells.append(cl)
        gmax = max(gmax, lmax)
    T = mor.align_finish(scr, np.concatenate(all_cells), gmax)
    oc2, _ = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], 2, c["d"], 0.4)
    _, To = oracle.pose_from_cells(oc2, c["mp"], c["mn"], c["sp"], c["sn"], c["d"])
    assert cells_equal(mor.last_cells()[0], oc2) and np.array_equal(T, To)
    # a threshold that leaves one cell: the reference's kernels return early and the pose is zero
    one = ppf.Model(c["mp"], c["mn"], d_dist=c["d"], vote_count_threshold=0.999)
    T1 = one.ppf_lookup(sc)
    assert one.stats["num_top"] == 1 and np.all(T1 == 0)


def test_device_pose_tail_random_clouds(ppf, oracle, built_lib, pose_tail_on_device):
    rng = np.random.default_rng(977)
    for trial in range(10):
        M, S = int(rng.choice([17, 64, 333, 1100])), int(rng.choice([65, 257, 700, 1500]))
        mp = rng.uniform(-1, 1, (M, 3)).astype(np.float32)
        mn = (rng.normal(size=(M, 3)) * rng.uniform(0.2, 3.0, (M, 1))).astype(np.float32)
        sp = np.concatenate([mp[: min(M, S // 2)] + np.float32(0.3), rng.uniform(-2, 2, (S - min(M, S // 2), 3))]).astype(np.float32)
        sn = np.concatenate([mn[: min(M, S // 2)], rng.normal(size=(S - min(M, S // 2), 3))]).astype(np.float32)
        _align_and_compare(ppf, oracle, dict(mp=mp, mn=mn, sp=sp, sn=sn, d=float(rng.choice([0.1, 0.25]))),
                           df=int(rng.choice([1, 2, 5])))
