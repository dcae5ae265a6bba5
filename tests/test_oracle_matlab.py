"""CPU: SURVEY.md 8 row a23 -- the reference's MATLAB prototype (matlab/drost.m: model_description.m,
point_pair_feature.m, my_discretize.m, voting_scheme.m, trans_model_scene.m) restated in double precision
(oracle/oracle_matlab.c; it cannot run here and the reference holds no output of it: parity unpinned, see
that file's header) against the single-precision restatement of the CUDA path (oracle/oracle_ppf.c):

  - the two accumulators [reference point][model point][alpha] are equal except at bin-edge cases -- a
    feature or alpha within 1e-4 bin of a boundary, where float and double may land in different bins --
    and those are counted: the L1 difference of every slice stays within twice their number;
  - MATLAB's selection (per reference point the first maximum in column order, then > 0.9 of the largest)
    is a different rule from the CUDA path's (every cell > 0.4 of the global maximum); where the margin
    exceeds the counted edge cases the per-reference argmax is the same in both precisions, and the cell
    MATLAB selects gives a pose inside the reference's acceptance test.
The GPU side of this row is tests/test_gpu_configs.py::test_matlab_argmax_set_on_device."""
import importlib

import numpy as np
import pytest

synth = importlib.import_module("objective-slam_amd.synth")


def fold_last_bin(acc32):
    """CUDA accumulator [M, 32] (bins 0..30, kernel.cu:341) -> MATLAB's 30 columns: alpha + pi == 2 pi is put
    into the last bin by min(round(.) + 1, n_angle) (voting_scheme.m:74)."""
    a = acc32[:, :30].astype(np.int64)
    a[:, 29] += acc32[:, 30]
    return a


def matlab_case(M, S, seed, noise):
    mp, mn = synth.make_model(0, M)
    d = synth.d_dist_for(mp, 0.05)
    sp, sn, poses = synth.make_scene([0], S, seed, instance_points=M, noise_sigma=noise * d)
    return mp, mn, sp, sn, d, poses[0][1]


@pytest.mark.parametrize("M,S,seed,noise", [(150, 360, 2071, 0.02), (300, 700, 2072, 0.1)])
def test_double_precision_prototype_equals_float_path_up_to_counted_edge_cases(oracle, M, S, seed, noise):
    mp, mn, sp, sn, d, truth = matlab_case(M, S, seed, noise)
    skip = 5                                                    # voting_scheme.m:11
    R = oracle.matlab_voting_scheme(mp, mn, sp, sn, skip, d)
    assert R["acc"].shape == ((S + skip - 1) // skip, M, 30)
    votes32 = l1_total = exact_slices = decided = 0
    for t, r in enumerate(range(0, S, skip)):
        f = fold_last_bin(oracle.accumulator_for_ref(mp, mn, sp, sn, r, d))
        votes32 += int(f.sum())
        l1 = int(np.abs(f - R["acc"][t].astype(np.int64)).sum())
        l1_total += l1
        assert l1 <= 2 * int(R["edge_votes"][t]), (t, l1, int(R["edge_votes"][t]))
        if R["edge_votes"][t] == 0:
            exact_slices += 1                                   # no edge case at all: the slices are identical (l1 == 0)
        top2 = np.sort(f.ravel())[-2:]
        if top2[1] - top2[0] > 2 * int(R["edge_votes"][t]):     # the maximum cannot change hands
            decided += 1
            assert oracle.matlab_argmax(f) == (R["argmax_row"][t], R["argmax_col"][t], R["max_tots"][t]), t
    # edge cases are rare: well under 2 % of the votes; the totals agree to the same bound
    assert int(R["edge_votes"].sum()) < 0.02 * R["votes"]
    assert abs(votes32 - R["votes"]) <= int(R["edge_votes"].sum())
    assert l1_total <= 0.001 * R["votes"] + 8
    assert decided >= 10
    # selection rule of voting_scheme.m:90-92
    best = int(R["max_tots"].max())
    assert np.array_equal(R["selected"], R["max_tots"].astype(np.float64) / best > 0.9)
    assert R["selected"].any()


def test_selected_cell_of_the_prototype_gives_the_pose(oracle):
    """drost.m:80-82 (the intended expression, drost.m:81): invht(T_s_g) * rotx(alpha) * T_m_g of the selected
    cells.  The cell MATLAB selects -- reference point, model point, alpha bin -- taken through the CUDA path's
    pose computation (K5, kernel.cu:352-401, lower bin edge) lands inside the reference's acceptance test
    (12 degrees, 0.1 diameters; alignment.cpp:141-144) of the instance's true pose."""
    ppf = importlib.import_module("objective-slam_amd.ppf")
    mp, mn, sp, sn, d, truth = matlab_case(300, 700, 2072, 0.1)
    R = oracle.matlab_voting_scheme(mp, mn, sp, sn, 5, d)
    ok = 0
    for t in np.flatnonzero(R["selected"]):
        cell = np.zeros(2, oracle.CELL_DTYPE)                   # K5 returns early for a single cell (kernel.cu:609)
        cell["code"] = (int(5 * t) << 32) | (int(R["argmax_row"][t]) << 6) | int(R["argmax_col"][t])
        cell["count"] = int(R["max_tots"][t])
        T = oracle.trans_calc2(cell, mp, mn, sp, sn)[0].reshape(4, 4)
        dt, dr = oracle.ht_dist(T, truth)
        ok += int(dr < np.deg2rad(12.0) and dt < 0.1 * synth.bbox_extent(mp))
    assert ok >= 1 and ok >= 0.5 * int(R["selected"].sum())


def test_d_dist_rule_of_the_prototype(oracle):
    """model_description.m:6-15: d_dist = 0.1 x the largest distance from the bounding box's centre (the CLI of
    the CUDA path uses tau_d x the largest bounding-box extent instead, alignment.cpp:246-253)."""
    pts = np.float64([[0, 0, 0], [2, 0, 0], [2, 4, 0], [0, 4, 6]])
    c = np.float64([1, 2, 3])
    assert np.isclose(oracle.matlab_d_dist(pts), 0.1 * np.linalg.norm(pts - c, axis=1).max())
