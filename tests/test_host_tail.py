"""CPU: the host pose tail at sizes where its loops matter.  The greedy clustering of
transformation_clustering.cpp:62-122 (cpu_clustering) and the in-place translation averaging of
kernel.cu:747-758 (use_averaged_clusters) are sequential by definition; oslam_pose.c runs them as do-across loops
over a grid / a cell-hash table on several host threads.  Whatever the thread count, the pose must be the one
the oracle's plain sequential loops give (O(poses x clusters) there)."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def many_cells(synth, oracle):
    mp, mn = synth.make_model(0, 400)
    d = synth.d_dist_for(mp, 0.05)
    sp, sn, _ = synth.make_scene([0], 2500, 2002, instance_points=400, noise_sigma=0.1 * d)
    cells, st = oracle.votes_fused(mp, mn, sp, sn, 1, d, 0.12)
    assert len(cells) > 3000
    return dict(mp=mp, mn=mn, sp=sp, sn=sn, d=d, cells=cells)


@pytest.mark.parametrize("flags", [dict(cpu_clustering=True), dict(use_averaged_clusters=True), dict(),
                                   dict(use_averaged_clusters=True, use_l1_norm=True)])
def test_host_tail_equals_the_sequential_statement_on_any_thread_count(ppf, oracle, built_lib, many_cells, flags):
    c = many_cells
    rc, To = oracle.pose_from_cells(c["cells"], c["mp"], c["mn"], c["sp"], c["sn"], c["d"], **flags)
    try:
        for threads in (1, 2, 5):
            ppf.set_host_threads(threads)
            T, poses = ppf.pose_stage(c["cells"], c["mp"], c["mn"], c["sp"], c["sn"], c["d"], **flags)
            assert np.array_equal(T, To), (flags, threads)
    finally:
        ppf.set_host_threads(0)


def test_greedy_clustering_in_crowded_and_empty_neighbourhoods(ppf, oracle, built_lib, many_cells):
    """The same cells with every translation threshold from "everything is one cluster" to "every pose is its own":
    d_dist scales the grid of the greedy clustering (trans_thresh = d_dist, model.cu:262-263)."""
    c = many_cells
    ppf.set_host_threads(3)
    try:
        for scale in (0.05, 0.5, 4.0, 60.0):
            d = c["d"] * scale
            rc, To = oracle.pose_from_cells(c["cells"][:1500], c["mp"], c["mn"], c["sp"], c["sn"], d, cpu_clustering=True)
            T, _ = ppf.pose_stage(c["cells"][:1500], c["mp"], c["mn"], c["sp"], c["sn"], d, cpu_clustering=True)
            assert np.array_equal(T, To), scale
    finally:
        ppf.set_host_threads(0)
