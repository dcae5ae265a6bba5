"""CPU: the C-ABI library loads, exports every symbol include/oslam.h declares, and its
host-side stage (poses, clustering, ht_dist) equals the oracle.  No compute call that needs
the GPU is made; on a machine without a HIP device those calls must fail loudly."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import make_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "oslam.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(oslam_[A-Za-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_lib, ppf):
    names = declared_symbols()
    assert len(names) >= 24
    raw = C.CDLL(ppf.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), n
    assert set(names) == set(ppf._SIGNATURES), set(names) ^ set(ppf._SIGNATURES)


def test_params_default_match_the_reference_cli(built_lib, ppf):
    p = ppf.default_params()
    # alignment.cpp:119-172
    assert p.ref_point_df == 1 and abs(p.vote_count_threshold - 0.4) < 1e-7
    assert (p.cpu_clustering, p.use_l1_norm, p.use_averaged_clusters) == (0, 0, 0)
    assert p.vote_mode == ppf.VOTE_EXACT and (p.shard_rank, p.shard_world) == (0, 1)


def test_d_dist_rule(built_lib, ppf, synth):
    mp, _ = synth.make_model(0, 300)
    assert ppf.d_dist_from_cloud(mp, 0.05) == synth.d_dist_for(mp, 0.05)


@pytest.mark.parametrize("flags", [dict(), dict(cpu_clustering=True), dict(use_l1_norm=True),
                                   dict(use_averaged_clusters=True)])
def test_host_stage_equals_oracle(built_lib, ppf, oracle, synth, flags):
    c = make_case(synth, 150, 320, 2020)
    cells, _ = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], 1, c["d"])
    rc, To = oracle.pose_from_cells(cells, c["mp"], c["mn"], c["sp"], c["sn"], c["d"], **flags)
    Tp, poses = ppf.pose_stage(cells, c["mp"], c["mn"], c["sp"], c["sn"], c["d"], **flags)
    assert np.array_equal(Tp, To)
    assert np.array_equal(poses, oracle.trans_calc2(cells, c["mp"], c["mn"], c["sp"], c["sn"]))
    assert ppf.ht_dist(Tp, c["truth"]) == oracle.ht_dist(Tp, c["truth"])
    # weights: all ones is the reference default (model.cu:67); halving them halves cluster scores only
    Tw, _ = ppf.pose_stage(cells, c["mp"], c["mn"], c["sp"], c["sn"], c["d"], weights=np.ones(len(c["mp"])), **flags)
    assert np.array_equal(Tw, Tp)


def test_host_stage_single_and_empty(built_lib, ppf, oracle, synth):
    c = make_case(synth, 60, 120, 2011)
    cells, _ = oracle.votes_fused(c["mp"], c["mn"], c["sp"], c["sn"], 1, c["d"])
    T, _ = ppf.pose_stage(cells[:1], c["mp"], c["mn"], c["sp"], c["sn"], c["d"])
    assert np.all(T == 0)                               # kernel.cu:609: one cell, no pose
    with pytest.raises(ppf.OslamError):
        ppf.pose_stage(cells[:0], c["mp"], c["mn"], c["sp"], c["sn"], c["d"])
    bad = cells[:2].copy()
    bad["code"][0] = (1 << 40)                          # scene index out of range
    with pytest.raises(ppf.OslamError):
        ppf.pose_stage(bad, c["mp"], c["mn"], c["sp"], c["sn"], c["d"])


def test_filter_and_sort(built_lib, ppf):
    cells = np.zeros(6, ppf.CELL_DTYPE)
    cells["code"] = [5, 3, 9, 1, 7, 2]
    cells["count"] = [10, 4, 10, 5, 3, 4]
    L = ppf.lib()
    n = L.oslam_filter_cells(cells.ctypes.data_as(C.c_void_p), 6, 0.4, 10)    # keep count > 4.0
    assert n == 3
    kept = cells[:n].copy()
    L.oslam_sort_cells(kept.ctypes.data_as(C.c_void_p), n)
    assert list(kept["code"]) == [5, 9, 1] and list(kept["count"]) == [10, 10, 5]


def test_no_cpu_fallback(built_lib, ppf, synth):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    mp, mn = synth.make_model(0, 50)
    with pytest.raises(ppf.OslamError) as e:
        ppf.Model(mp, mn, d_dist=0.1)
    assert e.value.code == ppf.OSLAM_E_DEVICE and "no CPU fallback" in str(e.value)


def test_cpp_adaptor_compiles_and_links(built_lib, ppf):
    """include/oslam_pcl.hpp (the reference's C++ signatures over the C-ABI) with plain structs
    shaped like pcl::PointNormal / Eigen::Matrix4f; PCL and Eigen are not in this image."""
    import subprocess
    out = os.path.join(ROOT, "build", "pcl_adaptor_check")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    libdir = os.path.dirname(ppf.LIB_PATH)
    subprocess.run(["g++", "-std=c++14", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "native", "pcl_adaptor_check.cpp"), "-o", out,
                    "-L", libdir, "-loslam_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    assert subprocess.run([out]).returncode == 0


def test_ply_reader_and_writer(built_lib, ppf, synth, tmp_path):
    """PLY I/O (pcl::io::loadPLYFile<PointNormal>, alignment.cpp:212,241): both formats round-trip
    bit-exactly in binary, to 9 significant digits in ascii; foreign headers (doubles, extra
    properties, nx/ny/nz names, a face element after the vertices) are understood."""
    p, n = synth.make_model(0, 300)
    for binary in (True, False):
        f = str(tmp_path / ("c%d.ply" % binary))
        ppf.ply_write(f, p, n, binary=binary)
        p2, n2 = ppf.ply_read(f)
        assert np.array_equal(p, p2) and np.array_equal(n, n2)
    f = str(tmp_path / "foreign.ply")
    with open(f, "w") as fh:
        fh.write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty double x\nproperty double y\n"
                 "property double z\nproperty uchar red\nproperty float nx\nproperty float ny\nproperty float nz\n"
                 "element face 1\nproperty list uchar int vertex_indices\nend_header\n"
                 "0.5 1 2 255 0 0 1\n-1 -2 -3 7 0 1 0\n3 0 1 0\n")
    p2, n2 = ppf.ply_read(f)
    assert np.array_equal(p2, np.float32([[0.5, 1, 2], [-1, -2, -3]])) and np.array_equal(n2, np.float32([[0, 0, 1], [0, 1, 0]]))
    import struct
    f = str(tmp_path / "bin.ply")
    with open(f, "wb") as fh:
        fh.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float x\nproperty float y\n"
                 b"property float z\nproperty float normal_x\nproperty float normal_y\nproperty float normal_z\n"
                 b"property float curvature\nend_header\n" + struct.pack("<7f", 1, 2, 3, 0, 0, 1, 9))
    p2, n2 = ppf.ply_read(f)
    assert np.array_equal(p2, np.float32([[1, 2, 3]])) and np.array_equal(n2, np.float32([[0, 0, 1]]))
    for bad in ("ply\nformat binary_big_endian 1.0\nelement vertex 0\nend_header\n",
                "ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nend_header\n0 0 0\n",
                "not a ply\n"):
        f = str(tmp_path / "bad.ply")
        open(f, "w").write(bad)
        with pytest.raises(ppf.OslamError):
            ppf.ply_read(f)
    with pytest.raises(ppf.OslamError):
        ppf.ply_read(str(tmp_path / "missing.ply"))


def test_model_file_rejections_need_no_gpu(built_lib, ppf, tmp_path):
    """oslam_model_load checks the file before it touches a device: a missing file, a foreign file and
    another layout version are OSLAM_E_INVALID on any machine."""
    import struct
    with pytest.raises(ppf.OslamError) as e:
        ppf.Model.load(str(tmp_path / "missing.oslam"))
    assert e.value.code == ppf.OSLAM_E_INVALID
    f = str(tmp_path / "foreign.oslam")
    open(f, "wb").write(b"ply\nformat ascii 1.0\n" + bytes(200))
    with pytest.raises(ppf.OslamError) as e:
        ppf.Model.load(f)
    assert e.value.code == ppf.OSLAM_E_INVALID and "not a model file" in str(e.value)
    f = str(tmp_path / "oldversion.oslam")
    open(f, "wb").write(struct.pack("<QII", 0x4c444d4f534c4f00, 1, 0) + bytes(200))
    with pytest.raises(ppf.OslamError) as e:
        ppf.Model.load(f)
    assert e.value.code == ppf.OSLAM_E_INVALID and "layout version" in str(e.value)


def test_depth_entry_points_check_arguments_first(built_lib, ppf):
    """oslam_depth_to_cloud / oslam_scene_from_depth validate the image and the camera before they look
    for a device; with good arguments and no GPU they report OSLAM_E_DEVICE (no CPU fallback)."""
    import torch
    img = np.full((8, 8), 1000, np.uint16)
    for bad in (dict(fx=0.0), dict(depth_scale=0.0), dict(z_min=0.0), dict(z_min=2.0, z_max=1.0), dict(max_jump=-1.0)):
        kw = dict(fx=10.0, fy=10.0, cx=4.0, cy=4.0, depth_scale=0.001, z_min=0.1, z_max=10.0, max_jump=0.05)
        kw.update(bad)
        with pytest.raises(ppf.OslamError) as e:
            ppf.depth_to_cloud(img, **kw)
        assert e.value.code == ppf.OSLAM_E_INVALID
    with pytest.raises(ppf.OslamError) as e:
        ppf.Scene.from_depth(img[:2], 10.0, 10.0, 4.0, 4.0, leaf=0.1)         # fewer than 3 rows
    assert e.value.code == ppf.OSLAM_E_INVALID
    with pytest.raises(ValueError):
        ppf.depth_to_cloud(img.astype(np.int32), 10.0, 10.0, 4.0, 4.0)
    if not torch.cuda.is_available():
        with pytest.raises(ppf.OslamError) as e:
            ppf.depth_to_cloud(img, 10.0, 10.0, 4.0, 4.0)
        assert e.value.code == ppf.OSLAM_E_DEVICE
