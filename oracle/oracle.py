"""ctypes loader for the CPU oracle (oracle/liboracle_ppf.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Cell(C.Structure):
    _fields_ = [("code", C.c_uint64), ("count", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("num_scene_ppfs", C.c_uint64), ("num_hits", C.c_uint64), ("num_votes", C.c_uint64),
                ("num_unique_votes", C.c_uint64), ("num_model_keys", C.c_uint64),
                ("max_count", C.c_uint32), ("num_top", C.c_uint64)]

    def asdict(self):
        return {f: int(getattr(self, f)) for f, _ in self._fields_}


CELL_DTYPE = np.dtype([("code", "<u8"), ("count", "<u4"), ("_pad", "<u4")])


def build():
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "liboracle_ppf.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle_ppf.c", "oracle_voxel.c", "oracle_depth.c", "oracle_matlab.c", "oracle_ppf.h")]
    if not os.path.exists(path) or any(os.path.exists(f) and os.path.getmtime(f) > os.path.getmtime(path) for f in srcs):
        build()
    L = C.CDLL(path)
    fp, u32p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
    L.orc_hash.restype = C.c_uint32
    L.orc_hash.argtypes = [vp, C.c_int, C.c_uint32]
    L.orc_d_angle0.restype = C.c_float
    L.orc_ppf_all_pairs.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, vp, vp]
    L.orc_ppf_row_keys.argtypes = [vp, vp, C.c_int, C.c_int, C.c_float, vp]
    L.orc_votes_literal.restype = C.c_void_p
    L.orc_votes_literal.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float,
                                    C.POINTER(C.c_size_t), C.POINTER(Stats), C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_size_t)]
    L.orc_votes_fused.restype = C.c_void_p
    L.orc_votes_fused.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float,
                                  C.c_long, C.c_long, C.c_long, C.c_int, C.POINTER(C.c_size_t),
                                  C.POINTER(Stats)]
    L.orm_voting_scheme.restype = C.c_uint64
    L.orm_voting_scheme.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, vp, vp, vp, vp, vp, vp]
    L.orm_d_dist.restype = C.c_double
    L.orm_d_dist.argtypes = [vp, C.c_int]
    L.orc_fused_create.restype = C.c_void_p
    L.orc_fused_create.argtypes = [vp, vp, C.c_int, C.c_float]
    L.orc_fused_free.argtypes = [vp]
    L.orc_fused_votes.restype = C.c_void_p
    L.orc_fused_votes.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_long, C.c_long,
                                  C.c_long, C.c_int, C.POINTER(C.c_size_t), C.POINTER(Stats)]
    L.orc_accumulator_for_ref.argtypes = [vp, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_float, vp]
    L.orc_trans_model_scene.restype = C.c_uint
    L.orc_trans_calc2.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, vp]
    L.orc_mat2transquat.argtypes = [vp, C.c_size_t, vp, vp]
    L.orc_trans2idx.argtypes = [vp, C.c_size_t, C.c_float, vp, vp]
    L.orc_cluster_gpu_style.restype = C.c_size_t
    L.orc_cluster_gpu_style.argtypes = [vp, C.c_size_t, vp, vp, C.c_float, C.c_int, C.c_int, vp]
    L.orc_cluster_poses_cpu.restype = C.c_int
    L.orc_cluster_poses_cpu.argtypes = [vp, vp, C.c_size_t, C.c_float, C.c_float, vp, vp]
    L.orc_ht_dist.argtypes = [vp, vp, vp]
    L.orc_voxel_grid.restype = C.c_long
    L.orc_voxel_grid.argtypes = [vp, vp, C.c_size_t, C.c_float, vp, vp]
    L.orc_depth_to_cloud.restype = C.c_long
    L.orc_depth_to_cloud.argtypes = [vp, C.c_int, C.c_int, C.c_int] + [C.c_float] * 8 + [vp, vp]
    L.orc_pose_from_cells_w.restype = C.c_int
    L.orc_pose_from_cells_w.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, C.c_int, vp, vp]
    L.orc_pose_from_cells.restype = C.c_int
    L.orc_pose_from_cells.argtypes = [vp, C.c_size_t, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int,
                                      C.c_int, vp]
    _LIB = L
    return L


class F3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]


def _f3(a):
    a = np.asarray(a, np.float32)
    return F3(float(a[0]), float(a[1]), float(a[2]))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _c32(a):
    a = np.ascontiguousarray(a, np.float32)
    assert a.ndim == 2 and a.shape[1] == 3
    return a


def hash_bytes(data: bytes, seed=2166136261):
    buf = C.create_string_buffer(data, len(data))
    return int(lib().orc_hash(C.cast(buf, C.c_void_p), len(data), seed))


def d_angle0():
    return float(lib().orc_d_angle0())


def ppf_all_pairs(points, normals, df, d_dist, want_ppf=True):
    p, n = _c32(points), _c32(normals)
    N = len(p)
    keys = np.zeros((N, N), np.uint32)
    ppf = np.zeros((N, N, 4), np.float32) if want_ppf else None
    lib().orc_ppf_all_pairs(_p(p), _p(n), N, int(df), float(d_dist), _p(ppf) if want_ppf else None, _p(keys))
    return ppf, keys


def ppf_row_keys(points, normals, ref, d_dist):
    p, n = _c32(points), _c32(normals)
    keys = np.zeros(len(p), np.uint32)
    lib().orc_ppf_row_keys(_p(p), _p(n), len(p), int(ref), float(d_dist), _p(keys))
    return keys


def _cells_from(ptr, n):
    if not ptr or n == 0:
        out = np.zeros(0, CELL_DTYPE)
    else:
        buf = (C.c_char * (16 * n)).from_address(ptr)
        out = np.frombuffer(buf, dtype=CELL_DTYPE).copy()
    if ptr:
        C.CDLL(None).free(C.c_void_p(ptr))
    return out


def votes_literal(mp, mn, sp, sn, df, d_dist, thresh=0.4, want_all=False):
    mp, mn, sp, sn = _c32(mp), _c32(mn), _c32(sp), _c32(sn)
    n_out = C.c_size_t(0)
    st = Stats()
    allp, alln = C.c_void_p(0), C.c_size_t(0)
    ptr = lib().orc_votes_literal(_p(mp), _p(mn), len(mp), _p(sp), _p(sn), len(sp), int(df), float(d_dist),
                                  float(thresh), C.byref(n_out), C.byref(st),
                                  C.byref(allp) if want_all else None, C.byref(alln) if want_all else None)
    # the returned buffer holds all unique cells sorted; the first n_out are kept
    cells = _cells_from(ptr, int(st.num_unique_votes))[: n_out.value]
    if want_all:
        return cells, st.asdict(), _cells_from(allp.value, alln.value)
    return cells, st.asdict()


def votes_fused(mp, mn, sp, sn, df, d_dist, thresh=0.4, ref_begin=0, ref_step=1, ref_limit=-1, threads=0):
    mp, mn, sp, sn = _c32(mp), _c32(mn), _c32(sp), _c32(sn)
    n_out = C.c_size_t(0)
    st = Stats()
    ptr = lib().orc_votes_fused(_p(mp), _p(mn), len(mp), _p(sp), _p(sn), len(sp), int(df), float(d_dist),
                                float(thresh), int(ref_begin), int(ref_step), int(ref_limit), int(threads),
                                C.byref(n_out), C.byref(st))
    cells = _cells_from(ptr, n_out.value)
    return cells, st.asdict()


class FusedModel:
    """Model table built once on the CPU; votes() can then be timed on its own."""

    def __init__(self, mp, mn, d_dist):
        self.mp, self.mn = _c32(mp), _c32(mn)
        self.d_dist = float(d_dist)
        self.h = lib().orc_fused_create(_p(self.mp), _p(self.mn), len(self.mp), self.d_dist)

    def votes(self, sp, sn, df, thresh=0.4, ref_begin=0, ref_step=1, ref_limit=-1, threads=0):
        sp, sn = _c32(sp), _c32(sn)
        n_out = C.c_size_t(0)
        st = Stats()
        ptr = lib().orc_fused_votes(self.h, _p(sp), _p(sn), len(sp), int(df), self.d_dist, float(thresh),
                                    int(ref_begin), int(ref_step), int(ref_limit), int(threads),
                                    C.byref(n_out), C.byref(st))
        return _cells_from(ptr, n_out.value), st.asdict()

    def close(self):
        if self.h:
            lib().orc_fused_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def accumulator_for_ref(mp, mn, sp, sn, s_r, d_dist):
    mp, mn, sp, sn = _c32(mp), _c32(mn), _c32(sp), _c32(sn)
    acc = np.zeros((len(mp), 32), np.uint32)
    lib().orc_accumulator_for_ref(_p(mp), _p(mn), len(mp), _p(sp), _p(sn), len(sp), int(s_r), float(d_dist), _p(acc))
    return acc


def trans_model_scene(m_r, n_r_m, m_i, s_r, n_r_s, s_i, want_T=False):
    L = lib()
    f = L.orc_trans_model_scene
    f.argtypes = [F3, F3, F3, F3, F3, F3, C.c_void_p]
    T = np.zeros(16, np.float32)
    idx = f(_f3(m_r), _f3(n_r_m), _f3(m_i), _f3(s_r), _f3(n_r_s), _f3(s_i), _p(T) if want_T else None)
    return (int(idx), T.reshape(4, 4)) if want_T else int(idx)


def trans_calc2(cells, mp, mn, sp, sn):
    mp, mn, sp, sn = _c32(mp), _c32(mn), _c32(sp), _c32(sn)
    cells = np.ascontiguousarray(cells, CELL_DTYPE)
    T = np.zeros((len(cells), 16), np.float32)
    lib().orc_trans_calc2(_p(cells), len(cells), _p(mp), _p(mn), _p(sp), _p(sn), _p(T))
    return T


def pose_from_cells(cells, mp, mn, sp, sn, d_dist, cpu_clustering=False, use_l1_norm=False,
                    use_averaged_clusters=False, weights=None):
    """weights: Model::SetModelPointVoteWeights (model.cu:84-93); None = all 1 (model.cu:67)."""
    mp, mn, sp, sn = _c32(mp), _c32(mn), _c32(sp), _c32(sn)
    cells = np.ascontiguousarray(cells, CELL_DTYPE)
    T = np.zeros(16, np.float32)
    w = None if weights is None else np.ascontiguousarray(weights, np.float32)
    rc = lib().orc_pose_from_cells_w(_p(cells), len(cells), _p(mp), _p(mn), _p(sp), _p(sn), float(d_dist),
                                     int(cpu_clustering), int(use_l1_norm), int(use_averaged_clusters),
                                     _p(w) if w is not None else None, _p(T))
    return rc, T.reshape(4, 4)


def cluster_gpu_style(cells, trans, quat, d_dist, use_l1_norm=False, use_averaged_clusters=False):
    cells = np.ascontiguousarray(cells, CELL_DTYPE)
    trans = np.ascontiguousarray(trans, np.float32).copy()
    quat = np.ascontiguousarray(quat, np.float32)
    scores = np.zeros(len(cells), np.float32)
    best = lib().orc_cluster_gpu_style(_p(cells), len(cells), _p(trans), _p(quat), float(d_dist),
                                       int(use_l1_norm), int(use_averaged_clusters), _p(scores))
    return int(best), scores, trans


def mat2transquat(T):
    T = np.ascontiguousarray(T, np.float32).reshape(-1, 16)
    tr = np.zeros((len(T), 3), np.float32)
    qu = np.zeros((len(T), 4), np.float32)
    lib().orc_mat2transquat(_p(T), len(T), _p(tr), _p(qu))
    return tr, qu


def voxel_grid(points, normals, leaf):
    """pcl::VoxelGrid as the reference applies it (alignment.cpp:79-87); parity unpinned."""
    p, n = _c32(points), _c32(normals)
    po, no = np.zeros_like(p), np.zeros_like(n)
    k = lib().orc_voxel_grid(_p(p), _p(n), len(p), float(leaf), _p(po), _p(no))
    if k < 0:
        raise ValueError("leaf size too small for the cloud extent")
    return po[:k].copy(), no[:k].copy()


def depth_to_cloud(depth, fx, fy, cx, cy, depth_scale=0.001, z_min=0.1, z_max=10.0, max_jump=0.05):
    """Depth image -> (points, normals): this build's own front-end specification (oracle_depth.c);
    the reference has no such step, so there is nothing to pin it on."""
    d = np.ascontiguousarray(depth)
    assert d.dtype in (np.uint16, np.float32) and d.ndim == 2
    po, no = np.zeros((d.size, 3), np.float32), np.zeros((d.size, 3), np.float32)
    k = lib().orc_depth_to_cloud(_p(d), int(d.dtype == np.uint16), d.shape[1], d.shape[0], fx, fy, cx, cy,
                                 depth_scale, z_min, z_max, max_jump, _p(po), _p(no))
    return po[:k].copy(), no[:k].copy()


def ht_dist(A, B):
    A = np.ascontiguousarray(A, np.float32).reshape(16)
    B = np.ascontiguousarray(B, np.float32).reshape(16)
    out = np.zeros(2, np.float32)
    lib().orc_ht_dist(_p(A), _p(B), _p(out))
    return float(out[0]), float(out[1])


def align(mp, mn, sp, sn, df, d_dist, thresh=0.4, cpu_clustering=False, use_l1_norm=False,
          use_averaged_clusters=False, fused=True, threads=0):
    """Whole path on the CPU: cells -> pose.  Returns (T 4x4, cells, stats)."""
    if fused:
        cells, st = votes_fused(mp, mn, sp, sn, df, d_dist, thresh, threads=threads)
    else:
        cells, st = votes_literal(mp, mn, sp, sn, df, d_dist, thresh)
    rc, T = pose_from_cells(cells, mp, mn, sp, sn, d_dist, cpu_clustering, use_l1_norm, use_averaged_clusters)
    st["rc"] = rc
    return T, cells, st


# ---- the MATLAB prototype in double precision (oracle_matlab.c; matlab/voting_scheme.m and friends) ----
def matlab_d_dist(points):
    """model_description.m:6-15: a tenth of the largest distance from the bounding box's centre."""
    p = np.ascontiguousarray(points, np.float64)
    return float(lib().orm_d_dist(_p(p), len(p)))


def matlab_voting_scheme(mp, mn, sp, sn, skip, d_dist, eps_bins=1e-4):
    """voting_scheme.m in double precision (the dictionary key = the four bin indices).  Returns a dict:
    acc [n_ref, M, 30], argmax_row / argmax_col / max_tots [n_ref] (0-based, -1 = no vote), selected [n_ref]
    (> 0.9 of the largest maximum), edge_votes [n_ref] (votes within eps_bins of a bin boundary on their
    way), votes (total)."""
    mp, mn = np.ascontiguousarray(mp, np.float64), np.ascontiguousarray(mn, np.float64)
    sp, sn = np.ascontiguousarray(sp, np.float64), np.ascontiguousarray(sn, np.float64)
    M, S = len(mp), len(sp)
    n_ref = (S + skip - 1) // skip
    acc = np.zeros((n_ref, M, 30), np.uint32)
    row, col = np.zeros(n_ref, np.int32), np.zeros(n_ref, np.int32)
    mx, sel, ev = np.zeros(n_ref, np.uint32), np.zeros(n_ref, np.uint8), np.zeros(n_ref, np.uint64)
    votes = lib().orm_voting_scheme(_p(mp), _p(mn), M, _p(sp), _p(sn), S, int(skip), float(d_dist), float(eps_bins),
                                    _p(acc), _p(row), _p(col), _p(mx), _p(sel), _p(ev))
    if votes == 2 ** 64 - 1:
        raise MemoryError("orm_voting_scheme")
    return dict(acc=acc, argmax_row=row, argmax_col=col, max_tots=mx, selected=sel.astype(bool), edge_votes=ev, votes=int(votes))


def matlab_argmax(acc2d):
    """[Y_rows, I_row] = max(A); [max_tot, I_col] = max(Y_rows) (voting_scheme.m:83-88) on one [M, 30] slice:
    (row, col, max) with MATLAB's first-maximum rule, (-1, -1, 0) for an empty slice."""
    a = np.asarray(acc2d)
    colmax = a.max(axis=0)
    if colmax.max() == 0:
        return -1, -1, 0
    c = int(np.argmax(colmax))                # first column with the largest maximum
    r = int(np.argmax(a[:, c]))               # first row of that column
    return r, c, int(colmax[c])
