"""Python mirror of the reference's registration interface over the C-ABI.

The reference exposes (pcl/alignment/include/): ``ppf_registration(...)``
(ppf.h:9-15), ``class Scene`` (scene.h:10-52) and ``class Model`` (model.h:14-115)
with ``Model::ppf_lookup(Scene*)``.  The same names, argument meaning and result
fields live here, on top of ``liboslam_hip.so`` (include/oslam.h) through ctypes.
Clouds are numpy arrays: points ``[n,3]`` float32 and normals ``[n,3]`` float32,
or one ``[n,12]`` float32 array laid out like ``pcl::PointNormal`` (48 bytes).

There is no CPU fallback: if the shared library is missing, or no HIP device is
present, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OSLAM_LIB", os.path.join(_HERE, "liboslam_hip.so"))   # OSLAM_LIB: A/B builds
_LIB = None

OSLAM_OK, OSLAM_E_INVALID, OSLAM_E_DEVICE, OSLAM_E_NOMEM, OSLAM_E_NO_VOTES, OSLAM_E_LIMIT, OSLAM_E_PEER = range(7)
STAGE_VOTE, STAGE_SELECT, STAGE_GROW = 1, 2, 3       # oslam_comm_inject_failure
VOTE_EXACT, VOTE_FAST = 0, 1
COMM_ID_BYTES = 128


class OslamError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("oslam error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("ref_point_df", C.c_uint), ("vote_count_threshold", C.c_float),
                ("cpu_clustering", C.c_int), ("use_l1_norm", C.c_int), ("use_averaged_clusters", C.c_int),
                ("dev", C.c_int), ("vote_mode", C.c_int), ("shard_rank", C.c_int), ("shard_world", C.c_int),
                ("max_cells", C.c_uint), ("pose_gpu_min", C.c_uint), ("no_bucket_spread", C.c_int),
                ("scratch_gib", C.c_uint), ("pose_two_sorts", C.c_int), ("reserved", C.c_int * 2)]


class Stats(C.Structure):
    _fields_ = [("num_scene_ppfs", C.c_uint64), ("num_hits", C.c_uint64), ("num_votes", C.c_uint64),
                ("num_unique_votes", C.c_uint64), ("num_model_keys", C.c_uint64), ("num_top", C.c_uint64),
                ("max_count", C.c_uint32), ("num_emitted", C.c_uint32), ("ms_vote", C.c_float),
                ("ms_total", C.c_float), ("vote_launches", C.c_uint32), ("ms_vote_kernel", C.c_float),
                ("ms_key_kernel", C.c_float), ("wide_workgroups", C.c_uint32), ("num_pairs_probed", C.c_uint64),
                ("scratch_bytes", C.c_uint64), ("num_entries_streamed", C.c_uint64), ("num_items", C.c_uint64)]

    def asdict(self):
        return {f: getattr(self, f) for f, _ in self._fields_}


CELL_DTYPE = np.dtype([("code", "<u8"), ("count", "<u4"), ("pad", "<u4")])

# every function include/oslam.h declares: (name, restype, argtypes)
_vp, _sz, _f, _i, _u = C.c_void_p, C.c_size_t, C.c_float, C.c_int, C.c_uint
_SIGNATURES = {
    "oslam_params_default": (_i, [C.POINTER(Params)]),
    "oslam_d_dist_from_cloud": (_i, [_vp, _sz, _sz, _f, C.POINTER(_f)]),
    "oslam_model_create": (_i, [_vp, _vp, _sz, _sz, _f, C.POINTER(Params), C.POINTER(_vp)]),
    "oslam_model_destroy": (None, [_vp]),
    "oslam_model_set_point_weights": (_i, [_vp, _vp, _sz]),
    "oslam_model_save": (_i, [_vp, C.c_char_p]),
    "oslam_model_load": (_i, [C.c_char_p, C.POINTER(Params), C.POINTER(_vp)]),
    "oslam_model_info": (_i, [_vp, C.POINTER(_sz), C.POINTER(_f), C.POINTER(C.c_uint64)]),
    "oslam_scene_create": (_i, [_vp, _vp, _sz, _sz, _f, _u, C.POINTER(Params), C.POINTER(_vp)]),
    "oslam_scene_destroy": (None, [_vp]),
    "oslam_align": (_i, [_vp, _vp, _vp, C.POINTER(Stats)]),
    "oslam_align_prepare": (_i, [_vp, _vp]),
    "oslam_ppf_registration": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp, _sz, _sz, _vp, _u, _f, _i, _i, _i, _i, _vp, _vp]),
    "oslam_ht_dist": (_i, [_vp, _vp, _vp]),
    "oslam_ply_read": (_i, [C.c_char_p, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_sz)]),
    "oslam_ply_write": (_i, [C.c_char_p, _vp, _vp, _sz, _i]),
    "oslam_free": (None, [_vp]),
    "oslam_depth_to_cloud": (_i, [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _sz, C.POINTER(_sz)]),
    "oslam_scene_from_depth": (_i, [_vp, _i, _i, _i, _vp, _f, _f, _u, C.POINTER(Params), C.POINTER(_vp), C.POINTER(_sz)]),
    "oslam_voxel_grid": (_i, [_vp, _vp, _sz, _sz, _f, _i, _vp, _vp, _sz, C.POINTER(_sz)]),
    "oslam_build_T_g": (None, [_vp, _vp, _vp]),
    "oslam_sort_cells": (None, [_vp, _sz]),
    "oslam_filter_cells": (_sz, [_vp, _sz, _f, C.c_uint32]),
    "oslam_pose_stage": (_i, [_vp, _sz, _vp, _vp, _sz, _vp, _vp, _sz, _f, _i, _i, _i, _vp, _vp, _vp]),
    "oslam_align_local": (_i, [_vp, _vp, _vp, _sz, C.POINTER(_sz), C.POINTER(C.c_uint32), C.POINTER(Stats)]),
    "oslam_align_finish": (_i, [_vp, _vp, _vp, _sz, C.c_uint32, _vp, C.POINTER(Stats)]),
    "oslam_local_peaks": (_i, [_vp, C.c_uint32, _vp, _sz, C.POINTER(_sz)]),
    "oslam_last_result": (_i, [_vp, _vp, _vp, _vp, _vp, _sz, C.POINTER(_sz), C.POINTER(C.c_uint32)]),
    "oslam_pose_stage_ex": (_i, [_vp, _sz, _vp, _vp, _sz, _vp, _vp, _sz, _f, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "oslam_comm_unique_id": (_i, [_vp]),
    "oslam_comm_create": (_i, [_vp, _i, _i, _i, C.POINTER(_vp)]),
    "oslam_comm_destroy": (None, [_vp]),
    "oslam_align_multi": (_i, [_vp, _vp, _vp, _vp, C.POINTER(Stats)]),
    "oslam_comm_create_loopback": (_i, [_i, _i, C.POINTER(_vp)]),
    "oslam_comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "oslam_comm_inject_failure": (_i, [_vp, _i]),
    "oslam_comm_abort": (_i, [_vp]),
    "oslam_db_align_multi": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    "oslam_db_destroy_with_models": (None, [_vp]),
    "oslam_release_scratch": (_i, [_i]),
    "oslam_db_create": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "oslam_db_destroy": (None, [_vp]),
    "oslam_db_align": (_i, [_vp, _vp, _vp, _vp]),
    "oslam_db_size": (_i, [_vp, C.POINTER(_sz), C.POINTER(_sz)]),
    "oslam_scene_keys": (_i, [_vp, _sz, _vp]),
    "oslam_model_keys": (_i, [_vp, _sz, _vp]),
    "oslam_model_bucket": (_i, [_vp, C.c_uint32, _vp, _sz, C.POINTER(_sz)]),
    "oslam_model_bucket_words": (_i, [_vp, C.c_uint32, _i, _vp, _sz, C.POINTER(_sz)]),
    "oslam_vote_accumulator": (_i, [_vp, _vp, _sz, _vp]),
    "oslam_last_cells": (_i, [_vp, _vp, _vp, _sz, C.POINTER(_sz)]),
    "oslam_set_stream": (_i, [_vp]),
    "oslam_last_error": (C.c_char_p, []),
    "oslam_set_host_threads": (_i, [_i]),
    "oslam_selftest_math": (_i, [_sz, C.c_uint64, C.POINTER(C.c_uint64)]),
}


def lib():
    """Load liboslam_hip.so (built in-tree by __graft_entry__.build()); raise if absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise OslamError(OSLAM_E_DEVICE, "%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                             "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _LIB = L
    return _LIB


def _check(rc):
    if rc != OSLAM_OK:
        raise OslamError(rc, lib().oslam_last_error().decode("utf-8", "replace"))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


# Fields every default_params() call starts from, on top of the library's defaults: how tests and tools steer
# library-wide switches (pose_gpu_min, scratch_gib ...) -- through oslam_params, not through the environment.
DEFAULT_OVERRIDES = {}


def default_params(**kw):
    p = Params()
    _check(lib().oslam_params_default(C.byref(p)))
    for k, v in {**DEFAULT_OVERRIDES, **kw}.items():
        if not hasattr(p, k):
            raise TypeError("unknown parameter %r" % k)
        setattr(p, k, v)
    return p


def _cloud_args(points, normals=None):
    """-> (xyz pointer holder, nrm pointer, n, stride, keepalive array(s))"""
    pts = np.asarray(points)
    if normals is None:
        a = np.ascontiguousarray(pts, np.float32)
        if a.ndim != 2 or a.shape[1] != 12:
            raise ValueError("a single cloud array must be [n,12] float32 (pcl::PointNormal layout)")
        base = a.ctypes.data
        return C.c_void_p(base), C.c_void_p(base + 16), len(a), 48, (a,)
    p = np.ascontiguousarray(pts, np.float32)
    n = np.ascontiguousarray(normals, np.float32)
    if p.ndim != 2 or p.shape[1] != 3 or n.shape != p.shape:
        raise ValueError("points and normals must both be [n,3]")
    return _p(p), _p(n), len(p), 12, (p, n)


def d_dist_from_cloud(points, tau_d):
    """d_dist = tau_d * max bbox extent (alignment.cpp:246-253)."""
    p = np.ascontiguousarray(points, np.float32)
    out = C.c_float(0)
    _check(lib().oslam_d_dist_from_cloud(_p(p), len(p), p.strides[0], float(tau_d), C.byref(out)))
    return out.value


class Scene:
    """Scene(cloud, d_dist, ref_point_downsample_factor=1)  (scene.h:15-16)."""

    def __init__(self, points, normals=None, d_dist=None, ref_point_downsample_factor=1, params=None):
        if d_dist is None:
            raise TypeError("d_dist is required")
        self._h = C.c_void_p(0)
        xyz, nrm, n, stride, keep = _cloud_args(points, normals)
        self.params = params if params is not None else default_params()
        self.d_dist = float(d_dist)
        self.df = int(ref_point_downsample_factor)
        self.n = n
        _check(lib().oslam_scene_create(xyz, nrm, n, stride, self.d_dist, self.df, C.byref(self.params), C.byref(self._h)))

    @classmethod
    def from_depth(cls, depth, fx, fy, cx, cy, leaf, d_dist=0.0, ref_point_downsample_factor=1, depth_scale=0.001,
                   z_min=0.1, z_max=10.0, max_jump=0.05, params=None):
        """Depth frame -> points + normals -> voxel grid -> Scene in one call, the full-resolution cloud
        staying in HBM (oslam_scene_from_depth)."""
        d = np.ascontiguousarray(depth)
        if d.dtype not in (np.uint16, np.float32) or d.ndim != 2:
            raise ValueError("depth must be a 2-D uint16 or float32 image")
        self = cls.__new__(cls)
        self._h = C.c_void_p(0)
        self.params = params if params is not None else default_params()
        self.d_dist, self.df = float(d_dist), int(ref_point_downsample_factor)
        cam = Camera(fx, fy, cx, cy, depth_scale, z_min, z_max, max_jump)
        n = C.c_size_t(0)
        _check(lib().oslam_scene_from_depth(_p(d), int(d.dtype == np.uint16), d.shape[1], d.shape[0], C.byref(cam),
                                            float(leaf), self.d_dist, self.df, C.byref(self.params), C.byref(self._h),
                                            C.byref(n)))
        self.n = n.value
        return self

    def numPoints(self):
        return self.n

    def getHashKeys(self, ref_index):
        """Row `ref_index` of the reference's N x N hashKeys array (scene.cu:49-54)."""
        out = np.zeros(self.n, np.uint32)
        _check(lib().oslam_scene_keys(self._h, int(ref_index), _p(out)))
        return out

    def close(self):
        if self._h:
            lib().oslam_scene_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Model:
    """Model(cloud, d_dist, vote_count_threshold, cpu_clustering, use_l1_norm,
    use_averaged_clusters)  (model.h:17-19); results appear in the same fields
    after ppf_lookup: transformations, vote_counts, max_idx / best_T."""

    def __init__(self, points, normals=None, d_dist=None, vote_count_threshold=0.4, cpu_clustering=False,
                 use_l1_norm=False, use_averaged_clusters=False, params=None):
        if d_dist is None:
            raise TypeError("d_dist is required")
        self._h = C.c_void_p(0)
        xyz, nrm, n, stride, keep = _cloud_args(points, normals)
        self.params = params if params is not None else default_params()
        self.params.vote_count_threshold = float(vote_count_threshold)
        self.params.cpu_clustering = int(cpu_clustering)
        self.params.use_l1_norm = int(use_l1_norm)
        self.params.use_averaged_clusters = int(use_averaged_clusters)
        self.d_dist = float(d_dist)
        self.n = n
        self.best_T = None
        self.stats = None
        _check(lib().oslam_model_create(xyz, nrm, n, stride, self.d_dist, C.byref(self.params), C.byref(self._h)))

    def numPoints(self):
        return self.n

    # -- persistent model database (ppf.cu:64-66 asks for it; the reference rebuilds per pair) --
    def save(self, path):
        """Write the built table to `path` (oslam_model_save)."""
        _check(lib().oslam_model_save(self._h, os.fsencode(path)))

    @classmethod
    def load(cls, path, params=None):
        """A model whose table comes from a file written by save(): nothing is recomputed."""
        self = cls.__new__(cls)
        self._h = C.c_void_p(0)
        self.params = params if params is not None else default_params()
        self.best_T = None
        self.stats = None
        _check(lib().oslam_model_load(os.fsencode(path), C.byref(self.params) if params is not None else None,
                                      C.byref(self._h)))
        n, d = C.c_size_t(0), C.c_float(0)
        _check(lib().oslam_model_info(self._h, C.byref(n), C.byref(d), None))
        self.n, self.d_dist = n.value, d.value
        return self

    def table_bytes(self):
        b = C.c_uint64(0)
        _check(lib().oslam_model_info(self._h, None, None, C.byref(b)))
        return int(b.value)

    def SetModelPointVoteWeights(self, weights):
        w = np.ascontiguousarray(weights, np.float32)
        _check(lib().oslam_model_set_point_weights(self._h, _p(w), len(w)))

    def prepare(self, scene):
        """Allocate what the first ppf_lookup of this pair would (device scratch pool, pose-tail tables)."""
        _check(lib().oslam_align_prepare(self._h, scene._h))

    def ppf_lookup(self, scene, allow_no_votes=False):
        """Model::ppf_lookup (model.cu:269-306) + extraction (ppf.cu:74-93)."""
        T = np.zeros(16, np.float32)
        st = Stats()
        rc = lib().oslam_align(self._h, scene._h, _p(T), C.byref(st))
        if not (allow_no_votes and rc == OSLAM_E_NO_VOTES):
            _check(rc)
        self.best_T = T.reshape(4, 4)
        self.stats = st.asdict()
        return self.best_T

    def align_local(self, scene, cap=None):
        """This rank's votes.  cap=None: returns (number of peaks above the LOCAL threshold, local maximum) and
        leaves the records with the model for local_peaks(); with a cap: (records, local maximum), and
        OSLAM_E_LIMIT is raised when they do not fit (nothing is cut silently)."""
        n = C.c_size_t(0)
        lmax = C.c_uint32(0)
        st = Stats()
        if cap is None:
            _check(lib().oslam_align_local(self._h, scene._h, None, 0, C.byref(n), C.byref(lmax), C.byref(st)))
            self.stats = st.asdict()
            return int(n.value), int(lmax.value)
        cells = np.zeros(cap, CELL_DTYPE)
        _check(lib().oslam_align_local(self._h, scene._h, _p(cells), cap, C.byref(n), C.byref(lmax), C.byref(st)))
        self.stats = st.asdict()
        return cells[: n.value].copy(), int(lmax.value)

    def local_peaks(self, global_max):
        """Records of the last align_local above threshold * global_max (oslam_local_peaks)."""
        n = C.c_size_t(0)
        rc = lib().oslam_local_peaks(self._h, int(global_max), None, 0, C.byref(n))
        if rc not in (OSLAM_OK, OSLAM_E_LIMIT):
            _check(rc)
        cells = np.zeros(max(n.value, 1), CELL_DTYPE)
        _check(lib().oslam_local_peaks(self._h, int(global_max), _p(cells), len(cells), C.byref(n)))
        return cells[: n.value].copy()

    def align_multi(self, scene, comm, allow_no_votes=False):
        """One call per rank: votes of this rank's shard, RCCL exchange, pose tail on the union (oslam_align_multi)."""
        T = np.zeros(16, np.float32)
        st = Stats()
        rc = lib().oslam_align_multi(self._h, scene._h, comm._h, _p(T), C.byref(st))
        if not (allow_no_votes and rc == OSLAM_E_NO_VOTES):
            _check(rc)
        self.best_T = T.reshape(4, 4)
        self.stats = st.asdict()
        return self.best_T

    def align_finish(self, scene, cells, global_max, allow_no_votes=False):
        cells = np.ascontiguousarray(cells, CELL_DTYPE)
        T = np.zeros(16, np.float32)
        st = Stats()
        rc = lib().oslam_align_finish(self._h, scene._h, _p(cells), len(cells), int(global_max), _p(T), C.byref(st))
        if not (allow_no_votes and rc == OSLAM_E_NO_VOTES):
            _check(rc)
        self.best_T = T.reshape(4, 4)
        if self.stats is not None:                      # the local counters stay; the union's tail adds its own
            self.stats["num_top"], self.stats["max_count"] = st.num_top, st.max_count
        return self.best_T

    # -- result fields of the reference's Model (model.h:92-113) --
    def last_cells(self):
        n = C.c_size_t(0)
        _check(lib().oslam_last_cells(self._h, None, None, 0, C.byref(n)))
        cells = np.zeros(n.value, CELL_DTYPE)
        poses = np.zeros((n.value, 16), np.float32)
        if n.value:
            _check(lib().oslam_last_cells(self._h, _p(cells), _p(poses), n.value, C.byref(n)))
        return cells, poses

    @property
    def transformations(self):
        return self.last_cells()[1]

    def last_result(self, scene):
        """(transformation_trans [n,3], transformation_rots [n,4] wxyz, vote_counts_out [n], max_idx) of the last
        ppf_lookup against `scene` (model.h:100-113)."""
        n = C.c_size_t(0)
        _check(lib().oslam_last_cells(self._h, None, None, 0, C.byref(n)))
        k = max(n.value, 1)
        tr, ro, sc = np.zeros((k, 3), np.float32), np.zeros((k, 4), np.float32), np.zeros(k, np.float32)
        best = C.c_uint32(0)
        _check(lib().oslam_last_result(self._h, scene._h, _p(tr), _p(ro), _p(sc), k, C.byref(n), C.byref(best)))
        return tr[: n.value], ro[: n.value], sc[: n.value], int(best.value)

    def getHashKeys(self, ref_index):
        out = np.zeros(self.n, np.uint32)
        _check(lib().oslam_model_keys(self._h, int(ref_index), _p(out)))
        return out

    def bucket(self, key, cap=1 << 20):
        """Flat pair indices m_r*M + m_i stored under `key` (ParallelHashArray lookup)."""
        out = np.zeros(cap, np.uint32)
        n = C.c_size_t(0)
        _check(lib().oslam_model_bucket(self._h, int(key), _p(out), cap, C.byref(n)))
        return out[: min(n.value, cap)].copy(), n.value

    def bucket_words(self, key, slice_index=0, cap=1 << 20):
        """Stored entry words of `key`'s bucket in one slice, in storage order."""
        out = np.zeros(cap, np.uint32)
        n = C.c_size_t(0)
        _check(lib().oslam_model_bucket_words(self._h, int(key), int(slice_index), _p(out), cap, C.byref(n)))
        return out[: min(n.value, cap)].copy()

    def vote_accumulator(self, scene, ref_index):
        acc = np.zeros((self.n, 32), np.uint32)
        _check(lib().oslam_vote_accumulator(self._h, scene._h, int(ref_index), _p(acc)))
        return acc

    def close(self):
        if getattr(self, "_db", None) is not None:
            self._db.close()
        if self._h:
            lib().oslam_model_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Database:
    """Resident model database (oslam_db): models with one d_dist share the scene pass of every frame."""

    def __init__(self, models):
        self.models = list(models)
        self._h = C.c_void_p(0)
        arr = (C.c_void_p * len(self.models))(*[m._h for m in self.models])
        _check(lib().oslam_db_create(arr, len(self.models), C.byref(self._h)))
        n, g = C.c_size_t(0), C.c_size_t(0)
        _check(lib().oslam_db_size(self._h, C.byref(n), C.byref(g)))
        self.n_groups = g.value
        for m in self.models:              # the database borrows its models: it goes first
            m._db = self

    def align(self, scene):
        """-> (poses [n,4,4], list of per-model counters)."""
        n = len(self.models)
        T = np.zeros((n, 4, 4), np.float32)
        st = (Stats * n)()
        _check(lib().oslam_db_align(self._h, scene._h, _p(T), st))
        stats = [s.asdict() for s in st]
        for m, t, d in zip(self.models, T, stats):
            m.best_T, m.stats = t, d
        return T, stats

    def align_multi(self, scene, comm, n_total):
        """This rank's models (j = rank, rank + world, ... of n_total) against the whole scene, then every pose to
        every rank through the communicator (oslam_db_align_multi).  -> (poses [n_total,4,4], found [n_total])."""
        return db_align_multi(self, scene, comm, n_total)

    def close(self):
        if self._h:
            lib().oslam_db_destroy(self._h)
            self._h = C.c_void_p(0)
            for m in self.models:
                m._db = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def db_align_multi(db, scene, comm, n_total):
    """oslam_db_align_multi; db may be None on a rank that holds no model."""
    T = np.zeros((int(n_total), 4, 4), np.float32)
    found = np.zeros(int(n_total), np.int32)
    n_mine = len(db.models) if db is not None else 0
    st = (Stats * max(1, n_mine))()
    _check(lib().oslam_db_align_multi(db._h if db is not None else None, scene._h, comm._h, int(n_total), _p(T), _p(found), st))
    if db is not None:
        for m, d in zip(db.models, st):
            m.stats = d.asdict()
    return T, found


class Comm:
    """Communicator of the multi-GPU path (oslam_comm).  RCCL: rank 0 makes the id, everybody gets its bytes
    (here: through torch.distributed, any backend) and creates its end.  Comm.loopback(world): `world` emulated
    ranks on one device inside this process, one thread per rank -- the same exchange code, for tests."""

    def __init__(self, id_bytes, rank, world, dev, _handle=None):
        self._h = C.c_void_p(0)
        self.rank, self.world = rank, world
        if _handle is not None:
            self._h = C.c_void_p(_handle)
            return
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(id_bytes))
        _check(lib().oslam_comm_create(buf, int(rank), int(world), int(dev), C.byref(self._h)))

    @classmethod
    def loopback(cls, world, dev=0):
        arr = (C.c_void_p * int(world))()
        _check(lib().oslam_comm_create_loopback(int(world), int(dev), arr))
        return [cls(None, r, int(world), dev, _handle=arr[r]) for r in range(int(world))]

    def abort(self):
        _check(lib().oslam_comm_abort(self._h))

    def inject_failure(self, stage):
        _check(lib().oslam_comm_inject_failure(self._h, int(stage)))

    @property
    def broken(self):
        b = C.c_int(0)
        _check(lib().oslam_comm_info(self._h, None, None, C.byref(b)))
        return bool(b.value)

    @staticmethod
    def unique_id():
        buf = (C.c_char * COMM_ID_BYTES)()
        _check(lib().oslam_comm_unique_id(buf))
        return bytes(buf)

    def close(self):
        if self._h:
            lib().oslam_comm_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kernel_source_hash():
    """One hash over every file the device code is built from (kernels, their headers, the Makefile with its
    flags).  bench.py quotes PMC passes on file only for exactly this source; tools/make_pmc_traffic.py records it."""
    import hashlib
    d = os.path.join(_HERE, "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".inc")) or name == "Makefile":
            with open(os.path.join(d, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def release_scratch(dev=0):
    _check(lib().oslam_release_scratch(int(dev)))


def ppf_registration(scene_clouds, model_clouds, model_d_dists, ref_point_downsample_factor=1,
                     vote_count_threshold=0.4, cpu_clustering=False, use_l1_norm=False,
                     use_averaged_clusters=False, devUse=0, model_weights=None):
    """ppf_registration (ppf.h:9-15): clouds are (points, normals) tuples.
    Returns results[i][j] = 4x4 pose of model j in scene i."""
    L = lib()
    keep = []

    def pack(clouds):
        xs, ns, cnt = [], [], []
        for pts, nrm in clouds:
            p = np.ascontiguousarray(pts, np.float32)
            q = np.ascontiguousarray(nrm, np.float32)
            keep.extend([p, q])
            xs.append(p.ctypes.data)
            ns.append(q.ctypes.data)
            cnt.append(len(p))
        return ((C.c_void_p * len(xs))(*xs), (C.c_void_p * len(ns))(*ns), (C.c_size_t * len(cnt))(*cnt))

    sx, sn, sc = pack(scene_clouds)
    mx, mn, mc = pack(model_clouds)
    dd = np.ascontiguousarray(model_d_dists, np.float32)
    out = np.zeros((len(scene_clouds), len(model_clouds), 4, 4), np.float32)
    _check(L.oslam_ppf_registration(sx, sn, sc, len(scene_clouds), mx, mn, mc, len(model_clouds), 12, _p(dd),
                                    int(ref_point_downsample_factor), float(vote_count_threshold),
                                    int(cpu_clustering), int(use_l1_norm), int(use_averaged_clusters),
                                    int(devUse), None, _p(out)))
    return out


def ply_read(path):
    """(points [n,3], normals [n,3]) of a PLY file (pcl::io::loadPLYFile<PointNormal>)."""
    L = lib()
    px, pn, n = C.c_void_p(0), C.c_void_p(0), C.c_size_t(0)
    rc = L.oslam_ply_read(os.fsencode(path), C.byref(px), C.byref(pn), C.byref(n))
    if rc != OSLAM_OK:
        raise OslamError(rc, "cannot read PLY file %s" % path)
    try:
        shape = (n.value, 3)
        pts = np.ctypeslib.as_array(C.cast(px, C.POINTER(C.c_float)), shape=shape).copy() if n.value else np.zeros(shape, np.float32)
        nrm = np.ctypeslib.as_array(C.cast(pn, C.POINTER(C.c_float)), shape=shape).copy() if n.value else np.zeros(shape, np.float32)
    finally:
        L.oslam_free(px)
        L.oslam_free(pn)
    return pts, nrm


def ply_write(path, points, normals, binary=True):
    p = np.ascontiguousarray(points, np.float32)
    q = np.ascontiguousarray(normals, np.float32)
    rc = lib().oslam_ply_write(os.fsencode(path), _p(p), _p(q), len(p), int(binary))
    if rc != OSLAM_OK:
        raise OslamError(rc, "cannot write PLY file %s" % path)


def voxel_grid(points, normals=None, leaf=None, dev=0):
    """voxelGridDownsample (alignment.cpp:79-87): (points, normals) of the occupied voxels."""
    xyz, nrm, n, stride, keep = _cloud_args(points, normals)
    po, no = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    k = C.c_size_t(0)
    _check(lib().oslam_voxel_grid(xyz, nrm, n, stride, float(leaf), int(dev), _p(po), _p(no), n, C.byref(k)))
    return po[: k.value].copy(), no[: k.value].copy()


class Camera(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("depth_scale", C.c_float), ("z_min", C.c_float), ("z_max", C.c_float), ("max_jump", C.c_float)]


def depth_to_cloud(depth, fx, fy, cx, cy, depth_scale=0.001, z_min=0.1, z_max=10.0, max_jump=0.05, dev=0):
    """Depth image (uint16 or float32, [h,w]) -> (points, normals) in camera coordinates
    (oslam_depth_to_cloud; specification: oracle/oracle_depth.c)."""
    d = np.ascontiguousarray(depth)
    if d.dtype not in (np.uint16, np.float32) or d.ndim != 2:
        raise ValueError("depth must be a 2-D uint16 or float32 image")
    cam = Camera(fx, fy, cx, cy, depth_scale, z_min, z_max, max_jump)
    n = d.size
    po, no = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    k = C.c_size_t(0)
    _check(lib().oslam_depth_to_cloud(_p(d), int(d.dtype == np.uint16), d.shape[1], d.shape[0], C.byref(cam), int(dev),
                                      _p(po), _p(no), n, C.byref(k)))
    return po[: k.value].copy(), no[: k.value].copy()


def ht_dist(A, B):
    """ht_dist (linalg.cu:9-20): (|dt|, |rotation angle|)."""
    a = np.ascontiguousarray(A, np.float32).reshape(16)
    b = np.ascontiguousarray(B, np.float32).reshape(16)
    out = np.zeros(2, np.float32)
    _check(lib().oslam_ht_dist(_p(a), _p(b), _p(out)))
    return float(out[0]), float(out[1])


def pose_stage(cells, m_pts, m_nrm, s_pts, s_nrm, d_dist, cpu_clustering=False, use_l1_norm=False,
               use_averaged_clusters=False, weights=None, allow_no_votes=False):
    """Host stage only (no GPU): filtered+sorted cells -> (best T, all poses)."""
    cells = np.ascontiguousarray(cells, CELL_DTYPE)
    mp, mn = np.ascontiguousarray(m_pts, np.float32), np.ascontiguousarray(m_nrm, np.float32)
    sp, sn = np.ascontiguousarray(s_pts, np.float32), np.ascontiguousarray(s_nrm, np.float32)
    T = np.zeros(16, np.float32)
    poses = np.zeros((len(cells), 16), np.float32)
    w = None if weights is None else np.ascontiguousarray(weights, np.float32)
    rc = lib().oslam_pose_stage(_p(cells), len(cells), _p(mp), _p(mn), len(mp), _p(sp), _p(sn), len(sp),
                                float(d_dist), int(cpu_clustering), int(use_l1_norm), int(use_averaged_clusters),
                                _p(w) if w is not None else None, _p(T), _p(poses))
    if not (allow_no_votes and rc == OSLAM_E_NO_VOTES):
        if rc != OSLAM_OK:
            raise OslamError(rc, "pose stage failed")
    return T.reshape(4, 4), poses


def selftest_math(n=1 << 20, seed=1):
    bad = C.c_uint64(0)
    _check(lib().oslam_selftest_math(int(n), int(seed), C.byref(bad)))
    return int(bad.value)


def set_host_threads(n):
    _check(lib().oslam_set_host_threads(int(n)))


def set_stream(stream_ptr):
    _check(lib().oslam_set_stream(C.c_void_p(stream_ptr or 0)))
