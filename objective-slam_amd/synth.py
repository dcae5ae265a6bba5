"""Deterministic synthetic clouds for tests and bench.

No `.ply`/`.pcd` data ships with the reference (reference .gitignore:1-2), so
every workload is generated: a closed, asymmetric "bumpy" parametric surface as
the model, and scenes made of posed model instances + clutter surfaces + a
ground plane + point noise.  The PRNG is an explicit SplitMix64 so clouds are
bit-identical across numpy versions and machines.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


class SplitMix64:
    """Counter-based SplitMix64 stream, vectorised."""

    def __init__(self, seed):
        self.state = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)

    def u64(self, n):
        with np.errstate(over="ignore"):
            idx = np.arange(1, n + 1, dtype=np.uint64)
            z = self.state + idx * np.uint64(0x9E3779B97F4A7C15)
            self.state = z[-1] if n else self.state
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            return z ^ (z >> np.uint64(31))

    def uniform(self, n):
        """float64 in [0,1) with 24 random bits (exactly representable in f32)."""
        return (self.u64(n) >> np.uint64(40)).astype(np.float64) / float(1 << 24)

    def normal(self, n):
        m = (n + 1) // 2
        u1 = 1.0 - self.uniform(m)
        u2 = self.uniform(m)
        r = np.sqrt(-2.0 * np.log(u1))
        z = np.concatenate([r * np.cos(2 * np.pi * u2), r * np.sin(2 * np.pi * u2)])
        return z[:n]


def _surface(u, v, k):
    """Bumpy closed surface number k and its (unnormalised) normal, float64."""
    a, b, c = 3 + (k % 3), 2 + (k % 2), 5 + (k % 4)
    e1, e2 = 0.3 + 0.03 * (k % 5), 0.2 + 0.02 * (k % 7)
    sx, sy, sz = 1.5 - 0.05 * (k % 4), 1.0 + 0.04 * (k % 3), 0.7 + 0.03 * (k % 5)
    r = 1 + e1 * np.sin(a * u) * np.cos(b * v) + e2 * np.cos(c * v + u)
    ru = e1 * a * np.cos(a * u) * np.cos(b * v) - e2 * np.sin(c * v + u)
    rv = -e1 * b * np.sin(a * u) * np.sin(b * v) - e2 * c * np.sin(c * v + u)
    cu, su, cv, sv = np.cos(u), np.sin(u), np.cos(v), np.sin(v)
    p = np.stack([sx * r * cu * sv, sy * (r * su * sv + 0.3 * u), sz * r * cv], axis=1)
    pu = np.stack([sx * (ru * cu - r * su) * sv, sy * ((ru * su + r * cu) * sv + 0.3), sz * ru * cv], axis=1)
    pv = np.stack([sx * (rv * sv + r * cv) * cu, sy * (rv * sv + r * cv) * su, sz * (rv * cv - r * sv)], axis=1)
    n = np.cross(pv, pu)  # outward for this parametrisation
    return p, n


def sample_surface(k, n_points, rng):
    """n_points area-uniform samples (rejection on the area element)."""
    pts, nrm = [], []
    have = 0
    wmax = None
    while have < n_points:
        m = max(4096, 3 * (n_points - have))
        u = rng.uniform(m) * 2 * np.pi
        v = 0.2 + rng.uniform(m) * 2.7
        acc = rng.uniform(m)
        p, n = _surface(u, v, k)
        w = np.linalg.norm(n, axis=1)
        if wmax is None:
            wmax = 1.25 * w.max()
        keep = (acc * wmax < w) & (w > 1e-9)
        pts.append(p[keep])
        nrm.append(n[keep] / w[keep, None])
        have += int(keep.sum())
    pts = np.concatenate(pts)[:n_points]
    nrm = np.concatenate(nrm)[:n_points]
    return pts, nrm


def make_model(k=0, n_points=1000, seed=None):
    """Model cloud k: (points f32 [M,3], normals f32 [M,3])."""
    rng = SplitMix64(1000 + k if seed is None else seed)
    p, n = sample_surface(k, n_points, rng)
    return np.ascontiguousarray(p, np.float32), np.ascontiguousarray(n, np.float32)


def random_rotation(rng):
    """Shoemake uniform rotation (reference scene_generation.hpp:33-51 idea)."""
    u1, u2, u3 = rng.uniform(3)
    q = np.array([np.sqrt(1 - u1) * np.sin(2 * np.pi * u2), np.sqrt(1 - u1) * np.cos(2 * np.pi * u2),
                  np.sqrt(u1) * np.sin(2 * np.pi * u3), np.sqrt(u1) * np.cos(2 * np.pi * u3)])
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def bbox_extent(points):
    return float((points.max(axis=0) - points.min(axis=0)).max())


def d_dist_for(points, tau_d):
    """d_dist = tau_d * max bbox extent (reference alignment.cpp:246-253), in float32."""
    p = np.asarray(points, np.float32)
    ext = (p.max(axis=0) - p.min(axis=0)).astype(np.float32)
    return float(np.float32(tau_d) * ext.max())


def make_scene(model_ids, n_points, seed, n_instances=1, instance_points=None, noise_sigma=0.0,
               box_diameters=10.0, n_clutter=None, occlusion=0.0):
    """Scene with `n_instances` posed copies of each model id in `model_ids`.

    Returns (points f32 [S,3], normals f32 [S,3], poses) where poses is a list of
    (model_id, 4x4 float64 ground-truth pose, model -> scene).
    The instance points are a fresh sampling of the model surface, not the model
    cloud itself, as a sensor would see it.  occlusion in [0, 1): that share of every instance's
    surface samples is cut away on one side (a half space in a random direction, as an occluder
    in front of the object would); the freed point budget goes to the ground plane.
    """
    rng = SplitMix64(seed)
    diam = 3.5
    half = 0.5 * box_diameters * diam
    n_obj = len(model_ids) * n_instances
    if instance_points is None:
        instance_points = max(64, n_points // (4 * max(1, n_obj)))
    if n_clutter is None:
        n_clutter = 6
    n_inst_total = min(n_points, instance_points * n_obj)
    n_plane = (n_points - n_inst_total) // 3
    n_clut_total = n_points - n_inst_total - n_plane
    n_keep = max(8, int(round((n_inst_total // n_obj) * (1.0 - occlusion)))) if occlusion > 0 else n_inst_total // n_obj
    n_plane += n_obj * (n_inst_total // n_obj - n_keep)
    pts, nrm, poses = [], [], []
    for mid in model_ids:
        for _ in range(n_instances):
            R = random_rotation(rng)
            t = (rng.uniform(3) * 2 - 1) * half
            t[2] = abs(t[2]) * 0.3 + 1.5
            p, n = sample_surface(mid, n_inst_total // n_obj, rng)
            if occlusion > 0:
                dvec = random_rotation(rng)[:, 0]
                keep = np.argsort(p @ dvec, kind="stable")[:n_keep]
                keep.sort()
                p, n = p[keep], n[keep]
            pts.append(p @ R.T + t)
            nrm.append(n @ R.T)
            T = np.eye(4)
            T[:3, :3] = R
            T[:3, 3] = t
            poses.append((mid, T))
    for c in range(n_clutter):
        m = n_clut_total // n_clutter + (1 if c < n_clut_total % n_clutter else 0)
        if m == 0:
            continue
        R = random_rotation(rng)
        t = (rng.uniform(3) * 2 - 1) * half
        t[2] = abs(t[2]) * 0.3 + 1.5
        p, n = sample_surface(50 + c, m, rng)
        s = 0.6 + 0.8 * rng.uniform(1)[0]
        pts.append((p * s) @ R.T + t)
        nrm.append(n @ R.T)
    if n_plane:
        xy = (rng.uniform(2 * n_plane).reshape(-1, 2) * 2 - 1) * half
        pts.append(np.concatenate([xy, np.zeros((n_plane, 1))], axis=1))
        nrm.append(np.tile(np.array([[0.0, 0.0, 1.0]]), (n_plane, 1)))
    pts = np.concatenate(pts)
    nrm = np.concatenate(nrm)
    # fixed pseudo-random interleaving so reference points (every df-th) cover all objects
    order = np.argsort(rng.u64(len(pts)), kind="stable")
    pts, nrm = pts[order], nrm[order]
    if noise_sigma > 0:
        pts = pts + noise_sigma * rng.normal(pts.size).reshape(pts.shape)
    assert len(pts) == n_points
    return np.ascontiguousarray(pts, np.float32), np.ascontiguousarray(nrm, np.float32), poses


def transform_cloud(points, normals, T):
    """Apply a rigid 4x4 (model -> scene) to points and normals (float64 math, f32 out)."""
    R, t = T[:3, :3], T[:3, 3]
    return (np.ascontiguousarray(points.astype(np.float64) @ R.T + t, np.float32),
            np.ascontiguousarray(normals.astype(np.float64) @ R.T, np.float32))


def render_depth(points, width=640, height=480, fx=525.0, fy=525.0, cx=319.5, cy=239.5, depth_scale=0.001,
                 background_z=None, splat=1):
    """Z-buffer rendering of a dense point set (camera coordinates, z forward) into a uint16 depth
    image in units of depth_scale metres: every point is splatted over (2*splat+1)^2 pixels and
    the nearest wins; background_z (metres) fills the pixels nothing projects to (a wall behind
    the objects), 0 leaves them invalid.  For the streaming test and bench; not a product path."""
    p = np.asarray(points, np.float64)
    z = p[:, 2]
    ok = z > 1e-6
    p, z = p[ok], z[ok]
    u = np.rint(p[:, 0] * fx / z + cx).astype(np.int64)
    v = np.rint(p[:, 1] * fy / z + cy).astype(np.int64)
    raw = np.clip(np.rint(z / depth_scale), 1, 65535).astype(np.int64)
    img = np.full(width * height, 65536, np.int64)
    for dv in range(-splat, splat + 1):
        for du in range(-splat, splat + 1):
            uu, vv = u + du, v + dv
            m = (uu >= 0) & (uu < width) & (vv >= 0) & (vv < height)
            np.minimum.at(img, vv[m] * width + uu[m], raw[m])
    bg = 0 if background_z is None else int(round(background_z / depth_scale))
    img[img == 65536] = bg
    return img.reshape(height, width).astype(np.uint16)
