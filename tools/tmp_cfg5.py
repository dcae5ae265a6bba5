import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
ppf, synth = pkg.ppf, pkg.synth
mp, mn = synth.make_model(0, 1500); d = synth.d_dist_for(mp, 0.05)
g = ppf.voxel_grid(mp, mn, leaf=d); mo = ppf.Model(g[0], g[1], d_dist=d)
dense = synth.make_model(0, 300000)[0]
rng = synth.SplitMix64(55); T = np.eye(4, dtype=np.float32); T[:3, :3] = synth.random_rotation(rng); T[:3, 3] = [-2, -0.1, 7]
img = synth.render_depth(dense @ T[:3, :3].T + T[:3, 3], background_z=10.0, splat=1)
sp, sn = ppf.depth_to_cloud(img, 525.0, 525.0, 319.5, 239.5, z_min=0.5, z_max=12.0, max_jump=0.08)
sg = ppf.voxel_grid(sp, sn, leaf=d)
print("model", len(g[0]), "scene", len(sg[0]))
for it in range(3):
    t = time.perf_counter(); sc = ppf.Scene(sg[0], sg[1], d_dist=d, ref_point_downsample_factor=2); t_sc = time.perf_counter() - t
    t = time.perf_counter(); mo.ppf_lookup(sc); t_al = time.perf_counter() - t
    st = mo.stats
    print("scene_create %.2f ms, align %.2f ms (oslam total %.2f, events %.2f: vote kernel %.2f key+sort %.2f), launches %d, top %d emitted %d votes %d hits %d max %d"
          % (1e3 * t_sc, 1e3 * t_al, st["ms_total"], st["ms_vote"], st["ms_vote_kernel"], st["ms_key_kernel"], st["vote_launches"], st["num_top"], st["num_emitted"], st["num_votes"], st["num_hits"], st["max_count"]))
    sc.close()
