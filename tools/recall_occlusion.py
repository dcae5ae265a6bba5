"""Recall vs occlusion on synthetic scenes (the protocol of the reference's analyze_mian.py; the UWA
data set it was run on is not available here).  Needs a HIP device.
usage: python tools/recall_occlusion.py [trials_per_level] > profiles/rNN_recall_occlusion.json"""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("objective-slam_amd")
ev = importlib.import_module("objective-slam_amd.evaluate")
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 8
levels = [0.0, 0.3, 0.5, 0.6, 0.7, 0.75, 0.8, 0.85, 0.9]
rows, diam = ev.synthetic_rows(levels, trials_per_level=trials, log=lambda m: print(m, file=sys.stderr, flush=True))
table, cum = ev.recall_table(rows, diam)
print(json.dumps({"protocol": "analyze_mian.py: match iff dt <= 0.3 diameters and rotation <= 12 degrees",
                  "data": "synthetic: bumpy-surface models 0, 2, 4, 6 (1500 points, voxel grid at d_dist, tau_d 0.05), one instance per "
                          "20000-point cluttered scene, a half-space cut removes the occluded share; ref_point_df 5",
                  "pairs": len(rows), "recall_by_occlusion": table,
                  "cumulative_recall_at_max_occlusion": cum[-1] if cum else None}, indent=1))
