"""CPU: the recall-vs-occlusion protocol (objective-slam_amd/evaluate.py) restates the
reference's analyze_mian.py: file formats, match rule and cumulative curve on hand-made inputs."""
import importlib

import numpy as np

ev = importlib.import_module("objective-slam_amd.evaluate")


def test_protocol_on_hand_made_rows(tmp_path):
    occ = tmp_path / "occlusion.txt"
    occ.write_text("scene model occlusion\n1 chef 60.5\n1 trex 82.0\n2 chef 71.25\n2 para 90\n")
    rows = ev.read_occlusion_txt(str(occ))
    assert rows == [["1", "chef", 60.5], ["1", "trex", 82.0], ["2", "chef", 71.25], ["2", "para", 90.0]]
    # a log as oslam_alignment (or the reference's binary under Boost.Log) prints it
    log1 = tmp_path / "rs1_gpu.log"
    log1.write_text("[oslam] [info] Transformations for models/cheff_view.ply in scenes/rs1.ply:\n"
                    " 1 0 0 0\n[oslam] [info] Distance (trans, rot): 12.5, 0.1\n"
                    "[oslam] [info] Transformations for models/T-rex_high.ply in scenes/rs1.ply:\n"
                    "[oslam] [info] Distance (trans, rot): 80.0, 3.0\n")
    log2 = tmp_path / "rs2_gpu.log"
    log2.write_text("[oslam] [info] Transformations for models/cheff_view.ply in scenes/rs2.ply:\n"
                    "[oslam] [info] Distance (trans, rot): 10.0, 6.2\n"
                    "[oslam] [info] Transformations for models/parasaurolophus_high.ply in scenes/rs2.ply:\n"
                    "[oslam] [info] Distance (trans, rot): 60.0, 0.05\n")
    ev.read_alignment_log(rows, str(log1), "1")
    ev.read_alignment_log(rows, str(log2), "2")
    assert [r[3] for r in rows] == [[12.5, 0.1], [80.0, 3.0], [10.0, 6.2], [60.0, 0.05]]
    diam = {"chef": 136.59418, "trex": 98.828925, "para": 131.250275}        # analyze_mian.py:43-48
    table, cum = ev.recall_table(rows, diam, bins=(0, 75, 101))
    # sorted by occlusion: chef 60.5 (match), chef 71.25 (6.2 rad = 0.083 rad from a full turn: match),
    # trex 82 (80 > 0.3 * 98.8, 3 rad: no), para 90 (60 > 39.4: no)
    assert [r[1] for r in rows] == ["chef", "chef", "trex", "para"]
    assert [all(r[4]) for r in rows] == [True, True, False, False]
    assert cum == [1.0, 1.0, 2 / 3, 0.5]
    assert table == [{"occlusion": "[0, 75)", "pairs": 2, "recall": 1.0},
                     {"occlusion": "[75, 101)", "pairs": 2, "recall": 0.0}]


def test_occluded_scene_generation():
    synth = importlib.import_module("objective-slam_amd.synth")
    full = synth.make_scene([0], 4000, 31, instance_points=1000)
    cut = synth.make_scene([0], 4000, 31, instance_points=1000, occlusion=0.7)
    assert len(cut[0]) == 4000 == len(full[0])
    T = full[2][0][1]
    assert np.array_equal(T, cut[2][0][1])           # the same pose: the cut only removes samples
    # points on the instance = those whose back-transformed position is within the model's box
    def on_instance(p):
        q = (p.astype(np.float64) - T[:3, 3]) @ T[:3, :3]
        return int((np.abs(q).max(axis=1) < 2.2).sum())
    assert on_instance(cut[0]) < 0.6 * on_instance(full[0])
