"""CPU, world_size 2 over gloo: the multi-GPU exchange step (dist.gather_peaks) and the host
finish.  The vote kernels need a GPU, so each rank's local peaks come from the oracle run on
that rank's shard of reference points -- the exchange, the global threshold and the host stage
are the product's own code, and the result must equal the single-process answer."""
import importlib
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import oracle as O
    pkg = importlib.import_module("objective-slam_amd")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m_p, m_n = pkg.synth.make_model(0, 160)
    d = pkg.synth.d_dist_for(m_p, 0.05)
    s_p, s_n, _ = pkg.synth.make_scene([0], 420, 2030, instance_points=160)
    df = 2
    # this rank's shard: reference points df*(rank + world*t), as oslam_scene_create deals them
    local, st = O.votes_fused(m_p, m_n, s_p, s_n, df, d, 0.4, ref_begin=rank, ref_step=world)
    allrec, gmax = pkg.dist.gather_peaks(local, st["max_count"], "cpu", 0.4)
    T, kept = pkg.dist.finish_on_host(allrec, gmax, m_p, m_n, s_p, s_n, d)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), T=T, code=kept["code"], count=kept["count"], gmax=gmax)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_equals_single_process(tmp_path, oracle, synth):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    m_p, m_n = synth.make_model(0, 160)
    d = synth.d_dist_for(m_p, 0.05)
    s_p, s_n, _ = synth.make_scene([0], 420, 2030, instance_points=160)
    cells, st = oracle.votes_fused(m_p, m_n, s_p, s_n, 2, d, 0.4)
    rc, To = oracle.pose_from_cells(cells, m_p, m_n, s_p, s_n, d)
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        assert int(g["gmax"]) == st["max_count"]
        assert np.array_equal(g["code"], cells["code"]) and np.array_equal(g["count"], cells["count"])
        assert np.array_equal(g["T"], To)
