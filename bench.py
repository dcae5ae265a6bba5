#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

Metric (BASELINE.json): scene PPF votes/sec, 5k-point model vs 100k-point scene.
A "step" is one pass of the hot path (Model::ppf_lookup: scene pair keys -> table
probe -> votes -> peak extraction -> pose) of ONE resident model table against ONE
scene already resident in HBM.  value = scene PPFs (valid ordered pairs
(reference point r, i != r), each keyed, probed and fully voted) processed by all
ranks per second.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU; scene reference points are dealt round-robin to
ranks (no data-path collective); per step an RCCL all-reduce(MAX) of the local
vote maximum and an all-gather of the records above the global threshold, then the host stage.
Weak scaling: ref_point_df = 8 / N keeps 12.5k reference points per GPU.

One JSON line on stdout (rank 0); progress goes to stderr.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model-points", type=int, default=5000)
    ap.add_argument("--scene-points", type=int, default=100000)
    ap.add_argument("--tau-d", type=float, default=0.025)
    ap.add_argument("--vote-mode", choices=["exact", "fast"], default="exact")
    ap.add_argument("--df", type=int, default=0, help="ref_point_df; 0 = 8 // gpus (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d" % args.gpus)
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # OSLAM_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks
    # (ranks share devices, the exchange runs on CPU tensors); the driver's runs use nccl = RCCL.
    backend = os.environ.get("OSLAM_BENCH_BACKEND", "nccl")
    xdev = "cuda" if backend == "nccl" else "cpu"
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)

    pkg = importlib.import_module("objective-slam_amd")
    ppf, synth = pkg.ppf, pkg.synth
    # torch.distributed.run exports OMP_NUM_THREADS=1 per rank; the host stage (poses of the gathered
    # peaks) may use a few threads: 8 per rank is 64 on an 8-GPU node
    if world > 1:
        ppf.set_host_threads(min(8, max(1, len(os.sched_getaffinity(0)) // world)))

    M, S = args.model_points, args.scene_points
    df = args.df if args.df > 0 else max(1, 8 // world)
    mp, mn = synth.make_model(0, M)
    d_dist = synth.d_dist_for(mp, args.tau_d)
    sp, sn, poses = synth.make_scene([0], S, 2002, instance_points=M, noise_sigma=0.1 * d_dist)
    diam = synth.bbox_extent(mp)

    mode = ppf.VOTE_FAST if args.vote_mode == "fast" else ppf.VOTE_EXACT
    par = ppf.default_params(dev=dev_index, shard_rank=rank, shard_world=world, vote_mode=mode)
    stream = torch.cuda.current_stream()
    ppf.set_stream(stream.cuda_stream)
    ppf.Model(mp[:64], mn[:64], d_dist=d_dist, params=par).close()   # loads the code objects: not part of a build
    t0 = time.time()
    model = ppf.Model(mp, mn, d_dist=d_dist, params=par)       # table resident in HBM
    t_build = time.time() - t0
    scene = ppf.Scene(sp, sn, d_dist=d_dist, ref_point_downsample_factor=df, params=par)   # resident in HBM
    model.prepare(scene)      # device scratch pool and pose-tail tables: set-up, like the model table itself
    log("[rank %d] model build %.3fs, d_dist %.5f, df %d" % (rank, t_build, d_dist, df))

    def step():
        if world == 1:
            T = model.ppf_lookup(scene)
            return T, dict(model.stats)
        cells, lmax = model.align_local(scene, cap=pkg.dist.LOCAL_CAP)      # vote kernels on this shard
        st = dict(model.stats)
        allrec, gmax = pkg.dist.gather_peaks(cells, lmax, xdev)             # RCCL all-reduce + all-gather
        T = model.align_finish(scene, allrec, gmax)                          # pose tail on the union
        st["num_top"] = model.stats["num_top"]
        return T, st

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record(stream)
    t0 = time.perf_counter()
    stats_acc, T = [], None
    for _ in range(args.steps):
        T, st = step()
        stats_acc.append(st)
    ev1.record(stream)
    sync()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
    ppfs = torch.tensor([float(sum(s["num_scene_ppfs"] for s in stats_acc)),
                         float(sum(s["num_votes"] for s in stats_acc))], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ppfs, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_ppfs, total_votes = float(ppfs[0].item()), float(ppfs[1].item())

    if rank == 0:
        st = stats_acc[-1]
        n_slices = (M + 1023) // 1024
        launches = sum(s["vote_launches"] for s in stats_acc)
        ms_vote_kernel = float(sum(s["ms_vote_kernel"] for s in stats_acc))   # HIP events around each launch
        ms_key_kernel = float(sum(s["ms_key_kernel"] for s in stats_acc))
        ms_path = float(np.mean([s["ms_vote"] for s in stats_acc]))
        # Algorithmic bytes: 8 B per table probe, 8 B per vote (one model-pair
        # entry), 16 B per emitted peak record; the vote kernel probes each hit once per slice.
        vote_bytes_step = 8.0 * st["num_votes"] + 8.0 * st["num_hits"] * n_slices + 16.0 * st["num_emitted"]
        per_launch_bytes = vote_bytes_step * args.steps / launches
        per_launch_ms = ms_vote_kernel / launches
        achieved = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
        # whole path of one align: scene read once + probe per scene pair + votes + records
        path_bytes = 24.0 * S + 8.0 * st["num_scene_ppfs"] + 8.0 * st["num_votes"] + 16.0 * st["num_emitted"]
        dt, dr = ppf.ht_dist(T, poses[0][1])
        out = {
            "metric": "scene_ppf_votes_per_sec",
            "value": total_ppfs / elapsed,
            "unit": "scene PPFs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "5k-point model vs 100k-point scene (BASELINE.json metric config), "
                                   "one resident model table, Model::ppf_lookup per step",
                       "model_points": M, "scene_points": S, "tau_d": args.tau_d, "d_dist": d_dist,
                       "ref_point_df": df, "ref_points_per_gpu": int(st["num_scene_ppfs"] // (S - 1)),
                       "vote_mode": args.vote_mode, "vote_count_threshold": 0.4,
                       "parallelism": "scene-ref-shard x%d" % world},
            "vote_increments_per_sec": total_votes / elapsed,
            "per_step": {"scene_ppfs": st["num_scene_ppfs"], "hits": st["num_hits"], "votes": st["num_votes"],
                         "nonempty_cells": st["num_unique_votes"], "model_keys": st["num_model_keys"],
                         "max_cell": st["max_count"], "emitted_records": st["num_emitted"],
                         "top_cells": st.get("num_top", 0)},
            "pose": {"rot_err_deg": float(np.degrees(dr)), "trans_err_frac_diam": dt / diam,
                     "ok_at_reference_criterion_12deg_0.1diam": bool(dr < np.radians(12) and dt < 0.1 * diam)},
            "model_build_s": t_build,
            "roofline": {"bound": "hbm", "kernel": "k_vote", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "traffic": pmc_traffic(args, M, S, df),
                         "traffic_note": "HBM-side bytes per vote-kernel launch (rocprofv3 PMC, profiles/). Far below the "
                                         "algorithmic bytes, and frac above 1, because the SURVEY 8d model charges 8 B to "
                                         "every vote while the kernel packs an entry into 4 B and streams a bucket once "
                                         "for all hits of a reference point that share it; the kernel is bound by "
                                         "LDS-atomic and vector issue, not by HBM (DESIGN.md 4)",
                         "alg_bytes_per_launch": per_launch_bytes, "launch_ms": per_launch_ms,
                         "launches_per_step": launches / args.steps,
                         "key_kernel_ms_per_step": ms_key_kernel / args.steps,
                         "path_achieved_GBps": path_bytes / (ms_path * 1e-3) / 1e9,
                         "path_alg_bytes_per_step": path_bytes, "path_kernels_ms_per_step": ms_path,
                         "torch_event_ms_per_step": ev0.elapsed_time(ev1) / args.steps},
        }
        # What binds the vote kernel in fact: the LDS atomics, one per vote.  Peak from
        # tools/micro/lds_atomic_bench.hip on MI355X: one conflict-free ds_add_u32 wave-instruction
        # (64 lane-atomics) per 4.4 cycles per CU, 256 CUs, 2.4 GHz.
        lds_peak = 256 * 64 * 2.4e9 / 4.4
        out["roofline_lds_atomic"] = {"bound": "lds_atomic", "kernel": "k_vote", "unit": "votes/s",
                                      "achieved": st["num_votes"] * args.steps / (ms_vote_kernel * 1e-3),
                                      "peak": lds_peak,
                                      "frac": st["num_votes"] * args.steps / (ms_vote_kernel * 1e-3) / lds_peak,
                                      "note": "secondary, informational: votes per second of vote-kernel time against the "
                                              "measured conflict-free LDS-atomic rate of the chip"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mp, mn, sp, sn, df, d_dist)
            out["pose_recall_at_1deg"] = pose_recall(ppf, synth, mode)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(args, M, S, df):
    """HBM bytes per vote-kernel launch from the committed rocprofv3 PMC passes
    (profiles/rNN_pmc_traffic.json of the latest round; FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as
    read), or None when no pass for this workload is on file.  Counters cannot be read from
    inside the timed process, so this is the one roofline field that is not measured live."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:          # the latest round's passes
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    for r in rec.get("runs", []):
        if (r["model_points"], r["scene_points"], r["ref_point_df"], r["vote_mode"], r["tau_d"]) == \
                (M, S, df, args.vote_mode, args.tau_d):
            return r["hbm_bytes_per_vote_launch"]
    return None


def pose_recall(ppf, synth, mode, trials=6):
    """Pose recall@1 degree as SURVEY.md 8d defines it: pose of this build vs pose of the CPU
    restatement on the same clouds, success iff rotation difference < 1 degree and translation
    difference < 1 % of the model diameter.  The CPU restatement needs hours at 5k x 100k, so the
    trials are 600-point models in 3000-point scenes (different models, poses and clutter)."""
    from oracle import oracle as O
    ok = 0
    worst = (0.0, 0.0)
    for k in range(trials):
        m_p, m_n = synth.make_model(k, 600)
        d = synth.d_dist_for(m_p, 0.05)
        s_p, s_n, _ = synth.make_scene([k], 3000, 4000 + k, instance_points=600, noise_sigma=0.1 * d)
        par = ppf.default_params(vote_mode=mode)
        T = ppf.Model(m_p, m_n, d_dist=d, params=par).ppf_lookup(
            ppf.Scene(s_p, s_n, d_dist=d, ref_point_downsample_factor=3, params=par), allow_no_votes=True)
        To, _, _ = O.align(m_p, m_n, s_p, s_n, 3, d)
        dt, dr = ppf.ht_dist(T, To)
        worst = (max(worst[0], float(np.degrees(dr))), max(worst[1], dt / synth.bbox_extent(m_p)))
        ok += int(np.degrees(dr) < 1.0 and dt < 0.01 * synth.bbox_extent(m_p))
    return {"trials": trials, "recall": ok / trials, "against": "CPU restatement (oracle) on the same clouds",
            "sizes": "600-point models, 3000-point scenes, ref_point_df 3",
            "worst_rot_diff_deg": worst[0], "worst_trans_diff_frac_diam": worst[1]}


def cpu_baseline(mp, mn, sp, sn, df, d_dist):
    """The CPU oracle's fused path (kind "port"; no reference build exists here) on a bounded
    sample of the same workload: the first 8*cores reference points, OpenMP over them."""
    from oracle import oracle as O
    # a 1-GPU box shares its host: use at most its 16-core share (the affinity mask shows every core)
    cores = min(16, len(os.sched_getaffinity(0)))
    t0 = time.time()
    fm = O.FusedModel(mp, mn, d_dist)
    log("[cpu baseline] model table on the CPU: %.1fs" % (time.time() - t0))
    n_ref = 8 * cores
    t0 = time.perf_counter()
    _, st = fm.votes(sp, sn, df, ref_limit=n_ref, threads=cores)
    el = time.perf_counter() - t0
    fm.close()
    return {"value": st["num_scene_ppfs"] / el, "unit": "scene PPFs/s", "cores": cores, "kind": "port",
            "sample": "first %d scene reference points of the same workload (%d scene PPFs, %d votes) in %.1f s; "
                      "model table build excluded" % (n_ref, st["num_scene_ppfs"], st["num_votes"], el),
            "vote_increments_per_sec": st["num_votes"] / el}


if __name__ == "__main__":
    main()
