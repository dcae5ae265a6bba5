#!/usr/bin/env python3
"""bench.py -- the reference's headline metric on MI355X.

Metric (BASELINE.json): scene PPF votes/sec, 5k-point model vs 100k-point scene.
A "step" is one pass of the hot path (Model::ppf_lookup: scene pair keys -> table
probe -> votes -> peak extraction -> pose) of ONE resident model table against ONE
scene already resident in HBM.  value = scene PPFs (valid ordered pairs
(reference point r, i != r)) processed by all ranks per second.  Every pair is taken
through the path: its distance bin is tested against the model (exact: a pair in a bin
that holds no model key cannot match, FNV collisions included), the pairs that pass
(`pairs_probed`) are keyed and probed, the ones whose key is in the model (`hits`)
vote with their whole bucket.  `hits_per_sec` and `vote_increments_per_sec` are the
rates that do not depend on how much of the scene lies out of the model's reach.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU; scene reference points are dealt round-robin to
ranks (no data-path collective); per step one oslam_align_multi call per rank (C-ABI):
votes of the shard, RCCL all-reduce(MAX) of the vote maximum, all-gather of the records
above the global threshold (device buffers end to end), pose tail on the union.
Weak scaling: ref_point_df = 8 / N keeps 12.5k reference points per GPU.

One JSON line on stdout (rank 0); progress goes to stderr.
"""
import argparse
import hashlib
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


def kernel_source_hash():
    import importlib
    return importlib.import_module("objective-slam_amd").ppf.kernel_source_hash()


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
    # OSLAM_BENCH_BACKEND=gloo rehearses the N>1 path on a box with fewer GPUs than ranks (ranks share
    # devices, the exchange runs through host buffers); the driver's runs use nccl = RCCL, inside the C library.
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
    model.prepare(scene)      # per-device counters and pose-tail tables: set-up, like the model table itself
    # N > 1 over RCCL: the exchange runs inside the library (oslam_align_multi).  If its communicator cannot be made on
    # this node the same steps run through host buffers and torch.distributed instead -- on every rank or on none
    comm, exchange = None, "single GPU"
    if world > 1 and backend == "nccl":
        # every rank enters the same collectives whatever fails where (dist.make_comm: rank 0 always broadcasts, an id
        # or None; then the ranks agree on whether all of them have a communicator)
        comm, why = pkg.dist.make_comm(dev_index)
        if comm is None:
            log("[rank %d] no RCCL communicator (%s): exchange through host buffers" % (rank, why))
        exchange = "oslam_align_multi (RCCL inside the library)" if comm is not None else "host buffers + torch.distributed (nccl)"
    elif world > 1:
        exchange = "host buffers + torch.distributed (%s)" % backend
    log("[rank %d] model build %.3fs, d_dist %.5f, df %d" % (rank, t_build, d_dist, df))

    def step(sc=scene):
        if world == 1:
            T = model.ppf_lookup(sc)
            return T, dict(model.stats)
        if comm is not None:
            T = model.align_multi(sc, comm)                                    # one C call: votes + RCCL exchange + pose tail
            return T, dict(model.stats)
        allrec, gmax = pkg.dist.align_sharded_host(model, sc, xdev)            # rehearsal: same steps through host buffers
        st = dict(model.stats)
        T = model.align_finish(sc, allrec, gmax)
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
    tot = torch.tensor([float(sum(s["num_scene_ppfs"] for s in stats_acc)),
                        float(sum(s["num_votes"] for s in stats_acc)),
                        float(sum(s["num_hits"] for s in stats_acc))], dtype=torch.float64, device=xdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_ppfs, total_votes, total_hits = (float(x) for x in tot.tolist())

    # "H2D of the scene included and also reported separately": the same step with the Scene
    # built from host buffers inside it (upload + reference frames on the host), a few repetitions, rank-local
    t_incl = []
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        sc2 = ppf.Scene(sp, sn, d_dist=d_dist, ref_point_downsample_factor=df, params=par)
        t2 = time.perf_counter()
        step(sc2)
        torch.cuda.synchronize()
        t_incl.append((time.perf_counter() - t1, t2 - t1))
        sc2.close()
    ms_incl = 1e3 * float(np.median([a for a, _ in t_incl]))
    ms_scene = 1e3 * float(np.median([b for _, b in t_incl]))

    if rank == 0:
        st = stats_acc[-1]
        n_slices = (M + 2045) // 2046
        launches = sum(s["vote_launches"] for s in stats_acc)
        ms_vote_kernel = float(sum(s["ms_vote_kernel"] for s in stats_acc))   # HIP events around each launch
        ms_key_kernel = float(sum(s["ms_key_kernel"] for s in stats_acc))
        ms_path = float(np.mean([s["ms_vote"] for s in stats_acc]))
        per_launch_ms = ms_vote_kernel / launches
        steps_per_launch = args.steps / launches
        # ---- k_vote, bytes this design cannot avoid: every bucket a run of hits needs is read once
        # (4 B per model pair entry), 8 B of run record + 8 B of bucket record per (run, slice), 8 B per hit and
        # slice for its angle, 16 B per emitted peak record
        design_bytes_step = (4.0 * st["num_entries_streamed"] + 16.0 * st["num_items"] + 8.0 * st["num_hits"] * n_slices
                             + 16.0 * st["num_emitted"])
        design_launch = design_bytes_step * steps_per_launch
        achieved = design_launch / (per_launch_ms * 1e-3) / 1e9
        # ---- the same launch by SURVEY 8d's formula (8 B per vote, 8 B per probe, 16 B per record): not a
        # ceiling for this design (an entry is 4 B and a bucket is read once per run of hits, not once per hit)
        survey_bytes_step = 8.0 * st["num_votes"] + 8.0 * st["num_hits"] * n_slices + 16.0 * st["num_emitted"]
        survey_launch = survey_bytes_step * steps_per_launch
        survey_achieved = survey_launch / (per_launch_ms * 1e-3) / 1e9
        path_bytes = 24.0 * S + 8.0 * st["num_pairs_probed"] + 8.0 * st["num_votes"] + 16.0 * st["num_emitted"]
        votes_per_s_kernel = st["num_votes"] * args.steps / (ms_vote_kernel * 1e-3)
        # ceilings of the vote loop (tools/micro/): one conflict-free ds_add_u32 wave-instruction
        # (64 votes) per 4.4 cycles per CU (measured); one hit voting with a chunk = 256 votes = 13 vector
        # instructions in both modes (exact mode finds its near-edge votes by search, outside the loop), at their
        # stand-alone issue costs (v_sub_u32 2.33 cycles per wave-instruction, everything else in the loop 4.2-4.3:
        # tools/micro/valu_rate_bench.hip) on one of 1024 SIMDs; 2.4 GHz.  The kernel is bound by the issue of ALL its
        # instructions (3.7 cycles per instruction and SIMD, scalar ones included: profiles/r03_pmc_sq_k_vote.txt,
        # r03_ab_step_loads_and_hit_loop.txt); the loop's own 20 instructions per hit and chunk are 44 % of them,
        # which is why this fraction cannot reach 1.
        lds_peak = 256 * 64 * 2.4e9 / 4.4
        valu_instr = 13
        valu_cycles = 47.5
        valu_peak = 1024 * 256 * 2.4e9 / valu_cycles
        dt, dr = ppf.ht_dist(T, poses[0][1])
        traffic, traffic_note = pmc_traffic(args, M, S, df)
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
                       "parallelism": "scene-ref-shard x%d" % world, "exchange": exchange},
            "vote_increments_per_sec": total_votes / elapsed,
            "hits_per_sec": total_hits / elapsed,
            "ms_per_step_incl_scene": ms_incl,
            "ms_scene_create": ms_scene,
            "per_step": {"scene_ppfs": st["num_scene_ppfs"], "pairs_probed": st["num_pairs_probed"],
                         "hits": st["num_hits"], "votes": st["num_votes"],
                         "entries_streamed": st["num_entries_streamed"], "buckets_streamed": st["num_items"],
                         "nonempty_cells": st["num_unique_votes"], "model_keys": st["num_model_keys"],
                         "max_cell": st["max_count"], "emitted_records": st["num_emitted"],
                         "top_cells": st.get("num_top", 0), "hit_list_pool_bytes": st["scratch_bytes"]},
            "pose": {"rot_err_deg": float(np.degrees(dr)), "trans_err_frac_diam": dt / diam,
                     "ok_at_reference_criterion_12deg_0.1diam": bool(dr < np.radians(12) and dt < 0.1 * diam)},
            "model_build_s": t_build,
            "roofline": {"bound": "hbm", "kernel": "k_vote", "achieved": achieved, "peak": HBM_PEAK_GBPS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                         "measured_against": "design bytes vs the HBM peak; the 100 MB entry stream is served from the "
                                             "256 MiB Infinity Cache, so this prices the stream, not HBM pins",
                         "traffic": traffic, "traffic_note": traffic_note,
                         "traffic_ratio": (traffic / design_launch) if traffic else None,
                         "alg_bytes_per_launch": design_launch, "launch_ms": per_launch_ms,
                         "launches_per_step": launches / args.steps,
                         "alg_bytes_model": "4 B x model pair entries streamed (each bucket once per run of hits and slice) "
                                            "+ 16 B per (run, slice) + 8 B per hit and slice + 16 B per record",
                         "binding": "instruction issue, all kinds counted (3.7 cycles per instruction and SIMD with four "
                                    "waves per SIMD), then LDS atomics: see roofline_valu / roofline_lds_atomic "
                                    "(DESIGN.md 4); the entry stream comes out of the Infinity Cache",
                         "key_kernels_ms_per_step": ms_key_kernel / args.steps,
                         "path_kernels_ms_per_step": ms_path,
                         "torch_event_ms_per_step": ev0.elapsed_time(ev1) / args.steps},
            "roofline_valu": {"bound": "valu_issue", "kernel": "k_vote", "unit": "votes/s", "achieved": votes_per_s_kernel,
                              "peak": valu_peak, "frac": votes_per_s_kernel / valu_peak,
                              "note": "votes per second of vote-kernel time against 256 votes per %d vector instructions = "
                                      "%.1f SIMD cycles at their measured issue costs (one vote iteration of the inner "
                                      "loop, nothing else)" % (valu_instr, valu_cycles)},
            "roofline_lds_atomic": {"bound": "lds_atomic", "kernel": "k_vote", "unit": "votes/s",
                                    "achieved": votes_per_s_kernel, "peak": lds_peak, "frac": votes_per_s_kernel / lds_peak,
                                    "note": "against the measured conflict-free ds_add_u32 rate of the chip"},
            "roofline_survey_8d": {"bound": "hbm", "kernel": "k_vote", "achieved": survey_achieved, "peak": HBM_PEAK_GBPS,
                                   "unit": "GB/s", "frac": survey_achieved / HBM_PEAK_GBPS,
                                   "alg_bytes_per_launch": survey_launch,
                                   "path_achieved_GBps": path_bytes / (ms_path * 1e-3) / 1e9,
                                   "note": "SURVEY.md 8d's byte model (8 B per vote): above 1 by construction for a design that "
                                           "packs an entry into 4 B and streams a bucket once per run of hits; kept for comparison"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mp, mn, sp, sn, df, d_dist)
            out["pose_recall_at_1deg"] = pose_recall(ppf, synth, mode, T, model.last_cells()[0], args, df)
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(args, M, S, df):
    """HBM-side bytes per vote-kernel launch from the rocprofv3 PMC passes of the latest round
    (profiles/rNN_pmc_traffic.json; FETCH_SIZE doubled per the gfx950 correction, WRITE_SIZE as read).
    Counters cannot be read from inside the timed process, so this is the one roofline field that is
    not measured live; it is reported only when the passes were taken with the kernel source that is
    running now (hash on file) and for this workload -- otherwise null, with the reason."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    if not files:
        return None, "no PMC passes on file"
    try:
        with open(files[-1]) as f:          # the latest round's passes
            rec = json.load(f)
    except (OSError, ValueError):
        return None, "unreadable " + os.path.basename(files[-1])
    if rec.get("kernel_source_sha16") != kernel_source_hash():
        return None, ("%s was taken with other kernel sources (stale): re-run tools/profile_round.sh"
                      % os.path.basename(files[-1]))
    for r in rec.get("runs", []):
        if (r["model_points"], r["scene_points"], r["ref_point_df"], r["vote_mode"], r["tau_d"]) == \
                (M, S, df, args.vote_mode, args.tau_d):
            return r["hbm_bytes_per_vote_launch"], ("HBM-side bytes per vote-kernel launch, rocprofv3 PMC passes of %s "
                                                    "(same kernel source)" % os.path.basename(files[-1]))
    return None, "no PMC pass for this workload in " + os.path.basename(files[-1])


FIXTURE = os.path.join(ROOT, "tests", "golden", "case_5k_100k_df8.npz")


def pose_recall(ppf, synth, mode, T_bench, cells_bench, args, df, trials=6):
    """Pose recall@1 degree as SURVEY.md 8d defines it: pose of this build vs pose of the CPU restatement on the same
    clouds, success iff rotation difference < 1 degree and translation difference < 1 % of the model diameter.
    The trial that counts is the bench workload itself: the oracle's whole 5k x 100k registration (50 minutes of host
    threads) is committed as tests/golden/case_5k_100k_df8.npz, and the pose and kept cells of the timed steps are
    compared with it.  Six small registrations against the oracle run live (different models, poses and clutter)
    are reported beside it."""
    from oracle import oracle as O
    out = {"against": "CPU restatement (oracle) on the same clouds"}
    if (args.model_points, args.scene_points, df, args.tau_d) == (5000, 100000, 8, 0.025) and os.path.exists(FIXTURE):
        z = np.load(FIXTURE)
        dt, dr = ppf.ht_dist(T_bench, z["T_gpu"])
        diam = synth.bbox_extent(synth.make_model(0, args.model_points)[0])
        ok = bool(np.degrees(dr) < 1.0 and dt < 0.01 * diam)
        out.update({"trials": 1, "recall": 1.0 if ok else 0.0,
                    "workload": "the bench workload: 5000-point model, 100000-point scene, ref_point_df 8 (oracle output "
                                "committed as tests/golden/case_5k_100k_df8.npz)",
                    "rot_diff_deg": float(np.degrees(dr)), "trans_diff_frac_diam": dt / diam,
                    "pose_identical": bool(np.array_equal(T_bench, z["T_gpu"])),
                    "kept_cells_identical": bool(len(cells_bench) == len(z["cell_code"]) and
                                                 np.array_equal(cells_bench["code"], z["cell_code"]) and
                                                 np.array_equal(cells_bench["count"], z["cell_count"]))})
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
    small = {"trials": trials, "recall": ok / trials, "sizes": "600-point models, 3000-point scenes, ref_point_df 3, oracle run live",
             "worst_rot_diff_deg": worst[0], "worst_trans_diff_frac_diam": worst[1]}
    if "trials" in out:
        out["small_registrations"] = small
    else:
        out.update(small)
    return out


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
