"""Multi-GPU exchange step of the PPF path: one process per GPU, scene reference
points dealt round-robin to ranks (include/oslam.h: params.shard_rank/shard_world),
no collective on the vote path.  After the local vote kernels every rank holds its
peak records (count > threshold * local maximum) and its local maximum; this module
does the only exchange: an all-reduce(MAX) of the maxima and an all-gather of
fixed-size record blocks (RECORD_CAP x 16 B per rank).  Backend "nccl" is RCCL over
xGMI on the GPU node; "gloo" runs the same code on CPUs (tests/test_distributed.py).
The reference has no multi-GPU code (src/cuda/ppf.cu:45 picks one device)."""
import numpy as np

from . import ppf

RECORD_CAP = 4096   # strongest peaks kept per rank; 64 KiB per rank on the wire


def gather_peaks(cells, local_max, device, cap=RECORD_CAP):
    """cells: this rank's peak records (ppf.CELL_DTYPE, strongest first).
    Returns (union of all ranks' records, global maximum)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    host = np.zeros(cap * 2, np.int64)
    n = min(len(cells), cap)
    if n:
        host[: 2 * n] = np.ascontiguousarray(cells[:n]).view(np.int64).reshape(-1)
    send = torch.from_numpy(host).to(device)
    recv = torch.zeros(world * cap * 2, dtype=torch.int64, device=device)
    gmax = torch.tensor([int(local_max)], dtype=torch.int64, device=device)
    dist.all_reduce(gmax, op=dist.ReduceOp.MAX)          # global vote maximum (model.cu:164 needs it)
    dist.all_gather_into_tensor(recv, send)              # per-GPU top pose votes
    allrec = recv.cpu().numpy().view(ppf.CELL_DTYPE)
    return allrec[allrec["count"] > 0].copy(), int(gmax.item())


def finish_on_host(cells, global_max, m_pts, m_nrm, s_pts, s_nrm, d_dist, vote_count_threshold=0.4,
                   cpu_clustering=False, use_l1_norm=False, use_averaged_clusters=False):
    """Host stage on the gathered union without a GPU handle: final threshold, canonical
    order, poses, clustering (what oslam_align_finish does)."""
    import ctypes as C
    L = ppf.lib()
    buf = np.ascontiguousarray(cells, ppf.CELL_DTYPE).copy()
    n = L.oslam_filter_cells(buf.ctypes.data_as(C.c_void_p), len(buf), float(vote_count_threshold), int(global_max))
    buf = buf[:n].copy()
    L.oslam_sort_cells(buf.ctypes.data_as(C.c_void_p), len(buf))
    T, _ = ppf.pose_stage(buf, m_pts, m_nrm, s_pts, s_nrm, d_dist, cpu_clustering, use_l1_norm,
                          use_averaged_clusters, allow_no_votes=True)
    return T, buf
