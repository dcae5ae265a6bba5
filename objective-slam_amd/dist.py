"""Multi-GPU exchange step of the PPF path: one process per GPU, scene reference
points dealt round-robin to ranks (include/oslam.h: params.shard_rank/shard_world),
no collective on the vote path.  After the local vote kernels every rank holds its
peak records (count > threshold * local maximum) and its local maximum; this module
does the only exchange: an all-reduce(MAX) of the maxima and an all-gather of
fixed-size record blocks (16 B per record; 4 KiB to a few hundred KiB per rank).  Backend "nccl" is RCCL over
xGMI on the GPU node; "gloo" runs the same code on CPUs (tests/test_distributed.py).
The reference has no multi-GPU code (src/cuda/ppf.cu:45 picks one device)."""
import numpy as np

from . import ppf

LOCAL_CAP = 1 << 18   # records a rank may hold before the global threshold is known
MIN_BLOCK = 256       # smallest all-gather block, in records (4 KiB)


def gather_peaks(cells, local_max, device, vote_count_threshold=0.4):
    """cells: this rank's peak records (ppf.CELL_DTYPE, count > threshold * local maximum).
    Returns (union over ranks of the records with count > threshold * GLOBAL maximum, global maximum).

    Three collectives, all tiny: all-reduce(MAX) of the vote maximum (the reference's threshold is
    global, model.cu:164-170); all-reduce(MAX) of the number of surviving records, which sizes the
    blocks; all-gather of one fixed-size block of 16-byte records per rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    meta = torch.tensor([int(local_max)], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    gmax = int(meta.item())
    bound = np.float32(vote_count_threshold) * np.float32(gmax)          # float compare, as model.cu:164-167
    mine = np.ascontiguousarray(cells[cells["count"].astype(np.float32) > bound])
    meta = torch.tensor([len(mine)], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    block = MIN_BLOCK
    while block < int(meta.item()):
        block *= 2
    host = np.zeros(block * 2, np.int64)
    if len(mine):
        host[: 2 * len(mine)] = mine.view(np.int64).reshape(-1)
    send = torch.from_numpy(host).to(device)
    recv = torch.zeros(world * block * 2, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(recv, send)                               # per-GPU top pose votes
    allrec = recv.cpu().numpy().view(ppf.CELL_DTYPE)
    return allrec[allrec["count"] > 0].copy(), gmax


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
