"""Multi-GPU exchange step of the PPF path: one process per GPU, scene reference
points dealt round-robin to ranks (include/oslam.h: params.shard_rank/shard_world),
no collective on the vote path.  After the local vote kernels every rank holds its
peak records (count > threshold * local maximum) and its local maximum; this module
does the only exchange: an all-reduce(MAX) of the maxima and an all-gather of the records above
the global threshold (16 B per record; 4 KiB to a few hundred KiB per rank).  On GPUs the exchange
lives behind the C-ABI (oslam_align_multi: RCCL over xGMI, device buffers end to end; make_comm
below only carries the communicator id to the ranks); the host-buffer form of the same steps over
torch.distributed ("gloo") is what tests/test_distributed.py runs on CPUs.
The reference has no multi-GPU code (src/cuda/ppf.cu:45 picks one device)."""
import numpy as np

from . import ppf

MIN_BLOCK = 256       # smallest all-gather block, in records (4 KiB)


def make_comm(device_index):
    """RCCL communicator for oslam_align_multi: rank 0's id travels through torch.distributed (any backend), then
    every rank joins with ncclCommInitRank inside the library.  Every rank enters the same two collectives whatever
    fails where: rank 0 ALWAYS broadcasts (the id, or None when it could not make one), and a MIN all-reduce
    tells every rank whether all of them have a communicator.  Returns (comm, None) or (None, reason) -- the same
    kind on every rank."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    box, why = [None], None
    if rank == 0:
        try:
            box[0] = ppf.Comm.unique_id()
        except Exception as e:                      # noqa: BLE001 -- sent on as None
            why = "rank 0: %s" % e
    dist.broadcast_object_list(box, src=0)
    comm = None
    if box[0] is not None:
        try:
            comm = ppf.Comm(box[0], rank, world, device_index)
        except Exception as e:                      # noqa: BLE001 -- agreed on below
            why = "rank %d: %s" % (rank, e)
    elif why is None:
        why = "rank 0 could not make a communicator id"
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if comm is not None:
            comm.close()
        return None, why or "another rank could not join the communicator"
    return comm, None


def align_multi(model, scene, comm):
    """The whole multi-GPU registration of one model in one C call per rank (device buffers end to end)."""
    return model.align_multi(scene, comm, allow_no_votes=True)


def align_sharded_host(model, scene, device, vote_count_threshold=0.4):
    """The same exchange through host buffers and torch.distributed collectives (gloo on CPUs, tests):
    votes of this rank, all-reduce(MAX) of the maxima, then the records above the GLOBAL threshold --
    all of them, whatever their number -- all-gathered in fixed-size blocks.  Returns (union, global max)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    _, lmax = model.align_local(scene)                                    # records stay with the model
    meta = torch.tensor([int(lmax)], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    gmax = int(meta.item())
    mine = model.local_peaks(gmax)
    return gather_records(mine, device, world), gmax


def gather_records(mine, device, world):
    """All-gather of variable-length record lists in fixed-size blocks (16 B per record)."""
    import torch
    import torch.distributed as dist

    meta = torch.tensor([len(mine)], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    block = MIN_BLOCK
    while block < int(meta.item()):
        block *= 2
    host = np.zeros(block * 2, np.int64)
    if len(mine):
        host[: 2 * len(mine)] = np.ascontiguousarray(mine).view(np.int64).reshape(-1)
    send = torch.from_numpy(host).to(device)
    recv = torch.zeros(world * block * 2, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(recv, send)
    allrec = recv.cpu().numpy().view(ppf.CELL_DTYPE)
    return allrec[allrec["count"] > 0].copy()


def gather_peaks(cells, local_max, device, vote_count_threshold=0.4):
    """cells: this rank's peak records (ppf.CELL_DTYPE, count > threshold * local maximum), complete.
    Returns (union over ranks of the records with count > threshold * GLOBAL maximum, global maximum)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    meta = torch.tensor([int(local_max)], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    gmax = int(meta.item())
    bound = np.float32(vote_count_threshold) * np.float32(gmax)          # float compare, as model.cu:164-167
    mine = np.ascontiguousarray(cells[cells["count"].astype(np.float32) > bound])
    return gather_records(mine, device, world), gmax


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


def register_database_by_model(models, scene, device, n_total=None, comm=None):
    """The other way to use several GPUs on a large model database (SURVEY 8e's alternative): the MODELS are
    dealt to the ranks, every rank registers its share against the whole scene (ppf.Database: one scene pass
    per d_dist group) and only the 4x4 poses travel.  `models`: this rank's ppf.Model objects, model j of the
    database living on rank j % world; `scene`: unsharded.  Returns poses [n_total, 4, 4] on every rank.

    With a communicator (ppf.Comm: RCCL, or the loopback of the tests) the whole thing is one C call,
    oslam_db_align_multi.  Without one the poses travel through torch.distributed (gloo on CPUs); a rank without
    models still enters every collective."""
    import torch
    import torch.distributed as dist

    if comm is not None:
        db = ppf.Database(models) if len(models) else None
        try:
            T, found = ppf.db_align_multi(db, scene, comm, n_total)
        finally:
            if db is not None:
                db.close()
        return T[found != 0] if n_total is None else T
    world = dist.get_world_size()
    T = np.zeros((0, 4, 4), np.float32)
    if len(models):
        db = ppf.Database(models)
        T, _ = db.align(scene)
        db.close()
    n_max = torch.tensor([len(models)], dtype=torch.int64, device=device)
    dist.all_reduce(n_max, op=dist.ReduceOp.MAX)
    block = int(n_max.item())
    send = torch.zeros(max(1, block * 17), dtype=torch.float32, device=device)
    if len(models):
        flat = np.concatenate([np.ones((len(models), 1), np.float32), T.reshape(len(models), 16)], axis=1)
        send[: flat.size] = torch.from_numpy(flat.reshape(-1)).to(device)
    recv = torch.zeros(world * max(1, block * 17), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(recv, send)
    rec = recv.cpu().numpy().reshape(world, -1)[:, : block * 17].reshape(world, block, 17)
    out = []
    for k in range(block):                       # model j = k * world + r lives on rank r as its k-th
        for r in range(world):
            if rec[r, k, 0] == 1.0:
                out.append(rec[r, k, 1:].reshape(4, 4))
    return np.array(out, np.float32).reshape(-1, 4, 4)
