"""Multi-GPU partition of the path: titles (independent MLP streams) are dealt to ranks; the only
exchange is the batch summary.  One process per GPU, torch.distributed ("nccl" = RCCL over xGMI
on the GPU node, "gloo" in the CPU tests) -- no data-path collective exists or is needed."""
import numpy as np


def shard_titles(sizes, world, rank):
    """Greedy longest-processing-time assignment of titles to ranks by compressed size
    (SURVEY.md 8(e)).  Deterministic: every rank computes the same partition.
    Returns the sorted indices of the titles owned by `rank`."""
    sizes = np.asarray(sizes, np.int64)
    order = np.argsort(-sizes, kind="stable")
    load = np.zeros(world, np.int64)
    owner = np.empty(len(sizes), np.int64)
    for i in order:
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += sizes[i]
    return np.flatnonzero(owner == rank)


def reduce_summary(dist, device, pcm_frames, samples, comp_bytes, errors, checksum, seconds, verified=True, checked=True,
                   per_rank=()):
    """The path's one collective: all ranks learn {sum frames, sum samples, sum bytes, sum errors,
    xor-free additive checksum}, the slowest and the fastest rank's time, the largest and the smallest rank's share
    of the compressed bytes (the load balance of the shard) and whether EVERY rank's bit-exact check passed.  `dist` is torch.distributed (already
    initialised) or None for a single process.  `per_rank`: a few numbers of this rank (its kernel's time, its roofline
    fraction ...) that every rank learns of every rank (one all_gather): returned as `per_rank[r]`; `ranks_counted` is
    the sum of one per rank, which has to equal dist.get_world_size() -- a line from N ranks says so itself."""
    import torch
    # (counters are integers and are summed as integers: float64 is exact only to 2^53)
    tot = torch.tensor([int(pcm_frames), int(samples), int(comp_bytes), int(errors), 1], dtype=torch.int64, device=device)
    chk = torch.tensor([int(checksum) & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)
    tmax = torch.tensor([float(seconds), float(comp_bytes)], dtype=torch.float64, device=device)
    # (the fastest rank's time, the smallest share, and "every rank's sample check passed" travel as minima)
    # ("checked": a rank that compared nothing -- profiling runs -- must not pass as verified)
    tmin = torch.tensor([float(seconds), float(comp_bytes), 1.0 if verified else 0.0, 1.0 if checked else 0.0],
                        dtype=torch.float64, device=device)
    mine = torch.tensor([float(x) for x in per_rank] or [0.0], dtype=torch.float64, device=device)
    gathered = [mine]
    world = 1
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        world = dist.get_world_size()
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(chk, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
    return {"pcm_frames": int(tot[0].item()), "samples": int(tot[1].item()),
            "compressed_bytes": int(tot[2].item()), "errors": int(tot[3].item()),
            "checksum": int(chk[0].item()), "seconds": float(tmax[0].item()),
            "seconds_min": float(tmin[0].item()), "bytes_max": int(tmax[1].item()), "bytes_min": int(tmin[1].item()),
            "all_verified": bool(tmin[2].item() > 0.5) if tmin[3].item() > 0.5 else None,
            "all_checked": bool(tmin[3].item() > 0.5),
            "world_size": world, "ranks_counted": int(tot[4].item()),
            "per_rank": [[float(v) for v in g.tolist()] for g in gathered] if per_rank else []}
