"""world_size-2 CPU test of the N>1 path: title sharding + the summary all-reduce over gloo.
Each rank "decodes" its shard with the CPU oracle (the GPU kernels are covered by -m gpu);
what is under test is that the shards are disjoint, cover every title, and that the reduced
summary equals the single-process totals."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import libdvd_audio_amd as pkg
    from tests import oracle_lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    syn = pkg.synth
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=8)
    titles = [syn.stream(cfg, 40 + i) for i in range(7)]
    sizes = [len(b) + 1000 * (i % 3) for i, (b, f) in enumerate(titles)]
    mine = pkg.shard.shard_titles(sizes, world, rank)
    ora = oracle_lib.Oracle()
    frames = samples = nbytes = errors = checksum = 0
    for i in mine:
        b, f = titles[i]
        pcm, r, st = ora.decode(b, 6, f)
        frames += r
        samples += pcm.size
        nbytes += len(b)
        errors += 1 if st else 0
        checksum += int(pcm.astype(np.int64).sum()) & 0xFFFFFFFF
    out = pkg.shard.reduce_summary(dist, torch.device("cpu"), frames, samples, nbytes, errors, checksum,
                                   0.1 * (rank + 1))
    q.put((rank, [int(i) for i in mine], out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_summary(pkg, oracle):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    owned = sorted(res[0][1] + res[1][1])
    assert owned == list(range(7))                      # disjoint cover
    assert set(res[0][1]).isdisjoint(res[1][1])
    assert res[0][2] == res[1][2]                       # every rank holds the same summary
    # single-process totals
    syn = pkg.synth
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=8)
    frames = samples = nbytes = checksum = 0
    for i in range(7):
        b, f = syn.stream(cfg, 40 + i)
        pcm, r, st = oracle.decode(b, 6, f)
        assert st == 0
        frames += r
        samples += pcm.size
        nbytes += len(b)
        checksum += int(pcm.astype(np.int64).sum()) & 0xFFFFFFFF
    s = res[0][2]
    assert (s["pcm_frames"], s["samples"], s["compressed_bytes"], s["errors"]) == (frames, samples, nbytes, 0)
    assert s["checksum"] == checksum
    assert abs(s["seconds"] - 0.2) < 1e-9               # max over ranks


def test_shard_is_balanced_and_deterministic(pkg):
    rng = np.random.RandomState(0)
    sizes = rng.randint(1000, 100000, size=101)
    parts = [pkg.shard.shard_titles(sizes, 8, r) for r in range(8)]
    allidx = np.sort(np.concatenate(parts))
    assert np.array_equal(allidx, np.arange(101))
    loads = np.array([sizes[p].sum() for p in parts])
    assert loads.max() - loads.min() <= sizes.max()
    again = pkg.shard.shard_titles(sizes, 8, 3)
    assert np.array_equal(again, parts[3])
    assert len(pkg.shard.shard_titles([], 4, 1)) == 0
