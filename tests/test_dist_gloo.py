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
    # "bit-exact on every rank": one rank that failed its check makes it false, one rank that checked nothing makes it
    # unknown (None) -- never true by default
    one_bad = pkg.shard.reduce_summary(dist, torch.device("cpu"), 0, 0, 1, 0, 0, 0.1, verified=rank != 1)
    one_blind = pkg.shard.reduce_summary(dist, torch.device("cpu"), 0, 0, 1, 0, 0, 0.1, checked=rank != 1)
    out["_one_bad"] = one_bad["all_verified"]
    out["_one_blind"] = one_blind["all_verified"]
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
    assert s["all_verified"] is True and s["_one_bad"] is False and s["_one_blind"] is None
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


def _run_bench(extra, env_extra):
    import json
    import subprocess
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True,
                       env=env, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r.returncode, (json.loads(lines[-1]) if lines else None), r.stderr


def test_bench_launcher_starts_two_ranks_over_gloo():
    """`python bench.py --gpus 2` with no torchrun environment must start 2 ranks itself.  On this box
    (no GPU) the ranks run in plumbing mode: rank environment, shard and the summary all-reduce are the
    real code, nothing is decoded (value null)."""
    rc, rec, err = _run_bench(["--gpus", "2", "--streams", "12", "--aus", "8"], {"DVDA_BENCH_PLUMBING": "1"})
    assert rc == 0, err[-2000:]
    assert rec["n_gpus"] == 2 and rec["plumbing_only"] is True and rec["value"] is None
    assert rec["scaling"] == "weak"
    assert rec["config"]["titles_all_ranks"] == 24               # weak: every rank brings its own 12 titles
    assert rec["config"]["samples_all_ranks"] == 24 * 8 * 80 * 6
    assert abs(rec["seconds_max_over_ranks"] - 0.002) < 1e-9     # max over ranks, not rank 0's
    # ... and what the N > 1 line says about the shard itself: fastest rank, balance, every rank's own check
    # (plumbing mode decodes nothing, so nothing was compared: "not checked" is null, never true)
    assert abs(rec["ranks"]["seconds_min"] - 0.001) < 1e-9 and rec["ranks"]["bit_exact_on_every_rank"] is None
    assert rec["ranks"]["compressed_bytes_max"] >= rec["ranks"]["compressed_bytes_min"] > 0
    assert 1.0 <= rec["ranks"]["load_imbalance"] < 1.2
    # (round 5) the line says by itself that two ranks took part and what each measured: the process group's size, the
    # all-reduced count of ranks, one entry per rank (plumbing mode: the step time is the rank's stand-in 1 / 2 ms)
    assert rec["ranks"]["world_size"] == 2 and rec["ranks"]["ranks_counted"] == 2
    per = rec["ranks"]["per_rank"]
    assert [p["rank"] for p in per] == [0, 1]
    assert all(set(p) >= {"rank", "k_decode_ms", "roofline_frac", "ms_per_step"} for p in per)
    assert [p["ms_per_step"] for p in per] == [1.0, 2.0]


def test_bench_c4_shards_the_1024_units_strongly():
    """configs[3]: the single-access-unit streams are dealt to the ranks (strong scaling): the ranks'
    shares add up to the whole list whatever N is."""
    one = _run_bench(["--gpus", "1", "--workload", "c4", "--streams", "96"], {"DVDA_BENCH_PLUMBING": "1"})
    two = _run_bench(["--gpus", "2", "--workload", "c4", "--streams", "96"], {"DVDA_BENCH_PLUMBING": "1"})
    assert one[0] == 0 and two[0] == 0, (one[2][-1000:], two[2][-1000:])
    assert one[1]["scaling"] == two[1]["scaling"] == "strong"
    for k in ("titles_all_ranks", "samples_all_ranks", "compressed_bytes_all_ranks"):
        assert one[1]["config"][k] == two[1]["config"][k], k
    assert one[1]["config"]["titles_all_ranks"] == 96 and one[1]["config"]["samples_all_ranks"] == 96 * 480


def test_bench_refuses_more_gpus_than_there_are():
    """Without the plumbing switch `--gpus 2` on a box with fewer GPUs fails cleanly (never a silent
    one-rank run that prints n_gpus: 1)."""
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("this box has 2 GPUs")
    rc, rec, err = _run_bench(["--gpus", "2"], {"DVDA_BENCH_PLUMBING": "0"})
    assert rc == 2 and rec is None and "needs 2 GPUs" in err
