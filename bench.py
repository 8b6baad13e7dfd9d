#!/usr/bin/env python3
"""bench.py -- decoded PCM Msamples/s of the MI355X-native MLP decode path.

One step = one pass of the hot path over one batch of synthetic MLP titles that is
already resident in HBM: frame index (major-sync scan, size-chain walk, prefix
sums) + fused segment decode (parse, FIR/IIR, rematrix, output shift, RIFF order)
into planar int32 PCM, also in HBM.  Workload = BASELINE.json configs[2]: synthetic
6-ch / 96 kHz / 24-bit MLP, 2 decorrelation matrices + 8-tap FIR, codebook 1, parity
and CRC-8 on, major sync + restart header every 8 access units (BASELINE.md recipe).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

N > 1: the title list is sharded (each rank decodes its own titles; no data-path
collective), one tiny RCCL all-reduce sums the per-rank totals.  Scaling is weak.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=1024, help="unique synthetic titles per GPU")
    ap.add_argument("--aus", type=int, default=512, help="access units per title")
    ap.add_argument("--replicas", type=int, default=4,
                    help="device-side copies of the unique title set (distinct addresses)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--verify", type=int, default=8, help="titles checked against the oracle")
    ap.add_argument("--substreams", type=int, default=1, choices=(1, 2),
                    help="1 = the BASELINE metric; 2 = the recipe's 2-substream variant (ch 0-1 | ch 2-5), "
                         "decoded with two lanes per segment")
    ap.add_argument("--assignment", type=int, default=12,
                    help="channel assignment of the synthetic titles (12 = 6-ch, the BASELINE metric; "
                         "1 = 2-ch for configs[1] exploration)")
    ap.add_argument("--layout", default="interleaved", choices=("interleaved", "planar"),
                    help="PCM layout written by the decode: interleaved = frame-major, the order the "
                         "reference's dvda_read() hands out (default, faster: one contiguous run per lane "
                         "and flush); planar = the order its decode_packet appends to `samples`")
    return ap.parse_args()


def cpu_baseline(syn, streams_sample, frames_sample, nch, assignment, rate_code, budget_s):
    """Times the CPU decoder on this host, 1 thread, on a bounded sample of the same
    titles.  Uses the compiled reference when oracle/_ref travelled with the repo
    (kind 'reference'), else this repo's C restatement (kind 'port')."""
    from tests import oracle_lib
    kind = "port"
    dec = None
    if oracle_lib.Reference.available():
        try:
            ref = oracle_lib.Reference()
            dec = lambda b, f: ref.decode(b, assignment, rate_code, 2, f)[1]
            kind = "reference"
        except OSError:
            dec = None
    if dec is None:
        ora = oracle_lib.Oracle()
        dec = lambda b, f: ora.decode(b, nch, f)[1]
    samples = 0
    used = 0
    t0 = time.perf_counter()
    for b, f in zip(streams_sample, frames_sample):
        r = dec(b, f)
        samples += r * nch
        used += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(samples / dt / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": kind,
            "sample": "%d titles x %d PCM frames x %d ch of the bench workload, %.1f s" % (
                used, frames_sample[0], nch, dt)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Plumbing check on a 1-GPU box only (never what the driver measures): DVDA_BENCH_ONE_GPU=1 puts
    # every rank on cuda:0 and DVDA_BENCH_BACKEND=gloo exchanges the summary on the host, because RCCL
    # refuses two ranks on one device.
    backend = os.environ.get("DVDA_BENCH_BACKEND", "nccl")
    if os.environ.get("DVDA_BENCH_ONE_GPU") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the decode path is HIP-only)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import libdvd_audio_amd as pkg
    syn, hip = pkg.synth, pkg.hipdec

    assignment, rate_code = args.assignment, 1
    nch = syn.channels(assignment)
    rpa = syn.rows_per_au(rate_code)
    cfg = syn.make_cfg(assignment=assignment, rate_code=rate_code, n_substreams=args.substreams, n_aus=args.aus)

    # ---- synthetic titles: unique set generated on the host cores, replicated on the device
    t_gen = time.perf_counter()
    flat, offs, sizes, frames = syn.batch(cfg, 1 + rank * args.streams, args.streams)
    t_gen = time.perf_counter() - t_gen
    unique_bytes = int(len(flat) - 64)
    R = max(1, args.replicas)
    n_streams = args.streams * R
    d_unique = torch.from_numpy(flat[:unique_bytes]).to(dev)
    d_bytes = torch.zeros(unique_bytes * R + 64, dtype=torch.uint8, device=dev)
    for r in range(R):
        d_bytes[r * unique_bytes:(r + 1) * unique_bytes] = d_unique
    del d_unique
    all_off = np.concatenate([offs + np.uint64(r * unique_bytes) for r in range(R)]).astype(np.int64)
    all_len = np.tile(sizes, R).astype(np.int64)
    all_frames = np.tile(frames, R).astype(np.int64)
    total_bytes = unique_bytes * R
    comp_bytes = int(all_len.sum())
    rows_total = int(all_frames.sum())
    samples_per_step = rows_total * nch
    out_off = np.zeros(n_streams, np.int64)
    out_off[1:] = np.cumsum(all_frames[:-1] * nch)
    n_segments = n_streams * ((args.aus + cfg.restart_interval - 1) // cfg.restart_interval)

    d_off = torch.from_numpy(all_off).to(dev)
    d_len = torch.from_numpy(all_len).to(dev)
    d_out_off = torch.from_numpy(out_off).to(dev)
    d_stride = torch.from_numpy(all_frames).to(dev)
    d_pcm = torch.empty(samples_per_step, dtype=torch.int32, device=dev)

    layout = hip.PCM_INTERLEAVED if args.layout == "interleaved" else hip.PCM_PLANAR
    ctx = hip.Context(local_rank, n_streams, n_segments, lanes_per_segment=args.substreams, layout=layout)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        ctx.index(d_bytes.data_ptr(), total_bytes, d_off.data_ptr(), d_len.data_ptr(), n_streams, stream)
        ctx.decode(d_pcm.data_ptr(), d_out_off.data_ptr(), d_stride.data_ptr(), stream)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.kernel_time()  # drop warmup launches
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = ctx.kernel_time()

    # ---- every title must have decoded cleanly, and a sample must match the oracle bit for bit
    infos = ctx.stream_info(stream=stream)
    bad = [(i, hex(inf.status)) for i, inf in enumerate(infos) if inf.status != 0 or
           inf.pcm_frames != all_frames[i]]
    if bad:
        raise SystemExit("decode reported errors: %s" % bad[:8])
    bit_exact = None
    if args.verify:
        from tests import oracle_lib
        ora = oracle_lib.Oracle()
        bit_exact = True
        pick = np.linspace(0, n_streams - 1, num=min(args.verify, n_streams), dtype=np.int64)
        for i in pick:
            u = int(i % args.streams)
            b = flat[int(offs[u]):int(offs[u] + sizes[u])]
            want, r, st = ora.decode(b, nch, int(frames[u]))
            got = d_pcm[int(out_off[i]):int(out_off[i]) + int(all_frames[i]) * nch].cpu().numpy()
            got = got.reshape(int(all_frames[i]), nch).T if layout == hip.PCM_INTERLEAVED \
                else got.reshape(nch, int(all_frames[i]))
            bit_exact = bit_exact and st == 0 and np.array_equal(got, want)
        if not bit_exact:
            raise SystemExit("HIP decode differs from the oracle")

    # ---- whole-job aggregate: the path's one collective is this summary (RCCL all-reduce of a
    #      few words over xGMI; nothing on the data path is exchanged)
    checksum = int(d_pcm.to(torch.int64).sum().item()) if args.verify else 0
    summ = pkg.shard.reduce_summary(dist if world > 1 else None, dev if backend == "nccl" else torch.device("cpu"),
                                    rows_total, samples_per_step,
                                    comp_bytes, 0, checksum, elapsed)
    elapsed_max = summ["seconds"]
    job_samples = float(summ["samples"])

    if rank == 0:
        ms_per_step = elapsed_max / args.steps * 1e3
        value = job_samples * args.steps / elapsed_max / 1e6
        algo_bytes = comp_bytes + 4 * samples_per_step      # per launch, this rank
        # HBM-side bytes of one k_decode launch from the rocprofv3 PMC passes of THIS workload
        # (tools/prof_pmc.sh -> tools/pmc_traffic.py -> profiles/traffic.json); null when the
        # committed profile was taken on another workload size
        traffic_bytes = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if tj.get("samples_per_launch") == samples_per_step and tj.get("compressed_bytes") == comp_bytes \
                    and tj.get("pcm_layout", "planar") == args.layout:
                traffic_bytes = tj["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        out = {
            "metric": "decoded PCM Msamples/s (bit-exact) on 6ch/96k/24b MLP",
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32 (int64 accumulate)",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: synthetic 6ch/96kHz/24bit MLP, 2 matrices + 8-tap FIR, "
                            "codebook 1, CRC on, restart every 8 AUs",
                "substreams": args.substreams,
                "pcm_layout": "interleaved int32 [frame][channel] (reference dvda_read order)"
                              if layout == hip.PCM_INTERLEAVED else
                              "planar int32 [channel][frame] (reference decode_packet order)",
                "titles_per_gpu": n_streams, "unique_titles_per_gpu": args.streams,
                "access_units_per_title": args.aus, "segments_per_gpu": n_segments,
                "samples_per_step_per_gpu": samples_per_step,
                "compressed_bytes_per_gpu": comp_bytes,
                "parallelism": "titles sharded over %d GPU(s), no data-path collective" % world,
                "bit_exact_vs_oracle": bit_exact,
            },
            "roofline": {
                "kernel": "k_decode", "bound": "hbm",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic_bytes,
                "kernel_ms": round(kernel_ms, 4), "launches": launches,
                "algorithmic_bytes_per_launch": algo_bytes,
            },
        }
        if not args.no_cpu and world == 1:       # the CPU baseline is timed at N=1 only
            ns = args.streams
            sample = [flat[int(offs[i]):int(offs[i] + sizes[i])] for i in range(ns)]
            out["cpu_baseline"] = cpu_baseline(syn, sample, [int(f) for f in frames[:ns]], nch,
                                               assignment, rate_code, args.cpu_seconds)
        out["host"] = {"gen_seconds": round(t_gen, 2), "cpus": os.cpu_count()}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
