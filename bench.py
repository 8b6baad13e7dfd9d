#!/usr/bin/env python3
"""bench.py -- decoded PCM Msamples/s of the MI355X-native MLP decode path.

One step = one pass of the hot path over one batch of synthetic MLP titles that is
already resident in HBM: frame index (major-sync scan, size-chain walk, prefix
sums) + fused segment decode (parse, FIR/IIR, rematrix, output shift, RIFF order)
into int32 PCM, also in HBM.

    python bench.py [--gpus N --steps K --warmup W] [--workload c3|c4]

Workloads (BASELINE.json `configs`):
  c3 (default, the headline): configs[2] -- synthetic 6-ch / 96 kHz / 24-bit MLP, 2 decorrelation
      matrices + 8-tap FIR, code book 1, parity + CRC-8 on, major sync + restart header every 8 access
      units.  Weak scaling: every rank decodes its own 4 096 titles.
  c4: configs[3] -- 1 024 independent single-access-unit streams of the same recipe, dealt to the ranks
      with shard.shard_titles (strong scaling); reports steady-state Msamples/s and single-batch latency.

N > 1: `python bench.py --gpus N` starts N ranks itself (one process per GPU through
torch.distributed.run, decided before anything touches the GPU); under torchrun (RANK / WORLD_SIZE
in the environment) it is one of the ranks.  The title list is sharded, no data-path collective; one
tiny RCCL all-reduce sums the per-rank totals and takes the slowest rank's time.

At N = 1 the JSON line also carries: the CPU baseline on 1 core and on all host cores (the compiled
reference when oracle/_ref travelled with the repo), a bit-exact comparison of EVERY unique title of
the batch against that CPU decode, and sub-records for other shapes of the path ("sub").
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
SIMDS = 1024               # 256 CUs x 4 SIMDs
VALU_CYCLES_PER_WAVE_INST = 4   # a wave64 VALU instruction occupies the 16-lane pipe for 4 cycles
METRIC = "decoded PCM Msamples/s (bit-exact) on 6ch/96k/24b MLP"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3", choices=("c3", "c4"))
    ap.add_argument("--streams", type=int, default=1024, help="unique synthetic titles per GPU (c3) / in all (c4)")
    ap.add_argument("--aus", type=int, default=512, help="access units per title (c3)")
    ap.add_argument("--replicas", type=int, default=4,
                    help="device-side copies of the unique title set (distinct addresses; c3)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU baseline budget per leg")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline and the full-size check")
    ap.add_argument("--no-sub", action="store_true", help="skip the sub-records (other shapes of the path)")
    ap.add_argument("--no-disc", action="store_true", help="skip sub.disc_tier (a synthetic disc extracted by build/dvda2wav_hip)")
    ap.add_argument("--chain-flight", type=int, default=0,
                    help="diagnostic: also time sub.chained_two_substreams with this many decode contexts in flight")
    ap.add_argument("--only-sub", default="", help="comma-separated sub-record names: run only those (profiling)")
    ap.add_argument("--verify", type=int, default=8, help="titles checked against the oracle when the full check is off")
    ap.add_argument("--substreams", type=int, default=1, choices=(1, 2),
                    help="1 = the BASELINE metric; 2 = the recipe's 2-substream variant (ch 0-1 | ch 2-5)")
    ap.add_argument("--assignment", type=int, default=12,
                    help="channel assignment of the synthetic titles (12 = 6-ch, the BASELINE metric)")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="decode contexts in flight (each its own HIP stream and PCM buffer; a step is one index + one "
                         "decode of the whole batch on the next of them): 1 = strictly one step after the other")
    ap.add_argument("--layout", default="interleaved", choices=("interleaved", "planar"),
                    help="PCM layout written by the decode: interleaved = frame-major, the order the "
                         "reference's dvda_read() hands out (default); planar = the order its decode_packet "
                         "appends to `samples`")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------- launcher (N > 1)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`python bench.py --gpus N` without a torchrun environment: start N fresh rank processes.
    Runs before this process has touched the GPU (device_count() does not initialise it) and starts
    the ranks as children -- a process that has initialised HIP is never re-exec'ed."""
    plumbing = os.environ.get("DVDA_BENCH_PLUMBING") == "1" or os.environ.get("DVDA_BENCH_ONE_GPU") == "1"
    if not plumbing:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.stderr.write("bench.py: --gpus %d needs %d GPUs, %d visible\n" % (args.gpus, args.gpus, have))
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


# ----------------------------------------------------------------------------- CPU side (checker + baseline)
def usable_cpus():
    """-> (threads worth starting, logical CPUs in the affinity mask, cgroup quota in CPUs or None).  A
    container may see every logical CPU of the host and still be limited to a few CPUs' worth of time
    (cgroup cpu.max): more threads than that only add throttling."""
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    n = logical if quota is None else max(1, min(logical, int(quota + 0.5)))
    return n, logical, quota


def cpu_legs(flat, offs, sizes, frames, assignment, rate_code, nch, budget_s, n_threads):
    """Decodes the unique titles on the host with the CPU decoder the GPU result is compared with and timed
    against: the compiled reference (oracle/_ref/libdvda_ref.so, kind 'reference') when it travelled with
    the repo, else this repo's C restatement (kind 'port').  A pthread pool inside that library
    (oracle/cpu_pool.c) runs one decoder instance per title -- instances share nothing, SURVEY 8(b) --
    with no Python between two decodes: 1 thread for ~budget_s (a bounded sample of the titles), then
    n_threads threads over ALL titles (each at least once, then round and round until budget_s is used).
    -> (records, pcm [n_titles, nch, frames] int32 of every title)"""
    import ctypes
    from tests import oracle_lib
    lib, kind = None, "port"
    if oracle_lib.Reference.available():
        try:
            lib, kind = oracle_lib.Reference().lib, "reference"
        except OSError:
            lib = None
    if lib is None:
        lib = oracle_lib.Oracle().lib
    fn = lib.cpu_pool_decode
    fn.restype = ctypes.c_ulong
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32] + [ctypes.c_uint] * 4 + \
                  [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
    n = len(sizes)
    fmax = int(frames.max())
    if not (frames == fmax).all():
        raise SystemExit("cpu_legs expects titles of one length")
    pcm = np.empty((n, nch, fmax), np.int32)
    o64 = np.ascontiguousarray(offs, np.uint64)
    s64 = np.ascontiguousarray(sizes, np.uint64)
    secs = ctypes.c_double()

    def run(n_titles, threads, budget):
        done = fn(flat.ctypes.data, o64.ctypes.data, s64.ctypes.data, n_titles, 2, rate_code, assignment, nch,
                  pcm.ctypes.data, fmax, threads, budget, ctypes.byref(secs))
        if not done:
            raise SystemExit("CPU decoder (%s) failed on the bench titles" % kind)
        return int(done), float(secs.value)

    # ---- 1 core: a bounded sample (as many titles as ~budget_s allows, found with a short probe)
    d0, t0 = run(min(n, 8), 1, 0.0)
    per_title = t0 / d0
    n1 = int(max(8, min(n, budget_s / per_title)))
    d1, t1 = run(n1, 1, 0.0)
    rec1 = {"value": round(d1 * fmax * nch / t1 / 1e6, 3), "unit": "Msamples/s", "cores": 1, "kind": kind,
            "sample": "%d titles x %d PCM frames x %d ch of the bench workload, %.1f s" % (d1, fmax, nch, t1)}
    # ---- all cores
    dN, tN = run(n, n_threads, budget_s)
    recN = {"value": round(dN * fmax * nch / tN / 1e6, 3), "unit": "Msamples/s", "cores": n_threads, "kind": kind,
            "sample": "%d title decodes (all %d unique titles at least once) x %d PCM frames x %d ch, %d pthreads, "
                      "one decoder instance per title, %.1f s" % (dN, n, fmax, nch, n_threads, tN)}
    return [rec1, recN], pcm


# ----------------------------------------------------------------------------- one batch on one GPU
class Batch:
    """A set of MLP streams resident in HBM + the decode context and output buffers for it."""

    def __init__(self, pkg, torch, dev, local_rank, flat, offs, sizes, frames, nchs, replicas, layout, lanes,
                 n_segments, depth=1, chained=False):
        """depth > 1: a pipeline of `depth` decode contexts, each with its own HIP stream and PCM buffer, over the
        same resident input -- step i runs on slot i % depth with the non-blocking decode call, so the index of one
        batch overlaps the decode of the batch before it (what a production feeder does; a step is still one index
        + one decode of the WHOLE batch).  chained: the titles' restart points carry FIR taps, the chain passes'
        workspaces are reserved ahead (dvda_mlp_hip_reserve) so that the non-blocking call can decode them."""
        hip = pkg.hipdec
        self.hip, self.torch, self.dev = hip, torch, dev
        self.layout = {"interleaved": hip.PCM_INTERLEAVED, "planar": hip.PCM_PLANAR, "wav24": hip.PCM_WAV24}[layout]
        self.out_bytes = 3 if layout == "wav24" else 4          # per sample
        unique_bytes = int(len(flat) - 64)
        R = max(1, replicas)
        self.unique, self.R = len(sizes), R
        self.n_streams = len(sizes) * R
        d_unique = torch.from_numpy(flat[:unique_bytes]).to(dev)
        self.d_bytes = torch.zeros(unique_bytes * R + 64, dtype=torch.uint8, device=dev)
        for r in range(R):
            self.d_bytes[r * unique_bytes:(r + 1) * unique_bytes] = d_unique
        del d_unique
        self.all_off = np.concatenate([offs.astype(np.int64) + r * unique_bytes for r in range(R)])
        self.all_len = np.tile(sizes, R).astype(np.int64)
        self.all_frames = np.tile(frames, R).astype(np.int64)
        self.all_nch = np.tile(np.asarray(nchs, np.int64), R)
        self.total_bytes = unique_bytes * R
        self.comp_bytes = int(self.all_len.sum())
        self.rows_total = int(self.all_frames.sum())
        self.samples = int((self.all_frames * self.all_nch).sum())
        # (offsets in int32 units; a packed WAV payload takes 3/4 of them, every stream starts on 16 bytes)
        words = (self.all_frames * self.all_nch * self.out_bytes + 15) // 16 * 4
        # (diagnostic, DVDA_BENCH_OUT_PAD=bytes: this many bytes between one title's PCM and the next -- where the titles'
        #  regions start is the caller's choice, d_out_off)
        words = words + (int(os.environ.get("DVDA_BENCH_OUT_PAD", "0")) // 16) * 4
        self.out_off = np.zeros(self.n_streams, np.int64)
        self.out_off[1:] = np.cumsum(words[:-1])
        self.pcm_words = int(words.sum())
        self.d_off = torch.from_numpy(self.all_off).to(dev)
        self.d_len = torch.from_numpy(self.all_len).to(dev)
        self.d_out_off = torch.from_numpy(self.out_off).to(dev)
        self.d_stride = torch.from_numpy(self.all_frames).to(dev)
        self.n_segments = n_segments
        self.small = n_segments <= 4096
        self.depth = max(1, depth)
        self.ctxs, self._tstreams, self.streams, self.d_pcms = [], [], [], []
        for _ in range(self.depth):
            self.d_pcms.append(torch.empty(max(self.pcm_words, 1), dtype=torch.int32, device=dev))
            c = hip.Context(local_rank, self.n_streams, n_segments, lanes_per_segment=lanes, layout=self.layout)
            if self.depth > 1 or chained:
                c.reserve(self.rows_total if chained else 0, n_segments if chained else 0, 0)
            self.ctxs.append(c)
            # a stream of its own (not the legacy default stream): the library replays the index as a hipGraph when
            # the same buffers are indexed again, and capture needs a real stream
            ts = torch.cuda.Stream(dev)
            self._tstreams.append(ts)
            self.streams.append(ts.cuda_stream)
        self.ctx, self.d_pcm, self.stream = self.ctxs[0], self.d_pcms[0], self.streams[0]
        self._turn = 0
        self._done = [torch.cuda.Event() for _ in range(self.depth)]       # slot k's decode has finished
        torch.cuda.synchronize(dev)

    def step(self, n_streams=None):
        n = self.n_streams if n_streams is None else n_streams
        total = self.total_bytes if n == self.n_streams else int(self.all_off[n])
        k = self._turn % self.depth
        self._turn += 1
        ctx, st = self.ctxs[k], self.streams[k]
        ctx.index(self.d_bytes.data_ptr(), total, self.d_off.data_ptr(), self.d_len.data_ptr(), n, st)
        if self.depth > 1:
            # the decodes themselves run one after the other (two fast-pass kernels side by side only take each
            # other's issue slots): this slot's decode waits for the previous slot's; what overlaps is this
            # slot's INDEX -- bound by HBM -- with that decode -- bound by instruction issue
            # (a small batch -- one the cooperative kernel decodes, a few workgroups per compute unit -- leaves most
            #  of the device idle: there the slots' decodes run side by side as well)
            # (diagnostic, DVDA_BENCH_FREE_OVERLAP=1: no such edge -- the slots' decodes are left to the hardware's own
            #  scheduling, which is how the parse pass of one batch and the fused chain pass of another were measured
            #  side by side in round 6; tools/probe/r06_overlap.sh)
            if self._turn > 1 and not self.small and os.environ.get("DVDA_BENCH_FREE_OVERLAP", "0") != "1":
                self._tstreams[k].wait_event(self._done[(k - 1) % self.depth])
            ctx.decode_async(self.d_pcms[k].data_ptr(), self.d_out_off.data_ptr(), self.d_stride.data_ptr(), st)
            self._done[k].record(self._tstreams[k])
        else:
            ctx.decode(self.d_pcms[k].data_ptr(), self.d_out_off.data_ptr(), self.d_stride.data_ptr(), st)

    def kernel_time(self):
        """mean fast-pass kernel ms and launches over all slots since the last call"""
        tot, n, dtot = 0.0, 0, 0.0
        for c in self.ctxs:
            dms, dk = c.decode_time()           # (first: kernel_time resets the ring)
            ms, k = c.kernel_time()
            tot += ms * k
            dtot += dms * dk
            n += k
        self.last_decode_ms = dtot / n if n else 0.0    # whole decode call: every pass + the gaps between them
        return (tot / n if n else 0.0), n

    def sync(self):
        self.torch.cuda.synchronize(self.dev)

    def timed(self, steps, warmup, n_streams=None):
        """-> (seconds for `steps` back-to-back steps, mean k_decode ms, launches)"""
        for _ in range(warmup):
            self.step(n_streams)
        self.sync()
        self.kernel_time()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(n_streams)
        self.sync()
        dt = time.perf_counter() - t0
        kms, launches = self.kernel_time()
        return dt, kms, launches

    def check_status(self, n_streams=None, benign=0):
        n = self.n_streams if n_streams is None else n_streams
        for k in range(min(self.depth, max(self._turn, 1))):
            infos = self.ctxs[k].stream_info(n, stream=self.streams[k])
            bad = [(i, hex(inf.status), int(inf.pcm_frames)) for i, inf in enumerate(infos)
                   if (inf.status & ~benign) != 0 or inf.pcm_frames != self.all_frames[i]]
            if bad:
                raise SystemExit("decode reported errors (pipeline slot %d): %s" % (k, bad[:8]))
        return infos

    def slots_equal(self):
        """every pipeline slot decoded the batch to the same PCM (compared on the device)"""
        return all(bool(self.torch.equal(self.d_pcms[0], p)) for p in self.d_pcms[1:min(self.depth, self._turn)])

    def title(self, host, i):
        """PCM of stream i out of a host copy of d_pcm, as [channels, frames]"""
        f, c, o = int(self.all_frames[i]), int(self.all_nch[i]), int(self.out_off[i])
        a = host[o:o + f * c]
        return a.reshape(f, c).T if self.layout == self.hip.PCM_INTERLEAVED else a.reshape(c, f)

    def verify_sample(self, flat, offs, sizes, picks):
        """bit-exact vs the oracle (this repo's C restatement) for the picked streams"""
        from tests import oracle_lib
        ora = oracle_lib.Oracle()
        for i in picks:
            u = int(i) % self.unique
            b = flat[int(offs[u]):int(offs[u] + sizes[u])]
            want, r, st = ora.decode(b, int(self.all_nch[i]), int(self.all_frames[i]))
            o, n = int(self.out_off[i]), int(self.all_frames[i] * self.all_nch[i])
            f, c = int(self.all_frames[i]), int(self.all_nch[i])
            if self.layout == self.hip.PCM_WAV24:
                host = self.d_pcm[o:o + (n * 3 + 3) // 4].cpu().numpy().view(np.uint8)[:n * 3]
                if st != 0 or r != f or host.tobytes() != ora.wav_pack(want, 24):
                    return False
                continue
            host = self.d_pcm[o:o + n].cpu().numpy()
            got = host.reshape(f, c).T if self.layout == self.hip.PCM_INTERLEAVED else host.reshape(c, f)
            if st != 0 or r != f or not np.array_equal(got, want):
                return False
        return True

    def replicas_equal(self):
        """the R device-side copies of the unique set decode to the same PCM (compared on the device)"""
        per = self.pcm_words // self.R
        first = self.d_pcm[:per]
        return all(bool(self.torch.equal(first, self.d_pcm[r * per:(r + 1) * per])) for r in range(1, self.R))

    def close(self):
        for c in self.ctxs:
            c.close()
        del self.d_pcm, self.d_pcms, self.d_bytes


def roofline_record(b, kms, launches, traffic, valu_per_sample=None):
    algo = b.comp_bytes + 4 * b.samples
    ach = algo / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    rec = {"kernel": "k_decode", "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": traffic, "kernel_ms": round(kms, 4), "launches": launches,
           "algorithmic_bytes_per_launch": algo}
    if valu_per_sample:
        # the bound the counters point at: VALU issue.  SQ_INSTS_VALU (wave instructions per launch, from the
        # committed PMC profile of this workload) x 4 cycles each, spread over 1024 SIMDs at the shader clock
        rec["issue"] = valu_per_sample
    return rec


def committed_profile(samples, comp_bytes, layout):
    """HBM bytes and VALU instructions per k_decode launch from the rocprofv3 PMC passes of THIS workload
    (tools/prof_pmc.sh -> tools/pmc_traffic.py -> profiles/traffic.json); None when the committed profile
    was taken on another workload size.  A carried constant, labelled as such -- not measured in this run."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj.get("samples_per_launch") == samples and tj.get("compressed_bytes") == comp_bytes \
                and tj.get("pcm_layout", "planar") == layout:
            return tj
    except (OSError, ValueError, KeyError):
        pass
    return None


def sub_traffic(name, samples, comp_bytes):
    """counter traffic (FETCH_SIZE x 2 + WRITE_SIZE over the decode call's kernels) of a sub-record from the committed
    PMC passes, tools/pmc_sub.py -> profiles/sub_traffic.json; None when that file has no entry of this size"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "sub_traffic.json"))).get(name)
        if tj and tj.get("samples_per_step") == samples and tj.get("compressed_bytes") == comp_bytes:
            return {"hbm_bytes_per_step": tj["hbm_bytes_per_step"], "per_kernel": tj.get("per_kernel"),
                    "source": tj.get("source")}
    except (OSError, ValueError, KeyError):
        pass
    return None


def gen_mixed(syn, specs, seed0):
    """Concatenates syn.batch() outputs of several configurations into one flat buffer.
    specs: list of (cfg, n).  -> flat, offs, sizes, frames, nchs, n_segments"""
    flats, offs, sizes, frames, nchs = [], [], [], [], []
    pos = 0
    nseg = 0
    for k, (cfg, n) in enumerate(specs):
        f, o, s, fr = syn.batch(cfg, seed0 + 1000 * k, n)
        body = f[:len(f) - 64]
        flats.append(body)
        offs.append(o.astype(np.int64) + pos)
        sizes.append(s.astype(np.int64))
        frames.append(fr.astype(np.int64))
        nchs.append(np.full(n, syn.channels(cfg.assignment), np.int64))
        pos += len(body)
        nseg += n * ((cfg.n_aus + cfg.restart_interval - 1) // cfg.restart_interval + 1)
    flat = np.concatenate(flats + [np.zeros(64, np.uint8)])
    return (flat, np.concatenate(offs), np.concatenate(sizes), np.concatenate(frames), np.concatenate(nchs), nseg)


# ----------------------------------------------------------------------------- sub-records (N = 1)
def disc_tier(pkg, oracle_mod=None, tracks=8, aus=65536):
    """The disc tier end to end, wall clock: a synthetic disc (IFO + one AOB file, `tracks` MLP tracks of `aus` access
    units, 6-ch / 96 kHz / 24-bit) extracted by build/dvda2wav_hip -- process start, HIP runtime start, file read,
    demux + index + decode in windows, WAV write.  The second of two runs (the first one warms the page cache), the
    first track's WAV payload compared with the oracle's.  Never `value`: it is host I/O and a process's fixed cost as
    much as the decode.  tools/disc_bench.py is the same with the reference's own tool beside it."""
    import re
    import tempfile
    syn, disc = pkg.synth, pkg.disc
    rec = {"unit": "Msamples/s", "tracks": tracks, "access_units_per_track": aus}
    try:
        tool = pkg._build.build_tool()
        with tempfile.TemporaryDirectory() as tmp:          # (the default temporary directory, as tools/disc_bench.py)
            cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=aus)
            tr, samples, first = [], 0, None
            for t in range(tracks):
                b, f = syn.stream(cfg, 100 + t)
                if t == 0:
                    first = (b, f)
                tr.append({"sectors": disc.mlp_track_sectors(b), "pcm_frames": f, "rate_code": 1})
                samples += f * 6
            ats = disc.write_disc_titles(tmp, [tr])
            rec["aob_bytes"] = os.path.getsize(os.path.join(ats, "ATS_01_1.AOB"))
            rec["samples"] = samples
            env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "libdvd-audio_amd") + ":/opt/rocm/lib:" +
                       os.environ.get("LD_LIBRARY_PATH", ""))
            out = os.path.join(tmp, "out")
            os.makedirs(out)
            secs = []
            for _ in range(2):
                t0 = time.perf_counter()
                r = subprocess.run([tool, "-A", ats, "-d", out], capture_output=True, text=True, env=env)
                secs.append(time.perf_counter() - t0)
                if r.returncode != 0:
                    raise RuntimeError("dvda2wav_hip: rc %d: %s" % (r.returncode, r.stderr[-300:]))
            pk = re.findall(r"host peak ([0-9.]+) MB, device peak ([0-9.]+) MB", r.stdout)
            if pk:
                rec["host_peak_mb_max"] = max(float(x[0]) for x in pk)
                rec["device_peak_mb_max"] = max(float(x[1]) for x in pk)
            rec["seconds"] = round(secs[1], 3)
            rec["seconds_first_run"] = round(secs[0], 3)
            rec["value"] = round(samples / secs[1] / 1e6, 1)
            if oracle_mod is not None:
                want, got_frames, st = oracle_mod.decode(first[0], 6, first[1])
                with open(os.path.join(out, "track-01-01.wav"), "rb") as fh:
                    wav = fh.read()
                rec["first_track_identical_to_oracle"] = bool(st == 0 and wav[68:] == oracle_mod.wav_pack(want, 24))
                if not rec["first_track_identical_to_oracle"]:
                    raise SystemExit("sub-record disc_tier: track 1's WAV payload differs from the oracle's")
            rec["note"] = ("wall clock of build/dvda2wav_hip on a synthetic %d-track disc (%d MB of AOB), second run; four worker "
                           "threads on one GPU, tracks read in windows of 8 192 sectors.  The tool leaves with _exit once its "
                           "files are closed (workers parked, the HIP runtime's teardown -- 60-80 ms -- left to the operating "
                           "system; DVDA_TOOL_FULL_TEARDOWN=1 is the orderly way); the figure moves with the HIP runtime's "
                           "start, 0.17-0.29 s of it, from box to box.  device_peak_mb_max is process-wide (hipMemGetInfo): "
                           "with four workers on one GPU it includes the other workers' buffers" % (tracks, rec["aob_bytes"] >> 20))
    except SystemExit:
        raise
    except Exception as e:          # (no room for the files, no tool: the record says so, the bench line stays)
        rec["error"] = "%s: %s" % (type(e).__name__, str(e)[:200])
    return rec


def streaming_tier(pkg, oracle_mod=None):
    """Tier B, the mlp.h mirror (dvda_hip_mlpdecoder_decode_packet): one 6-ch / 96 kHz title of 1 024 access units fed
    in PES-payload sized packets (2 011 bytes), the C entry point called directly; the PCM of every call collected and
    compared with the oracle's for the whole title.  Never `value`: the contract is a call per packet."""
    import ctypes
    syn, hip = pkg.synth, pkg.hipdec
    out = {}
    for name, extra in (("recipe", {}), ("chained", dict(profile=1, features=syn.SF["CHAINED"]))):
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=1024, **extra)
        b, frames = syn.stream(cfg, 5)
        L = hip.lib()
        planar = (ctypes.POINTER(ctypes.c_int32) * 6)()
        nch = ctypes.c_uint()
        pieces = [np.ascontiguousarray(b[o:o + 2011]) for o in range(0, len(b), 2011)]
        best = None
        for rep in range(3):
            dec = hip.MLPDecoder(2, 2, 1, 1, 12)
            chans = [[] for _ in range(6)]
            got = 0
            keep = rep == 0
            t0 = time.perf_counter()
            for p in pieces:
                n = L.dvda_hip_mlpdecoder_decode_packet(dec._h, p.ctypes.data, len(p), planar, ctypes.byref(nch))
                if keep and n:
                    for c in range(6):
                        chans[c].append(np.ctypeslib.as_array(planar[c], shape=(n,)).copy())
                got += n
            dt = time.perf_counter() - t0
            path, status = dec.path, dec.status
            dec.close()
            if keep:
                pcm = np.stack([np.concatenate(c) for c in chans]) if got else np.zeros((6, 0), np.int32)
            if got != frames or (status & ~hip.ST_BENIGN):
                raise SystemExit("sub-record streaming_tier: %d of %d PCM frames, status %#x" % (got, frames, status))
            best = dt if best is None or dt < best else best
        ok = None
        if oracle_mod is not None:
            want, r, st = oracle_mod.decode(b, 6, frames)
            ok = bool(st == 0 and want.shape == pcm.shape and np.array_equal(want, pcm))
            if not ok:
                raise SystemExit("sub-record streaming_tier: PCM differs from the oracle's")
        out[name] = {"value": round(got * 6 / best / 1e6, 2), "unit": "Msamples/s", "ms_per_call": round(best / len(pieces) * 1e3, 4),
                     "calls": len(pieces), "packet_bytes": 2011, "path": "decoder state on the device" if path == 0 else "batch tier",
                     "bit_exact": ok}
    return out


def sub_records(pkg, torch, dev, local_rank, args, main_batch, steps, warmup):
    """Other shapes of the same path, each: Msamples/s over `steps` back-to-back steps, k_decode ms,
    a bit-exact sample check against the oracle.  The headline stays the main record."""
    syn, hip = pkg.synth, pkg.hipdec
    out = {}

    def in_flight(depth, flat, offs, sizes, frames, nchs, nseg, layout, chained, benign, k_steps, name, replicas=1,
                  what=None):
        """the same small batch with `depth` decode contexts in flight (own HIP stream and PCM buffer each, the
        non-blocking decode call): a batch the cooperative kernel decodes leaves most of the device idle, a feeder
        with more than one such batch to decode overlaps them.  Every slot's PCM must be the same."""
        bp = Batch(pkg, torch, dev, local_rank, flat, offs, sizes, frames, nchs, replicas, layout, 0, nseg, depth=depth,
                   chained=chained)
        dtp, _, _ = bp.timed(k_steps, 2 * depth)
        bp.check_status(benign=benign)
        same = bp.slots_equal()
        okp = bp.verify_sample(flat, offs, sizes, np.linspace(0, bp.n_streams - 1, num=min(8, bp.n_streams), dtype=np.int64))
        if not (same and okp):
            raise SystemExit("sub-record %s: the pipelined decode differs (slots equal %s, oracle %s)" % (name, same, okp))
        r = {"contexts_in_flight": depth, "value": round(bp.samples * k_steps / dtp / 1e6, 1), "unit": "Msamples/s",
             "ms_per_step": round(dtp / k_steps * 1e3, 4), "steps": k_steps, "slots_bit_identical": same,
             "what": what or ("%d decode contexts in flight on their own HIP streams, non-blocking decode call; a step is "
                              "still one index + one decode of the whole batch" % depth)}
        bp.close()
        del bp
        torch.cuda.empty_cache()
        return r

    only = set(x for x in args.only_sub.split(",") if x)

    def run(name, flat, offs, sizes, frames, nchs, nseg, replicas, layout, lanes, benign=0, note=None, flight=0,
            chained=False):
        if only and name not in only:
            return
        t_run = time.perf_counter()
        b = Batch(pkg, torch, dev, local_rank, flat, offs, sizes, frames, nchs, replicas, layout, lanes, nseg)
        dt, kms, launches = b.timed(steps, warmup)
        b.check_status(benign=benign)
        picks = np.linspace(0, b.n_streams - 1, num=min(8, b.n_streams), dtype=np.int64)
        ok = b.verify_sample(flat, offs, sizes, picks)
        if not ok:
            raise SystemExit("sub-record %s: HIP decode differs from the oracle" % name)
        # kernel_ms: device time of the whole decode call (fast pass + chain passes + sequential pass + what lies
        # between them); fast_pass_ms: the fast-pass kernels alone
        algo = b.comp_bytes + b.out_bytes * b.samples
        kall = b.last_decode_ms
        rec = {"value": round(b.samples * steps / dt / 1e6, 1), "unit": "Msamples/s",
               "ms_per_step": round(dt / steps * 1e3, 3), "kernel_ms": round(kall, 4),
               "fast_pass_ms": round(kms, 4), "titles": b.n_streams,
               "samples_per_step": b.samples, "compressed_bytes": b.comp_bytes,
               "algorithmic_bytes_per_launch": algo,
               # the decode call's kernels against the HBM roofline: algorithmic bytes / device time of the whole
               # call (HIP events, first kernel to last); `traffic` = counter bytes of the same sub-record from the
               # committed PMC passes (profiles/sub_traffic.json), null when none was taken at this size
               "roofline": {"bound": "hbm", "kernels": "decode call (all passes)",
                            "achieved": round(algo / (kall * 1e-3) / 1e9, 2) if kall > 0 else 0.0,
                            "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(algo / (kall * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if kall > 0 else 0.0,
                            "traffic": sub_traffic(name, b.samples, b.comp_bytes)},
               "bit_exact_sample": ok}
        if note:
            rec["note"] = note
        out[name] = rec
        b.close()
        del b
        torch.cuda.empty_cache()
        if flight > 1:
            rec["in_flight"] = in_flight(flight, flat, offs, sizes, frames, nchs, nseg, layout, chained, benign,
                                         max(steps, 30), name, replicas=replicas)
        sys.stderr.write("bench: sub-record %s %.1f s\n" % (name, time.perf_counter() - t_run))

    rate = 1
    # ---- planar layout: the order the mlp.h contract appends to `samples`
    cfg = syn.make_cfg(assignment=12, rate_code=rate, n_substreams=1, n_aus=args.aus)
    flat, offs, sizes, frames = syn.batch(cfg, 1, args.streams)
    nseg = args.streams * args.replicas * ((args.aus + cfg.restart_interval - 1) // cfg.restart_interval)
    run("planar_layout", flat, offs, sizes, frames, np.full(len(sizes), 6), nseg, args.replicas, "planar", 0,
        note="same titles as the headline, PCM written planar [channel][frame] (reference decode_packet order)")
    # ---- the headline batch with TWO steps in flight (what a feeder with a second batch to decode does; `--pipeline 2` makes
    #      the main record run this way): the host's part of a step and most of the next batch's index hide behind this
    #      batch's decode.  Not the main record: k_decode shares the device with the other step's index kernels and its own
    #      time -- what the roofline figure is quoted on -- rises with it
    if not only or "two_steps_in_flight" in only:
        out["two_steps_in_flight"] = in_flight(
            2, flat, offs, sizes, frames, np.full(len(sizes), 6), nseg, args.layout, False, 0, max(steps, 30),
            "two_steps_in_flight", replicas=args.replicas,
            what="the headline batch, two decode contexts in flight (own HIP stream and PCM buffer each, non-blocking decode "
                 "call, the second step's decode ordered behind the first's by an event): a step is still one index + one "
                 "decode of the whole batch; whole-job throughput of a feeder that always has the next batch ready")
    # ---- the output stage fused into the decode: interleaved packed 24-bit WAV payload, 3 B per sample out
    run("wav24_output", flat, offs, sizes, frames, np.full(len(sizes), 6), nseg, args.replicas, "wav24", 0,
        note="same titles, d_pcm receives the little-endian 24-bit WAV payload dvda2wav writes (write_signed): "
             "algorithmic bytes = compressed in + 3 B per sample out")
    # ---- batch-size sweep on the headline layout (non-multiples of the 2 048 resident waves included)
    sweep = []
    for n in (() if only and "batch_sweep" not in only else (64, 512, 1000, 1536, 2500, 4096)):
        n = min(n, main_batch.n_streams)
        dt, kms, launches = main_batch.timed(max(5, steps // 4), 2, n_streams=n)
        smp = int((main_batch.all_frames[:n] * main_batch.all_nch[:n]).sum())
        k = max(5, steps // 4)
        sweep.append({"titles": n, "segments": n * ((args.aus + 7) // 8), "value": round(smp * k / dt / 1e6, 1),
                      "ms_per_step": round(dt / k * 1e3, 3), "kernel_ms": round(kms, 4)})
    if sweep:
        out["batch_sweep"] = {"unit": "Msamples/s", "points": sweep}
    # ---- two substreams (ch 0-1 | ch 2-5, matrices in substream 1)
    cfg2 = syn.make_cfg(assignment=12, rate_code=rate, n_substreams=2, n_aus=args.aus)
    flat, offs, sizes, frames = syn.batch(cfg2, 1, args.streams)
    run("two_substreams", flat, offs, sizes, frames, np.full(len(sizes), 6), nseg, args.replicas, args.layout, 0)
    # ---- configs[1] shape: 2-ch / 96 kHz / 24-bit titles
    cfg1 = syn.make_cfg(assignment=1, rate_code=rate, n_substreams=1, n_aus=args.aus)
    flat, offs, sizes, frames = syn.batch(cfg1, 1, args.streams)
    run("stereo_c2", flat, offs, sizes, frames, np.full(len(sizes), 2), nseg, args.replicas, "planar", 0,
        note="BASELINE configs[1] shape (2ch/96k/24b), planar layout")
    # ---- heterogeneous batches: header parses diverge inside the waves (the headline batch is ONE shape:
    #      every lane of a wave reaches every block and frame header in the same loop turn).
    #      "heterogeneous": regular titles as an encoder writes them, but eight different kinds in one batch
    #      (6/2/5/1-ch, 48/96/192 kHz, 1-5 blocks per access unit, restart every 4-16 units, FIR order 4-8, all
    #      three code books, 0-2 matrices, different lengths);
    #      "fuzz_fast_features": the test generator's fuzz profile -- parameters change on a third of all
    #      blocks, random block splits, per-channel code books ... (no IIR, the recipe's 2 matrices);
    #      "fuzz_all_features" adds what DVD-Audio discs rarely carry: IIR taps with transmitted state (cold
    #      workspace path) and up to 6 random matrices
    kinds = [dict(assignment=12, rate_code=1, restart_interval=8, blocks_per_au=2, fir_order=8, codebook=1,
                  huffman_lsbs=12, n_matrices=2, n_aus=64),
             dict(assignment=1, rate_code=1, restart_interval=16, blocks_per_au=1, fir_order=4, codebook=2,
                  huffman_lsbs=10, n_matrices=1, n_aus=72),
             dict(assignment=12, rate_code=2, restart_interval=8, blocks_per_au=4, fir_order=8, codebook=3,
                  huffman_lsbs=14, n_matrices=2, n_aus=40),
             dict(assignment=0x12, rate_code=0, restart_interval=8, blocks_per_au=1, fir_order=6, codebook=1,
                  huffman_lsbs=8, n_matrices=2, n_aus=96),
             dict(assignment=12, rate_code=0, restart_interval=12, blocks_per_au=2, fir_order=8, codebook=2,
                  huffman_lsbs=12, n_matrices=0, n_aus=80),
             dict(assignment=6, rate_code=1, restart_interval=8, blocks_per_au=2, fir_order=5, codebook=1,
                  huffman_lsbs=11, n_matrices=2, n_aus=64),
             dict(assignment=0, rate_code=2, restart_interval=4, blocks_per_au=2, fir_order=8, codebook=1,
                  huffman_lsbs=16, n_matrices=0, n_aus=48),
             dict(assignment=12, rate_code=1, restart_interval=10, blocks_per_au=5, fir_order=8, codebook=1,
                  huffman_lsbs=13, n_matrices=2, n_aus=56)]
    # (a) long titles, as on a disc: a title's segments are neighbouring lanes, so most waves hold one kind
    specs = [(syn.make_cfg(n_substreams=1, **dict(k, n_aus=512)), 256) for k in kinds]
    flat, offs, sizes, frames, nchs, nseg_h = gen_mixed(syn, specs, 60000)
    order = np.arange(len(sizes)).reshape(len(kinds), -1).T.ravel()
    flat2, offs2, lens2 = hip.pack_streams([flat[int(offs[i]):int(offs[i] + sizes[i])] for i in order])
    run("heterogeneous", flat2, offs2.astype(np.int64), lens2.astype(np.int64), frames[order], nchs[order], nseg_h * 2, 2,
        "planar", 0, note="8 kinds of regular titles interleaved title by title (6/2/5/1-ch, 48/96/192 kHz, 1-5 blocks "
                          "per access unit, restart every 4-16 units, FIR 4-8 taps, code books 1-3, 0-2 matrices), "
                          "4 096 titles of 512 access units: 32-128 segments per title, so a wave holds one or two kinds")
    # (b) the same kinds as short titles (5-12 segments each): every wave holds ~8 different kinds and every
    #     block / frame / restart header of every lane is a divergent parse for the whole wave
    specs = [(syn.make_cfg(n_substreams=1, **k), 512) for k in kinds]
    flat, offs, sizes, frames, nchs, nseg_h = gen_mixed(syn, specs, 70000)
    # interleave the kinds title by title: neighbouring lanes of a wave then hold different kinds
    order = np.arange(len(sizes)).reshape(len(kinds), -1).T.ravel()
    flat2, offs2, lens2 = hip.pack_streams([flat[int(offs[i]):int(offs[i] + sizes[i])] for i in order])
    run("heterogeneous_short_titles", flat2, offs2.astype(np.int64), lens2.astype(np.int64), frames[order], nchs[order],
        nseg_h * 4, 4, "planar", 0,
        note="the same 8 kinds as 16 384 titles of 40-96 access units (5-12 segments each), interleaved title by title")
    shapes = [(12, 1, 8), (1, 1, 5), (12, 2, 16), (0x12, 0, 3), (12, 0, 8), (6, 1, 4), (0, 2, 8), (12, 1, 2)]
    SF = syn.SF
    common = syn.SF_FAST & ~(SF["IIR"] | SF["MATRIXRAND"])
    for name, feats, note in (
            ("fuzz_fast_features", common, "no IIR, the recipe's 2 matrices"),
            ("fuzz_all_features", syn.SF_FAST, "IIR taps and up to 6 random matrices on top")):
        specs = [(syn.make_cfg(assignment=asg, rate_code=rc, n_substreams=1, n_aus=64, profile=1, features=feats,
                               restart_interval=ri), 512) for asg, rc, ri in shapes]
        flat, offs, sizes, frames, nchs, nseg_h = gen_mixed(syn, specs, 90000)
        run(name, flat, offs, sizes, frames, nchs, nseg_h * 4, 4, "planar", 0, benign=hip.ST_BENIGN,
            note="8 fuzz-profile configurations (6/2/5/1-ch, 48/96/192 kHz, restart every 2..16 AUs; %s), "
                 "16 384 titles of 64 access units" % note)
    # ---- chained titles: no raw lead-in after a title's first segment, the FIR history runs through
    #      the whole title (src/mlp.c never clears it)
    cfgc = syn.make_cfg(assignment=12, rate_code=rate, n_substreams=1, n_aus=128, profile=1,
                        features=syn.SF["CHAINED"])
    flat, offs, sizes, frames = syn.batch(cfgc, 1, 1024)
    run("chained_titles", flat, offs, sizes, frames, np.full(len(sizes), 6), 1024 * (128 // 8 + 2), 1, args.layout, 0,
        benign=hip.ST_BENIGN, note="1 024 titles x 128 access units, every segment depends on the one before")
    # ---- what a 6-channel disc carries: chained titles with two substreams (ch 0-1 | ch 2-5), at the bench batch's size
    cfgd = syn.make_cfg(assignment=12, rate_code=rate, n_substreams=2, n_aus=args.aus, profile=1,
                        features=syn.SF["CHAINED"])
    flat, offs, sizes, frames = syn.batch(cfgd, 1, args.streams)
    run("chained_two_substreams", flat, offs, sizes, frames, np.full(len(sizes), 6), nseg + args.streams * args.replicas * 2,
        args.replicas, args.layout, 0, benign=hip.ST_BENIGN, flight=args.chain_flight, chained=True,
        note="the bench batch as a disc would hold it: every segment continues the FIR history of the one before it, "
             "two substreams per title; decoded by the chain passes (parse in lane pairs, filter, rematrix)")
    # ---- the shape a disc really has: chained, two substreams, EVERY block carries parameters (most channels re-send
    #      their FIR taps), mixed code books, blocks at fixed positions
    SFd = syn.SF
    cfgp = syn.make_cfg(assignment=12, rate_code=rate, n_substreams=2, n_aus=args.aus, profile=1,
                        features=SFd["DISC"] | SFd["CHAINED"] | SFd["FIRRAND"] | SFd["MIXBOOKS"])
    flat, offs, sizes, frames = syn.batch(cfgp, 1, args.streams)
    run("disc_profile", flat, offs, sizes, frames, np.full(len(sizes), 6), nseg + args.streams * args.replicas * 2,
        args.replicas, args.layout, 0, benign=hip.ST_BENIGN,
        note="what an encoder writes: chained titles, two substreams (ch 0-1 | ch 2-5), parameters on every block -- most "
             "channels re-send their FIR taps --, per-channel code books, two blocks of 40 PCM frames per access unit")
    one = syn.make_cfg(assignment=12, rate_code=rate, n_substreams=1, n_aus=512, profile=1, features=syn.SF["CHAINED"])
    flat, offs, sizes, frames = syn.batch(one, 7, 1)
    run("chained_single_title", flat, offs, sizes, frames, np.full(1, 6), 512 // 8 + 2, 1, args.layout, 0,
        benign=hip.ST_BENIGN, note="ONE chained 6-ch title of 512 access units (the low-parallelism case)", flight=4,
        chained=True)
    long_one = syn.make_cfg(assignment=12, rate_code=rate, n_substreams=1, n_aus=8192, profile=1,
                            features=syn.SF["CHAINED"])
    flat, offs, sizes, frames = syn.batch(long_one, 9, 1)
    run("chained_single_long_title", flat, offs, sizes, frames, np.full(1, 6), 8192 // 8 + 2, 1, args.layout, 0,
        benign=hip.ST_BENIGN, note="ONE chained 6-ch title of 8 192 access units (68 s of 96 kHz audio): the filter "
                                   "pass's serial recurrence is what bounds it", flight=4, chained=True)
    if only and "c4" not in only and "mixed_corpus_c5" not in only:
        return out
    if not only or "mixed_corpus_c5" in only:
        out["mixed_corpus_c5"] = mixed_corpus(pkg, torch, dev, local_rank, args, steps, warmup)
    if only and "c4" not in only:
        return out
    # ---- BASELINE configs[3] on this GPU: 1 024 independent single-access-unit streams in ONE batch -- the low-
    #      parallelism regime, decoded by the wave-cooperative kernel (csrc/mlp_coop.h); throughput back to back and
    #      the latency of one batch (index + decode enqueued and waited for, host clock); all 1 024 vs the oracle
    t_run = time.perf_counter()
    cfg4 = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=1)
    flat4, offs4, sizes4, frames4 = syn.batch(cfg4, 1, 1024)
    b4 = Batch(pkg, torch, dev, local_rank, flat4, offs4, sizes4, frames4, np.full(1024, 6), 1, "interleaved", 0, 1024)
    k4 = 200
    dt, kms, _ = b4.timed(k4, 10)
    b4.check_status()
    lat = []
    for _ in range(100):
        b4.sync()
        t1 = time.perf_counter()
        b4.step()
        b4.sync()
        lat.append(time.perf_counter() - t1)
    lat.sort()
    ok4 = b4.verify_sample(flat4, offs4, sizes4, np.arange(1024))
    if not ok4:
        raise SystemExit("sub-record c4: HIP decode differs from the oracle")
    out["c4"] = {"value": round(b4.samples * k4 / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_step": round(dt / k4 * 1e3, 4),
                 "kernel_ms": round(b4.last_decode_ms, 4), "fast_pass_ms": round(kms, 4), "streams": 1024,
                 "samples_per_step": b4.samples, "bit_exact_all_1024": ok4,
                 "latency": {"median_ms": round(lat[len(lat) // 2] * 1e3, 4), "min_ms": round(lat[0] * 1e3, 4),
                             "p90_ms": round(lat[int(len(lat) * 0.9)] * 1e3, 4), "batches": len(lat),
                             "what": "one batch: index + decode enqueued and waited for, host clock"},
                 "note": "BASELINE configs[3]: 1 024 independent 6-ch/96 kHz streams of ONE access unit (major sync + "
                         "restart header + raw lead-in), one batch, one GPU; kernel picked by the library (wave-cooperative)"}
    b4.close()
    out["c4"]["in_flight"] = in_flight(3, flat4, offs4, sizes4, frames4, np.full(1024, 6), 1024, "interleaved", False, 0,
                                       400, "c4")
    sys.stderr.write("bench: sub-record c4 %.1f s\n" % (time.perf_counter() - t_run))
    return out


def mixed_corpus(pkg, torch, dev, local_rank, args, steps, warmup):
    """BASELINE configs[4] on one GPU: 6-ch / 192 kHz / 24-bit MLP titles and 6-ch / 24-bit raw-PCM AOB titles,
    resident in HBM, decoded at the same time -- the MLP batch on one HIP stream, the PCM sectors
    (sector walk + AOB byte un-swizzle) on another, each driven by its own host thread.  The
    record carries the two kinds alone and together, in samples and in algorithmic bytes per second."""
    import ctypes
    import threading
    syn, hip, disc = pkg.synth, pkg.hipdec, pkg.disc
    t_run = time.perf_counter()
    cfg = syn.make_cfg(assignment=12, rate_code=2, n_substreams=1, n_aus=256)        # 160 PCM frames per access unit
    flat, offs, sizes, frames = syn.batch(cfg, 4001, args.streams)
    nseg = args.streams * args.replicas * ((256 + cfg.restart_interval - 1) // cfg.restart_interval)
    mlp = Batch(pkg, torch, dev, local_rank, flat, offs, sizes, frames, np.full(len(sizes), 6), args.replicas,
                "interleaved", 0, nseg)
    # raw-PCM titles: 2 048 unique sectors of random 24-bit frames, repeated on the device
    rng = np.random.RandomState(5)
    unit_frames = 110 * 2048
    src = rng.randint(-(1 << 23), 1 << 23, size=(unit_frames, 6))
    unit = np.frombuffer(b"".join(disc.pcm_track_sectors(src, 2, 1, 12)), np.uint8)
    reps = 128
    d_sec = torch.from_numpy(unit.copy()).to(dev).repeat(reps)
    n_sec = d_sec.numel() // 2048
    cap = n_sec * 110 + 2
    d_pcm = torch.empty(6 * cap, dtype=torch.int32, device=dev)
    lib = hip.lib()
    d_work = torch.zeros(int(lib.dvda_pcm_hip_workspace_words(n_sec)), dtype=torch.int32, device=dev)
    pstream = torch.cuda.Stream(dev)
    pcm_samples = n_sec * 110 * 6
    pcm_bytes = n_sec * 2048 + pcm_samples * 4

    def pcm_step():
        hip._check(lib.dvda_pcm_hip_decode_sectors(d_sec.data_ptr(), n_sec, 24, 6, d_pcm.data_ptr(), cap,
                                                   d_work.data_ptr(), pstream.cuda_stream), "pcm")

    def pcm_timed(k):
        for _ in range(2):
            pcm_step()
        pstream.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            pcm_step()
        pstream.synchronize()
        return time.perf_counter() - t0

    # each kind alone
    dt_m, kms, _ = mlp.timed(steps, warmup)
    k_pcm = steps * 4
    dt_p = pcm_timed(k_pcm)
    # both at once: the PCM thread keeps going until the MLP thread is through
    done = threading.Event()
    count = [0]
    t_p = [0.0]

    def pcm_loop():
        torch.cuda.set_device(dev)
        t0 = time.perf_counter()
        while not done.is_set():
            pcm_step()
            count[0] += 1
            if count[0] % 8 == 0:
                pstream.synchronize()
        pstream.synchronize()
        t_p[0] = time.perf_counter() - t0

    th = threading.Thread(target=pcm_loop)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    th.start()
    dt_both, kms_both, _ = mlp.timed(steps, 1)
    done.set()
    th.join()
    wall = time.perf_counter() - t0
    mlp.check_status()
    picks = np.linspace(0, mlp.n_streams - 1, num=4, dtype=np.int64)
    ok = mlp.verify_sample(flat, offs, sizes, picks)
    fr, bad = ctypes.c_uint64(), ctypes.c_uint32()
    lib.dvda_pcm_hip_result(d_work.data_ptr(), n_sec, ctypes.byref(fr), ctypes.byref(bad), pstream.cuda_stream)
    ok = ok and bad.value == 0 and fr.value == n_sec * 110 and \
        bool(np.array_equal(d_pcm.view(6, cap)[:, :unit_frames].cpu().numpy(), src.T))
    if not ok:
        raise SystemExit("sub-record mixed_corpus_c5: results differ from the oracles")
    mlp_bytes = mlp.comp_bytes + 4 * mlp.samples
    both_samples = mlp.samples * steps + pcm_samples * count[0]
    both_bytes = mlp_bytes * steps + pcm_bytes * count[0]
    rec = {"unit": "Msamples/s",
           "mlp_192k_alone": {"value": round(mlp.samples * steps / dt_m / 1e6, 1), "kernel_ms": round(kms, 4),
                              "GBs": round(mlp_bytes * steps / dt_m / 1e9, 1), "titles": mlp.n_streams},
           "raw_pcm_alone": {"value": round(pcm_samples * k_pcm / dt_p / 1e6, 1),
                             "GBs": round(pcm_bytes * k_pcm / dt_p / 1e9, 1), "sectors": n_sec},
           "together": {"value": round(both_samples / wall / 1e6, 1), "GBs": round(both_bytes / wall / 1e9, 1),
                        "frac_of_hbm_peak": round(both_bytes / wall / 1e9 / HBM_PEAK_GBS, 4),
                        "mlp_value": round(mlp.samples * steps / dt_both / 1e6, 1), "mlp_kernel_ms": round(kms_both, 4),
                        "pcm_value": round(pcm_samples * count[0] / t_p[0] / 1e6, 1), "pcm_passes": count[0],
                        "seconds": round(wall, 4)},
           "bit_exact_sample": ok,
           "note": "BASELINE configs[4] on one GPU: 6-ch/192 kHz MLP titles (256 access units each) on one HIP stream, "
                   "6-ch/24-bit raw-PCM AOB sectors on another, one host thread each; bytes are algorithmic "
                   "(compressed or sector bytes in + int32 PCM out)"}
    sys.stderr.write("bench: sub-record mixed_corpus_c5 %.1f s\n" % (time.perf_counter() - t_run))
    mlp.close()
    del mlp, d_sec, d_pcm, d_work
    torch.cuda.empty_cache()
    return rec


def host_to_host(pkg, torch, dev, local_rank, args, flat, offs, sizes, frames, steps, wav24=False):
    """Pinned host bytes -> H2D -> index + decode -> D2H -> pinned host PCM ('compressed bytes resident in
    pinned host memory -> PCM resident in host memory').  Never `value`.

    Round 6: the two directions of the link on streams of their own -- every H2D copy on one stream, every D2H copy on
    another -- and the index + decode of the sub-batches alternating between TWO compute streams (non-blocking decode
    call: one host thread enqueues the whole pass; four streams = the runtime's four hardware queues), tied together
    by events:  H2D(i) waits for the decode that last read its input buffer;  decode(i) waits for H2D(i) and for the
    D2H that last read its PCM buffer;  D2H(i) waits for decode(i).  A sub-batch of 64 titles takes the lane kernel
    one segment's time (~1.6 ms, a lone wave per SIMD) however few segments it holds, its copy back 0.9-1.2 ms: two
    decodes in flight (with GPU_MAX_HW_QUEUES=8 and three: 18.3 instead of 20.7 ms per pass on a box whose link gave 17.0
    -- not the default: see where this is called); the first two sub-batches are small (16 and 48 titles: the cooperative kernel's
    size) so that the first copy back starts after 0.6 ms instead of 2.2.
    (Rounds 3-5: 8 sub-batches on 3 streams, copy-in / decode / copy-out one after the other on each stream, a host
    thread per stream: the two directions mostly took turns.)
    `pcie_ceiling_GBs`: the same copies with nothing decoded and no event between them -- what this box's link gives with
    both directions busy."""
    hip = pkg.hipdec
    n = len(sizes)
    if n >= 256:
        cuts = [0, 16, 64] + list(range(128, n, 64)) + [n]
    else:
        cuts = [0, n]
    parts = len(cuts) - 1
    nch = 6
    layout = hip.PCM_WAV24 if wav24 else (hip.PCM_INTERLEAVED if args.layout == "interleaved" else hip.PCM_PLANAR)
    out_bytes = 3 if wav24 else 4
    slots = []
    for p in range(parts):
        lo, hi = cuts[p], cuts[p + 1]
        b0 = int(offs[lo])
        b1 = int(offs[hi]) if hi < n else int(len(flat) - 64)
        h_in = torch.from_numpy(flat[b0:b1 + 64].copy()).pin_memory()
        rows = frames[lo:hi].astype(np.int64)
        words = (rows * nch * out_bytes + 15) // 16 * 4            # int32 units per title, 16-byte aligned
        oo = np.zeros(hi - lo, np.int64)
        oo[1:] = np.cumsum(words[:-1])
        tot = int(words.sum())
        slots.append(dict(lo=lo, h_in=h_in, nbytes=b1 - b0, off=(offs[lo:hi].astype(np.int64) - b0), len=sizes[lo:hi].astype(np.int64),
                          rows=rows, oo=oo, tot=tot, h_out=torch.empty(tot, dtype=torch.int32).pin_memory()))
    NB = min(int(os.environ.get("DVDA_BENCH_H2H_BUFS", "4")), parts)
    bufs = []
    maxb = max(s["nbytes"] for s in slots) + 64
    maxt = max(s["tot"] for s in slots)
    maxn = max(len(s["len"]) for s in slots)
    nseg = maxn * ((args.aus + 7) // 8)
    n_cs = int(os.environ.get("DVDA_BENCH_H2H_STREAMS", "2"))
    for _ in range(NB):
        bufs.append(dict(d_in=torch.zeros(maxb, dtype=torch.uint8, device=dev),
                         d_pcm=torch.empty(maxt, dtype=torch.int32, device=dev),
                         ctx=hip.Context(local_rank, maxn, nseg, lanes_per_segment=0, layout=layout)))
    for s in slots:     # the small per-stream tables live on the device (they are part of the request, not payload)
        s["d_off"] = torch.from_numpy(s["off"]).to(dev)
        s["d_len"] = torch.from_numpy(s["len"]).to(dev)
        s["d_oo"] = torch.from_numpy(s["oo"]).to(dev)
        s["d_rows"] = torch.from_numpy(s["rows"]).to(dev)
    torch.cuda.synchronize(dev)
    # The pass's five streams and its events, made HERE with the runtime's own calls, one after the other: the runtime
    # deals hardware queues to streams in the order they are made, and the streams of torch's pool were made long ago,
    # between the streams of every context the sub-records before this one opened and closed -- two of the five then
    # shared a queue and the decodes took turns (26 ms per pass behind the other sub-records, 18 ms in a run of its own).
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    vp, u32, sz = ctypes.c_void_p, ctypes.c_uint, ctypes.c_size_t
    rt.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(vp), u32]
    rt.hipStreamDestroy.argtypes = [vp]
    rt.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(vp), u32]
    rt.hipEventDestroy.argtypes = [vp]
    rt.hipEventRecord.argtypes = [vp, vp]
    rt.hipStreamWaitEvent.argtypes = [vp, vp, u32]
    rt.hipMemcpyAsync.argtypes = [vp, vp, sz, ctypes.c_int, vp]
    rt.hipDeviceSynchronize.argtypes = []

    def chk(rc):
        if rc != 0:
            raise SystemExit("host_to_host: HIP runtime call failed (%d)" % rc)

    def mk_stream():
        h = vp()
        chk(rt.hipStreamCreateWithFlags(ctypes.byref(h), 1))        # hipStreamNonBlocking
        return h

    def mk_event():
        h = vp()
        chk(rt.hipEventCreateWithFlags(ctypes.byref(h), 2))         # hipEventDisableTiming
        return h

    s_in, s_out = mk_stream(), mk_stream()
    s_c = [mk_stream() for _ in range(n_cs)]
    for bf in bufs:
        bf["ev_in"], bf["ev_c"], bf["ev_out"] = mk_event(), mk_event(), mk_event()
    H2D, D2H = 1, 2

    def one_pass():
        for i, s in enumerate(slots):
            bf = bufs[i % NB]
            cs = s_c[i % len(s_c)]
            first = i < NB
            if not first:
                chk(rt.hipStreamWaitEvent(s_in, bf["ev_c"], 0))     # the decode that read this input buffer is through
            chk(rt.hipMemcpyAsync(bf["d_in"].data_ptr(), s["h_in"].data_ptr(), s["nbytes"] + 64, H2D, s_in))
            chk(rt.hipEventRecord(bf["ev_in"], s_in))
            chk(rt.hipStreamWaitEvent(cs, bf["ev_in"], 0))
            if not first:
                chk(rt.hipStreamWaitEvent(cs, bf["ev_out"], 0))     # the copy that read this PCM buffer is through
            bf["ctx"].index(bf["d_in"].data_ptr(), s["nbytes"], s["d_off"].data_ptr(), s["d_len"].data_ptr(),
                            len(s["len"]), cs.value)
            bf["ctx"].decode_async(bf["d_pcm"].data_ptr(), s["d_oo"].data_ptr(), s["d_rows"].data_ptr(), cs.value)
            chk(rt.hipEventRecord(bf["ev_c"], cs))
            chk(rt.hipStreamWaitEvent(s_out, bf["ev_c"], 0))
            chk(rt.hipMemcpyAsync(s["h_out"].data_ptr(), bf["d_pcm"].data_ptr(), s["tot"] * 4, D2H, s_out))
            chk(rt.hipEventRecord(bf["ev_out"], s_out))
        t_enq = time.perf_counter()
        chk(rt.hipDeviceSynchronize())
        return t_enq

    def copies_only():
        for i, s in enumerate(slots):
            bf = bufs[i % NB]
            chk(rt.hipMemcpyAsync(bf["d_in"].data_ptr(), s["h_in"].data_ptr(), s["nbytes"] + 64, H2D, s_in))
            chk(rt.hipMemcpyAsync(s["h_out"].data_ptr(), bf["d_pcm"].data_ptr(), s["tot"] * 4, D2H, s_out))
        chk(rt.hipDeviceSynchronize())

    # ---- what the link gives with both directions busy: the same copies, nothing decoded, nothing waited for
    copies_only()
    t0 = time.perf_counter()
    for _ in range(steps):
        copies_only()
    dt_copy = time.perf_counter() - t0
    # ---- the path itself
    one_pass()
    one_pass()
    t0 = time.perf_counter()
    enq = 0.0
    for _ in range(steps):
        t1 = time.perf_counter()
        enq += one_pass() - t1
    dt = time.perf_counter() - t0
    samples = sum(int((s["rows"] * nch).sum()) for s in slots)
    d2h = sum(s["tot"] for s in slots) * 4
    nbytes = sum(s["nbytes"] for s in slots)
    for i, bf in enumerate(bufs):
        infos = bf["ctx"].stream_info(stream=s_c[0].value)
        if any(inf.status for inf in infos):
            raise SystemExit("host_to_host: decode reported errors")
        bf["ctx"].close()
    chk(rt.hipDeviceSynchronize())
    for bf in bufs:
        for k in ("ev_in", "ev_c", "ev_out"):
            rt.hipEventDestroy(bf[k])
    for h in [s_in, s_out] + s_c:
        rt.hipStreamDestroy(h)
    # the host copy of the first, a middle and the last sub-batch against the oracle, one title each
    from tests import oracle_lib
    ora = oracle_lib.Oracle()
    ok = True
    for p, k in ((0, 0), (parts // 2, 1), (parts - 1, len(slots[parts - 1]["len"]) - 1)):
        s = slots[p]
        i = s["lo"] + k
        want, r, st = ora.decode(flat[int(offs[i]):int(offs[i] + sizes[i])], nch, int(frames[i]))
        if wav24:
            nb = int(s["rows"][k]) * nch * 3
            a = s["h_out"].numpy()[int(s["oo"][k]):int(s["oo"][k]) + (nb + 3) // 4].view(np.uint8)[:nb]
            ok = ok and st == 0 and a.tobytes() == ora.wav_pack(want, 24)
            continue
        a = s["h_out"].numpy()[int(s["oo"][k]):int(s["oo"][k]) + int(s["rows"][k]) * nch]
        got = a.reshape(-1, nch).T if layout == hip.PCM_INTERLEAVED else a.reshape(nch, -1)
        ok = ok and st == 0 and np.array_equal(got, want)
    if not ok:
        raise SystemExit("host_to_host: PCM in host memory differs from the oracle")
    return {"value": round(samples * steps / dt / 1e6, 1), "unit": "Msamples/s", "ms_per_pass": round(dt / steps * 1e3, 3),
            "titles": n, "sub_batches": parts, "h2d_bytes": nbytes, "d2h_bytes": d2h,
            "pcie_GBs": round((nbytes + d2h) * steps / dt / 1e9, 2),
            "pcie_ceiling_GBs": round((nbytes + d2h) * steps / dt_copy / 1e9, 2),
            "copy_only_ms_per_pass": round(dt_copy / steps * 1e3, 3), "host_enqueue_ms_per_pass": round(enq / steps * 1e3, 3),
            "bit_exact_sample": ok,
            "note": "pinned host -> H2D -> index+decode -> D2H -> pinned host, %d sub-batches (16, 48, then 64 titles); H2D and "
                    "D2H on streams of their own, two compute streams, event edges; pcie_ceiling_GBs = the same copies with "
                    "nothing decoded and nothing waited for%s" % (
                parts, "; the decode writes the packed 24-bit WAV payload: 3 B per sample go back" if wav24 else "")}


# ----------------------------------------------------------------------------- plumbing-only ranks (CPU tests)
def plumbing_rank(args, rank, world):
    """DVDA_BENCH_PLUMBING=1: what tests/test_dist_gloo.py drives on a box without GPUs -- the launcher,
    the rank environment, the shard and the summary all-reduce (gloo).  Nothing is decoded and nothing is
    measured: value is null."""
    import torch
    import torch.distributed as dist
    import libdvd_audio_amd as pkg
    syn = pkg.synth
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.workload == "c4":
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=1)
        flat, offs, sizes, frames = syn.batch(cfg, 1, args.streams)
        mine = pkg.shard.shard_titles(sizes, world, rank)
    else:
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=args.aus)
        flat, offs, sizes, frames = syn.batch(cfg, 1 + rank * args.streams, args.streams)
        mine = np.arange(len(sizes))
    rows = int(frames[mine].sum())
    summ = pkg.shard.reduce_summary(dist if world > 1 else None, torch.device("cpu"), rows, rows * 6,
                                    int(sizes[mine].sum()), 0, len(mine), 0.001 * (rank + 1), checked=False,
                                    per_rank=(0.0, 0.0, 1.0 * (rank + 1)))
    if rank == 0:
        print(json.dumps({"metric": METRIC, "value": None, "unit": "Msamples/s", "n_gpus": world, "plumbing_only": True,
                          "scaling": "strong" if args.workload == "c4" else "weak",
                          "config": {"workload": args.workload, "titles_all_ranks": summ["checksum"],
                                     "samples_all_ranks": summ["samples"],
                                     "compressed_bytes_all_ranks": summ["compressed_bytes"]},
                          "seconds_max_over_ranks": summ["seconds"],
                          "ranks": {"seconds_min": summ["seconds_min"], "compressed_bytes_max": summ["bytes_max"],
                                    "compressed_bytes_min": summ["bytes_min"],
                                    "load_imbalance": round(summ["bytes_max"] * world / max(summ["compressed_bytes"], 1), 4),
                                    "bit_exact_on_every_rank": summ["all_verified"],
                                    "world_size": summ["world_size"], "ranks_counted": summ["ranks_counted"],
                                    "per_rank": [{"rank": r, "k_decode_ms": v[0], "roofline_frac": v[1], "ms_per_step": v[2]}
                                                 for r, v in enumerate(summ["per_rank"])]}}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


# ----------------------------------------------------------------------------- main
def main():
    args = parse_args()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(env_world or "1")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("DVDA_BENCH_PLUMBING") == "1":
        return plumbing_rank(args, rank, world)

    import torch
    import torch.distributed as dist
    # Plumbing check on a 1-GPU box only (never what the driver measures): DVDA_BENCH_ONE_GPU=1 puts
    # every rank on cuda:0 and DVDA_BENCH_BACKEND=gloo exchanges the summary on the host, because RCCL
    # refuses two ranks on one device.
    backend = os.environ.get("DVDA_BENCH_BACKEND", "nccl")
    if os.environ.get("DVDA_BENCH_ONE_GPU") == "1":
        local_rank = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the decode path is HIP-only)")
    # the rank's GPU first, then the process group bound to it: RCCL otherwise picks the device at the first
    # collective (every rank on GPU 0)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import libdvd_audio_amd as pkg
    syn, hip = pkg.synth, pkg.hipdec
    assignment, rate_code = args.assignment, 1
    nch = syn.channels(assignment)

    # ---- synthetic titles, generated on the host cores
    t_gen = time.perf_counter()
    if args.workload == "c4":
        # 1 024 single-access-unit streams (major sync + restart header + raw lead-in block), different
        # seeds, dealt to the ranks by compressed size: total work fixed as N grows
        cfg = syn.make_cfg(assignment=assignment, rate_code=rate_code, n_substreams=args.substreams, n_aus=1)
        flat_all, offs_all, sizes_all, frames_all = syn.batch(cfg, 1, args.streams)
        mine = pkg.shard.shard_titles(sizes_all, world, rank)
        streams = [flat_all[int(offs_all[i]):int(offs_all[i] + sizes_all[i])] for i in mine]
        flat, offs, sizes = hip.pack_streams(streams)
        frames = frames_all[mine]
        replicas, aus = 1, 1
    else:
        cfg = syn.make_cfg(assignment=assignment, rate_code=rate_code, n_substreams=args.substreams, n_aus=args.aus)
        flat, offs, sizes, frames = syn.batch(cfg, 1 + rank * args.streams, args.streams)
        replicas, aus = args.replicas, args.aus
    t_gen = time.perf_counter() - t_gen
    n_seg = len(sizes) * replicas * ((aus + cfg.restart_interval - 1) // cfg.restart_interval)
    b = Batch(pkg, torch, dev, local_rank, flat, offs, sizes, frames, np.full(len(sizes), nch), replicas, args.layout,
              0, max(n_seg, 64), depth=args.pipeline)   # 0: the library picks the kernels from the indexed counts

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[local_rank])
            else:
                dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        b.step()
    barrier()
    b.kernel_time()  # drop warmup launches
    t0 = time.perf_counter()
    for _ in range(args.steps):
        b.step()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches = b.kernel_time()
    # ---- the same step with nothing in flight beside it: one slot, the blocking decode call, waited for every time
    serial = None
    if args.pipeline > 1 and world == 1:
        ks = max(3, min(10, args.steps))
        b.sync()
        t1 = time.perf_counter()
        for _ in range(ks):
            b.ctxs[0].index(b.d_bytes.data_ptr(), b.total_bytes, b.d_off.data_ptr(), b.d_len.data_ptr(), b.n_streams, b.streams[0])
            b.ctxs[0].decode(b.d_pcms[0].data_ptr(), b.d_out_off.data_ptr(), b.d_stride.data_ptr(), b.streams[0])
            b.sync()
        t_serial = (time.perf_counter() - t1) / ks
        skms, _ = b.kernel_time()
        serial = {"ms_per_step": round(t_serial * 1e3, 4), "value": round(b.samples / t_serial / 1e6, 1),
                  "kernel_ms": round(skms, 4), "steps": ks,
                  "what": "index + blocking decode on ONE context, device idle in between (no overlap of one "
                          "batch's index with another's decode)"}

    # ---- every title must have decoded cleanly ...
    b.check_status()
    # ---- single-batch latency (c4): one step at a time, host clock around index + decode + sync
    latency = None
    if args.workload == "c4":
        lat = []
        for _ in range(50):
            b.sync()
            t1 = time.perf_counter()
            b.step()
            b.sync()
            lat.append(time.perf_counter() - t1)
        lat.sort()
        latency = {"median_ms": round(lat[len(lat) // 2] * 1e3, 4), "min_ms": round(lat[0] * 1e3, 4),
                   "p90_ms": round(lat[int(len(lat) * 0.9)] * 1e3, 4), "batches": len(lat),
                   "what": "one batch: index + decode enqueued and waited for, host clock"}

    # ---- ... and match the CPU decode bit for bit: every unique title against the all-core CPU leg
    #      (rank 0 at N = 1), the replicas against each other on the device; else a sample vs the oracle
    cpu = None
    bit_exact = None
    checked = 0
    full = (not args.no_cpu) and world == 1 and args.workload == "c3"
    if full:
        n_threads, logical_cpus, cpu_quota = usable_cpus()
        cpu, ref_pcm = cpu_legs(flat, offs, sizes, frames.astype(np.int64), assignment, rate_code, nch,
                                args.cpu_seconds, n_threads)
        per = b.pcm_words // b.R
        host = b.d_pcm[:per].cpu().numpy()
        bit_exact = True
        for i in range(len(sizes)):
            if not np.array_equal(b.title(host, i), ref_pcm[i]):
                bit_exact = False
                break
        checked = len(sizes)
        del host, ref_pcm
        bit_exact = bit_exact and b.replicas_equal() and b.slots_equal()
        if not bit_exact:
            raise SystemExit("HIP decode differs from the CPU decoder (%s)" % cpu[0]["kind"])
    elif args.verify or args.workload == "c4":
        picks = np.arange(b.n_streams) if args.workload == "c4" else \
            np.linspace(0, b.n_streams - 1, num=min(args.verify, b.n_streams), dtype=np.int64)
        bit_exact = b.verify_sample(flat, offs, sizes, picks)
        checked = len(picks)
        if not bit_exact:
            raise SystemExit("HIP decode differs from the oracle")

    # ---- whole-job aggregate: the path's one collective is this summary (RCCL all-reduce of a
    #      few words over xGMI; nothing on the data path is exchanged)
    checksum = int(b.d_pcm.to(torch.int64).sum().item())
    # (every rank's own k_decode time and roofline fraction travel with the summary: an N > 1 line carries them per rank)
    rank_frac = (b.comp_bytes + 4.0 * b.samples) / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else 0.0
    summ = pkg.shard.reduce_summary(dist if world > 1 else None, dev if backend == "nccl" else torch.device("cpu"),
                                    b.rows_total, b.samples, b.comp_bytes, 0, checksum, elapsed,
                                    verified=bool(bit_exact), checked=checked != 0,
                                    per_rank=(kernel_ms, rank_frac, elapsed / args.steps * 1e3))
    elapsed_max = summ["seconds"]
    job_samples = float(summ["samples"])

    if rank == 0:
        ms_per_step = elapsed_max / args.steps * 1e3
        value = job_samples * args.steps / elapsed_max / 1e6
        prof = committed_profile(b.samples, b.comp_bytes, args.layout) if args.workload == "c3" else None
        issue = None
        if prof and prof.get("valu_insts_per_launch") and kernel_ms > 0:
            clock_ghz = float(prof.get("shader_clock_ghz", 2.4))
            floor_ms = prof["valu_insts_per_launch"] * VALU_CYCLES_PER_WAVE_INST / (SIMDS * clock_ghz * 1e9) * 1e3
            issue = {"bound": "valu-issue", "valu_wave_insts_per_launch": prof["valu_insts_per_launch"],
                     "lane_insts_per_sample": round(prof["valu_insts_per_launch"] * 64 / b.samples, 1),
                     "floor_ms": round(floor_ms, 3), "frac": round(floor_ms / kernel_ms, 4),
                     "clock_ghz": clock_ghz,
                     "what": "SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x clock) vs measured k_decode ms; the count is "
                             "the committed PMC profile of this workload (profiles/traffic.json), not re-measured here"}
        out = {
            "metric": METRIC,
            "value": round(value, 1),
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if args.workload == "c4" else "weak",
            "vs_baseline": None,
            "dtype": "int32 (int64 accumulate)",
            "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[3]: %d independent single-access-unit 6ch/96kHz/24bit MLP streams "
                             "(major sync + restart + raw lead-in each), dealt to the ranks by size" % args.streams)
                if args.workload == "c4" else
                "BASELINE configs[2]: synthetic 6ch/96kHz/24bit MLP, 2 matrices + 8-tap FIR, "
                "codebook 1, CRC on, restart every 8 AUs",
                "substreams": args.substreams,
                "pcm_layout": "interleaved int32 [frame][channel] (reference dvda_read order)"
                              if args.layout == "interleaved" else
                              "planar int32 [channel][frame] (reference decode_packet order)",
                "titles_per_gpu": b.n_streams, "unique_titles_per_gpu": b.unique,
                "access_units_per_title": aus, "segments_per_gpu": n_seg,
                "samples_per_step_per_gpu": b.samples,
                "compressed_bytes_per_gpu": b.comp_bytes,
                "parallelism": "titles sharded over %d GPU(s), no data-path collective" % world,
                "pipeline": ("%d decode contexts in flight per GPU (own HIP stream and PCM buffer each, non-blocking "
                             "decode call): the index of batch n + 1 overlaps the decode of batch n" % args.pipeline)
                if args.pipeline > 1 else "one step after the other (blocking decode call)",
                "bit_exact": bit_exact,
                "bit_exact_titles_checked": checked,
                "bit_exact_against": (cpu[0]["kind"] + " CPU decode of every unique title; replicas compared on the device")
                if full else "oracle (sample)",
            },
            "roofline": roofline_record(b, kernel_ms, launches, prof["hbm_bytes_per_launch"] if prof else None, issue),
        }
        if world > 1:
            # the shard as it ran: the slowest and the fastest rank's step, the largest rank's share of the compressed
            # bytes against the mean (greedy longest-processing-time deal, shard.py), every rank's own sample check
            mean_bytes = summ["compressed_bytes"] / world
            out["ranks"] = {"ms_per_step_max": round(summ["seconds"] / args.steps * 1e3, 4),
                            "ms_per_step_min": round(summ["seconds_min"] / args.steps * 1e3, 4),
                            "compressed_bytes_max": summ["bytes_max"], "compressed_bytes_min": summ["bytes_min"],
                            "load_imbalance": round(summ["bytes_max"] / mean_bytes, 4) if mean_bytes else None,
                            "bit_exact_on_every_rank": summ["all_verified"],
                            "titles_checked_per_rank": checked,
                            # (round 5: the line says by itself that N ranks took part -- the process group's size and the
                            #  all-reduced count of ranks -- and what each of them measured)
                            "world_size": summ["world_size"], "ranks_counted": summ["ranks_counted"],
                            "per_rank": [{"rank": r, "k_decode_ms": round(v[0], 4), "roofline_frac": round(v[1], 5),
                                          "ms_per_step": round(v[2], 4)} for r, v in enumerate(summ["per_rank"])]}
            out["config"]["bit_exact"] = (bool(bit_exact) and bool(summ["all_verified"])) if checked else None
        if serial:
            out["serial_step"] = serial
        if latency:
            out["latency"] = latency
        if cpu:
            out["cpu_baseline"] = cpu[0]
            out["cpu_baseline_all_cores"] = cpu[1]
            cpu[1]["host_logical_cpus"] = logical_cpus
            cpu[1]["cgroup_cpu_quota"] = cpu_quota
            out["speedup_vs_cpu"] = {"vs_1_core": round(value / cpu[0]["value"], 1),
                                     "vs_%d_cores_usable_here" % cpu[1]["cores"]: round(value / cpu[1]["value"], 1),
                                     "n_gpus": world}
            if cpu_quota is not None and logical_cpus > n_threads:
                # the container is limited to `cores` CPUs of a bigger host: what the whole host would do if
                # the measured per-thread rate held on every logical CPU (an upper bound: SMT siblings share a
                # core) -- stated as an extrapolation, not a measurement
                per_thread = cpu[1]["value"] / cpu[1]["cores"]
                out["speedup_vs_cpu"]["vs_all_%d_host_cpus_extrapolated" % logical_cpus] = round(
                    value / (per_thread * logical_cpus), 1)
        out["host"] = {"gen_seconds": round(t_gen, 2), "cpus": os.cpu_count()}
        if world == 1 and not args.no_sub and args.workload == "c3" and args.substreams == 1 and assignment == 12:
            sub_steps = max(5, min(args.steps, 20))
            # (the host-path records first: behind the other sub-records -- dozens of contexts and streams opened and
            #  closed -- the same passes ran 26 ms where they take 18-21 in a fresh process, copies-only unchanged.  The
            #  runtime's GPU_MAX_HW_QUEUES is left at its default of 4: with 8 the pass's five streams get a hardware
            #  queue each and it runs 13.9 instead of 11.6 Gsamples/s, but the records that overlap two or three kernels
            #  on purpose pay for it -- mixed_corpus_c5 429 -> 370 --, and in a child process of its own, beside this
            #  one's idle context, the pass was slower than either: measured, round 6)
            t_h = time.perf_counter()
            only = set(x for x in args.only_sub.split(",") if x)
            h2h = {}
            if not only or "host_to_host" in only:
                h2h["host_to_host"] = host_to_host(pkg, torch, dev, local_rank, args, flat, offs, sizes, frames,
                                                   max(3, sub_steps // 4))
            if not only or "host_to_host_wav24" in only:
                h2h["host_to_host_wav24"] = host_to_host(pkg, torch, dev, local_rank, args, flat, offs, sizes, frames,
                                                         max(3, sub_steps // 4), wav24=True)
            sys.stderr.write("bench: host_to_host records %.1f s\n" % (time.perf_counter() - t_h))
            out["sub"] = sub_records(pkg, torch, dev, local_rank, args, b, sub_steps, 2)
            out["sub"].update(h2h)
            if not only or "streaming_tier" in only:
                from tests import oracle_lib as _ol
                out["sub"]["streaming_tier"] = streaming_tier(pkg, _ol.Oracle())
            if (not only or "disc_tier" in only) and not args.no_disc:
                from tests import oracle_lib as _ol
                t_d = time.perf_counter()
                out["sub"]["disc_tier"] = disc_tier(pkg, _ol.Oracle())
                sys.stderr.write("bench: disc tier record %.1f s\n" % (time.perf_counter() - t_d))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
