#!/usr/bin/env python3
"""tools/shape_bench.py [SHAPE ...] -- one bench sub-record shape at a time, for A/B builds (DVDA_MLP_HIP_LIB):
fuzz_fast, fuzz_all, hetero_short, chained (1 024 x 128), chained2 (4 096 x 512, two substreams), two (two substreams).
Prints Msamples/s, ms per step, whole-call and fast-pass kernel ms.  Diagnostic."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import libdvd_audio_amd as pkg  # noqa: E402
from bench import Batch, gen_mixed  # noqa: E402

syn, hip = pkg.synth, pkg.hipdec
dev = torch.device("cuda", 0)
SF = syn.SF


def run(name, flat, offs, sizes, frames, nchs, nseg, replicas, layout="planar", lanes=0, steps=10):
    b = Batch(pkg, torch, dev, 0, flat, offs, sizes, frames, nchs, replicas, layout, lanes, nseg)
    dt, kms, _ = b.timed(steps, 2)
    b.check_status(benign=hip.ST_BENIGN)
    ok = b.verify_sample(flat, offs, sizes, np.linspace(0, b.n_streams - 1, num=min(8, b.n_streams), dtype=np.int64))
    print("%-14s %9.1f Msamples/s  %8.3f ms/step  call %8.3f ms  fast pass %8.3f ms  bit-exact %s" % (
        name, b.samples * steps / dt / 1e6, dt / steps * 1e3, b.last_decode_ms, kms, ok), flush=True)
    b.close()
    del b
    torch.cuda.empty_cache()


shapes = [(12, 1, 8), (1, 1, 5), (12, 2, 16), (0x12, 0, 3), (12, 0, 8), (6, 1, 4), (0, 2, 8), (12, 1, 2)]
which = sys.argv[1:] or ["fuzz_fast", "fuzz_all", "chained", "two"]
for w in which:
    if w in ("fuzz_fast", "fuzz_all"):
        feats = syn.SF_FAST & ~(SF["IIR"] | SF["MATRIXRAND"]) if w == "fuzz_fast" else syn.SF_FAST
        specs = [(syn.make_cfg(assignment=a, rate_code=rc, n_substreams=1, n_aus=64, profile=1, features=feats,
                               restart_interval=ri), 512) for a, rc, ri in shapes]
        flat, offs, sizes, frames, nchs, nseg = gen_mixed(syn, specs, 90000)
        run(w, flat, offs, sizes, frames, nchs, nseg * 4, 4)
    elif w.startswith("fuzzone_"):
        # fuzzone_K_R: only shape K of the fuzz batch's eight, 512 titles x R replicas (how long ONE segment shape takes
        # with the chip to itself: 512 titles of shape 2 are 32 waves)
        _, k, r = w.split("_")
        a, rc, ri = shapes[int(k)]
        feats = syn.SF_FAST & ~(SF["IIR"] | SF["MATRIXRAND"])
        cfg = syn.make_cfg(assignment=a, rate_code=rc, n_substreams=1, n_aus=64, profile=1, features=feats, restart_interval=ri)
        flat, offs, sizes, frames, nchs, nseg = gen_mixed(syn, [(cfg, 512)], 90000)
        run(w, flat, offs, sizes, frames, nchs, nseg * int(r), int(r))
    elif w == "chained":
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=128, profile=1, features=SF["CHAINED"])
        flat, offs, sizes, frames = syn.batch(cfg, 1, 1024)
        run(w, flat, offs, sizes, frames, np.full(1024, 6), 1024 * 17, 1, "interleaved")
    elif w in ("chained1", "chained2", "chained1_500", "chained2_500"):
        # (_500: titles of 500 access units -- the chains' workspaces then do not start at multiples of a power of two)
        S = 1 if w.startswith("chained1") else 2
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=500 if w.endswith("_500") else 512, profile=1,
                           features=SF["CHAINED"])
        flat, offs, sizes, frames = syn.batch(cfg, 1, 1024)
        run(w, flat, offs, sizes, frames, np.full(1024, 6), 4096 * 65, 4, "interleaved")
    elif w == "stereo":
        # the bench's stereo_c2 sub-record: 2-channel titles (BASELINE configs[1]), planar layout
        cfg = syn.make_cfg(assignment=1, rate_code=1, n_substreams=1, n_aus=512)
        flat, offs, sizes, frames = syn.batch(cfg, 1, 1024)
        run(w, flat, offs, sizes, frames, np.full(1024, 2), 4096 * 64, 4, "planar")
    elif w == "two":
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=512)
        flat, offs, sizes, frames = syn.batch(cfg, 1, 1024)
        run(w, flat, offs, sizes, frames, np.full(1024, 6), 4096 * 64, 4, "interleaved")
    elif w.startswith("one_chained"):
        # one_chained = ONE chained title of 512 access units, one_chained_long of 8 192, one_chained_N of N
        n_aus = 512 if w == "one_chained" else 8192 if w == "one_chained_long" else int(w.split("_")[-1])
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=n_aus, profile=1, features=SF["CHAINED"])
        flat, offs, sizes, frames = syn.batch(cfg, 1, 1)
        run(w, flat, offs, sizes, frames, np.full(1, 6), n_aus // 8 + 8, 1, "interleaved", steps=20)
    elif w.startswith("headline"):
        # headline = the bench batch (4 096 titles: two rounds of two waves per SIMD); headlineN = N x 512 titles, i.e.
        # N x 512 waves of k_decode on the chip's 1 024 SIMDs (headline2: one wave per SIMD, headline4: two)
        half = int(w[8:] or 8)
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=512)
        flat, offs, sizes, frames = syn.batch(cfg, 1, 512)
        run(w, flat, offs, sizes, frames, np.full(512, 6), 512 * half * 64, half, "interleaved")
