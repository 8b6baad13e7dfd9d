#!/usr/bin/env python3
"""tools/chain_bench.py -- decode time of a batch whose segments all chain (no raw lead-in after a
title's first segment: the FIR history runs through the whole title), i.e. the general pass's
worst case, beside the same batch with independent segments.  Diagnostic."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import libdvd_audio_amd as pkg  # noqa: E402

syn, hip = pkg.synth, pkg.hipdec
n_titles, n_aus = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 128
n_ss = int(sys.argv[3]) if len(sys.argv) > 3 else 1       # substreams per title (2: ch 0-1 | ch 2-5, what 6-channel discs carry)
layout = int(sys.argv[4]) if len(sys.argv) > 4 else 0     # 0 planar, 1 frame-major (hipdec.PCM_*)
dev = torch.device("cuda", 0)
for name, feats in (("independent", 0), ("chained", syn.SF["CHAINED"])):
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=n_ss, n_aus=n_aus, profile=1 if feats else 0, features=feats)
    flat, offs, sizes, frames = syn.batch(cfg, 1, n_titles)
    d_bytes = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_len = torch.from_numpy(sizes.astype(np.int64)).to(dev)
    out_off = np.zeros(n_titles, np.int64)
    out_off[1:] = np.cumsum(frames[:-1].astype(np.int64) * 6)
    d_oo = torch.from_numpy(out_off).to(dev)
    d_st = torch.from_numpy(frames.astype(np.int64)).to(dev)
    d_pcm = torch.empty(int(frames.sum()) * 6, dtype=torch.int32, device=dev)
    ctx = hip.Context(0, n_titles, n_titles * (n_aus // 8 + 2), lanes_per_segment=0, layout=layout)
    best = 1e9
    for it in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ctx.index(d_bytes.data_ptr(), len(flat) - 64, d_off.data_ptr(), d_len.data_ptr(), n_titles, 0)
        ctx.decode(d_pcm.data_ptr(), d_oo.data_ptr(), d_st.data_ptr(), 0)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    infos = ctx.stream_info()
    bad = sum(1 for i in infos if i.status & ~hip.ST_BENIGN)
    samples = int(frames.sum()) * 6
    print("%-12s %5d titles x %4d AUs: %8.2f ms  %9.1f Msamples/s  (streams with errors: %d, chained flag on %d)" % (
        name, n_titles, n_aus, best * 1e3, samples / best / 1e6, bad, sum(1 for i in infos if i.status & hip.ST["CHAINED"])))
    ctx.close()
