#!/usr/bin/env python3
"""tools/coop_bench.py -- small-batch shapes through the library's own kernel choice (lanes 0), the lane kernels
(lanes 1) and the wave-cooperative kernel (lanes 64): configs[3] (1 024 single-unit streams), 64 / 128 / 256 titles
of 512 units, ONE title.  Prints ms per batch (index + decode, median of 30, host clock) and Msamples/s.  Diagnostic."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import libdvd_audio_amd as pkg  # noqa: E402
from bench import Batch  # noqa: E402

syn = pkg.synth
dev = torch.device("cuda", 0)


def run(name, cfg, n, lanes, seed=1):
    flat, offs, sizes, frames = syn.batch(cfg, seed, n)
    nch = syn.channels(cfg.assignment)
    nseg = n * ((cfg.n_aus + cfg.restart_interval - 1) // cfg.restart_interval)
    b = Batch(pkg, torch, dev, 0, flat, offs, sizes, frames, np.full(n, nch), 1, "interleaved", lanes, max(nseg, 64))
    for _ in range(5):
        b.step()
    b.sync()
    lat = []
    for _ in range(30):
        b.sync()
        t = time.perf_counter()
        b.step()
        b.sync()
        lat.append(time.perf_counter() - t)
    lat.sort()
    dt, kms, _ = b.timed(20, 2)
    b.check_status(benign=pkg.hipdec.ST_BENIGN)
    ok = b.verify_sample(flat, offs, sizes, np.linspace(0, n - 1, num=min(n, 16), dtype=np.int64))
    print("%-28s lanes %2d: median %.4f ms  back-to-back %.4f ms/step  kernel %.4f ms  %.0f Msamples/s  bit-exact %s" % (
        name, lanes, lat[len(lat) // 2] * 1e3, dt / 20 * 1e3, kms, b.samples / (dt / 20) / 1e6, ok), flush=True)
    b.close()


which = sys.argv[1:] or ["c4", "t64", "t128", "t256", "one"]
for lanes in (1, 64, 0):
    if "c4" in which:
        run("c4: 1024 x 1 unit", syn.make_cfg(assignment=12, rate_code=1, n_aus=1), 1024, lanes)
    if "t64" in which:
        run("64 titles x 512 units", syn.make_cfg(assignment=12, rate_code=1, n_aus=512), 64, lanes)
    if "t128" in which:
        run("128 titles x 512 units", syn.make_cfg(assignment=12, rate_code=1, n_aus=512), 128, lanes)
    if "t256" in which:
        run("256 titles x 512 units", syn.make_cfg(assignment=12, rate_code=1, n_aus=512), 256, lanes)
    if "one" in which:
        run("ONE title x 512 units", syn.make_cfg(assignment=12, rate_code=1, n_aus=512), 1, lanes)
