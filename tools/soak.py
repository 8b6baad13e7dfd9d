#!/usr/bin/env python3
"""tools/soak.py [N [SEED]] -- long parity soak on the GPU box: N random generator configurations (all
feature sets, all channel assignments, 1 and 2 substreams, three sample rates), each decoded by the
batch tier and compared bit for bit with the oracle.  Diagnostic; the committed tests hold a
fixed subset of the same cases."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libdvd_audio_amd as pkg  # noqa: E402
from tests import oracle_lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
oracle = oracle_lib.Oracle()
syn, hip = pkg.synth, pkg.hipdec
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 12345
rng = np.random.RandomState(seed0)
two = [12, 1, 0x14, 6, 9, 3, 17, 20]
bad = 0
batch, meta = [], []
for i in range(n):
    S = 1 + (i & 1)
    asg = int(rng.choice(two)) if S == 2 else int(rng.randint(0, 21))
    feats = int(rng.randint(0, 1 << 18)) if i % 3 else syn.SF_ALL
    if i % 7 == 3:
        feats |= syn.SF["SYNCONLY"]          # major syncs that restart nothing (src/mlp.c:449-460)
    if i % 5 == 0:
        feats &= syn.SF_FAST
    cfg = syn.make_cfg(assignment=asg, rate_code=int(rng.randint(0, 3)), n_substreams=S,
                       n_aus=int(rng.randint(4, 40)), profile=1, features=feats,
                       restart_interval=int(rng.randint(1, 9)))
    try:
        b, f = syn.stream(cfg, 10000 + i + (seed0 - 12345) * 7919)
    except Exception as e:          # generator refuses a combination
        continue
    batch.append(b)
    meta.append((i, asg, S, feats, f))
    if len(batch) == 32 or i == n - 1:
        # batches alternate between the two PCM layouts of the C ABI
        pcm, infos = hip.decode_streams(batch, lanes_per_segment=[0, 2][(i // 64) & 1],   # the library picks the kernels / forced lane pairs
                                        layout=hip.PCM_INTERLEAVED if (i // 32) & 1 else hip.PCM_PLANAR)
        for b, (idx, asg, S, feats, f), p, inf in zip(batch, meta, pcm, infos):
            want, r, st = oracle.decode(b, syn.channels(asg), f)
            ok = st == 0 and (inf.status & ~hip.ST_BENIGN) == 0 and p.shape == want.shape and np.array_equal(p, want)
            if not ok:
                bad += 1
                print("MISMATCH case", idx, "asg", asg, "S", S, "feats", hex(feats), "oracle st", hex(st), "r", r,
                      "gpu st", hex(inf.status), "frames", p.shape[1])
        batch, meta = [], []
print("soak: %d cases, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
