#!/usr/bin/env python3
"""tools/probe/fused_dbg.py -- where does k_chain_fused differ from the oracle?  (diagnostic)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import libdvd_audio_amd as pkg
from tests import oracle_lib
syn, hip = pkg.synth, pkg.hipdec
ora = oracle_lib.Oracle()
asg, S, rate = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
for seed in range(4):
    cfg = syn.make_cfg(assignment=asg, rate_code=rate, n_substreams=S, n_aus=24, profile=1,
                       features=syn.SF_FAST, restart_interval=[8, 3, 16, 5][seed])
    b, f = syn.stream(cfg, 100 + seed)
    nch = syn.channels(cfg.assignment)
    want, r, st = ora.decode(b, nch, f)
    for rep in range(2):
        for lay in (hip.PCM_PLANAR, hip.PCM_INTERLEAVED):
            pcm, infos = hip.decode_streams([b], layout=lay, lanes_per_segment=2)
            d = np.argwhere(pcm[0] != want)
            rows = sorted(set(d[:, 1].tolist()))
            print("seed %d rep %d layout %d status %#x: %d wrong values, rows %s .. %s, channels %s" % (
                seed, rep, lay, infos[0].status, len(d), rows[:12], rows[-3:], sorted(set(d[:, 0].tolist()))))
