#!/usr/bin/env python3
"""tools/stream_bench.py [chained] -- speed of tier B (the mlp.h mirror: dvda_hip_mlpdecoder_decode_packet) fed one
6-ch / 96 kHz title in PES-payload sized packets (2 011 bytes), calling the C entry point directly (no Python list
building): Msamples/s and ms per call.  Diagnostic."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libdvd_audio_amd as pkg  # noqa: E402

syn, hip = pkg.synth, pkg.hipdec
chained = len(sys.argv) > 1 and sys.argv[1] == "chained"
cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=1024, **(dict(profile=1, features=syn.SF["CHAINED"]) if chained else {}))
b, frames = syn.stream(cfg, 5)
dec = hip.MLPDecoder(2, 2, 1, 1, 12)
L = hip.lib()
planar = (ctypes.POINTER(ctypes.c_int32) * 6)()
nch = ctypes.c_uint()
h = dec._h
pieces = [np.ascontiguousarray(b[o:o + 2011]) for o in range(0, len(b), 2011)]
for rep in range(2):
    if rep:
        dec.close()
        dec = hip.MLPDecoder(2, 2, 1, 1, 12)
        h = dec._h
    got = 0
    t0 = time.perf_counter()
    for p in pieces:
        got += L.dvda_hip_mlpdecoder_decode_packet(h, p.ctypes.data, len(p), planar, ctypes.byref(nch))
    dt = time.perf_counter() - t0
print("tier B%s: %d PCM frames of %d in %d calls, %.3f ms per call, %.2f Msamples/s" % (
    " (chained title)" if chained else "", got, frames, len(pieces), dt / len(pieces) * 1e3, got * 6 / dt / 1e6))
