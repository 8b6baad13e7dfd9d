#!/usr/bin/env python3
"""tools/soak_pcm.py [N] -- raw-PCM and demux kernels on the GPU box against the oracle: random
layouts (1-6 channels, 16/24 bit, ragged lengths), then the same sectors with header bytes
corrupted -- the call must return, and everything in front of the first damaged sector must still
be identical (the reference stops there; the kernels skip the sector and go on).  Diagnostic."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libdvd_audio_amd as pkg  # noqa: E402
from tests import oracle_lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
hip, disc, syn = pkg.hipdec, pkg.disc, pkg.synth
lib = ctypes.CDLL(oracle_lib.build_oracle())
lib.pcm_oracle_decode_sectors.restype = ctypes.c_long
lib.pcm_oracle_decode_sectors.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint,
                                          ctypes.c_void_p, ctypes.c_size_t]
rng = np.random.RandomState(31)
bad = 0
for it in range(n):
    ch = 1 + it % 6
    bps_code = 0 if it % 2 else 2
    bits = 16 if bps_code == 0 else 24
    asg = {1: 0, 2: 1, 3: 2, 4: 3, 5: 6, 6: 12}[ch]
    fr = 2 * int(rng.randint(1, 3000))
    s = rng.randint(-(1 << (bits - 1)), 1 << (bits - 1), size=(fr, ch))
    data = np.frombuffer(b"".join(disc.pcm_track_sectors(s, bps_code, 1, asg)), np.uint8).copy()
    out, nbad = hip.pcm_decode_sectors(data, bits, ch)
    if nbad or out.shape[1] != fr or not np.array_equal(out, s.T):
        bad += 1
        print("PCM MISMATCH", it, ch, bits, fr, out.shape, nbad)
        continue
    # damage one sector's headers
    nsec = len(data) // 2048
    victim = int(rng.randint(0, nsec))
    d2 = data.copy()
    for _ in range(int(rng.randint(1, 6))):
        d2[victim * 2048 + int(rng.randint(0, 40))] = int(rng.randint(0, 256))
    out2, nbad2 = hip.pcm_decode_sectors(d2, bits, ch)
    want = np.zeros((ch, fr + 2100), np.int32)
    r = lib.pcm_oracle_decode_sectors(d2.ctypes.data, nsec, bits, ch, want.ctypes.data, want.shape[1])
    per = (2048 - 14 - 6 - 7 - 9) // (2 * ch * bits // 8) * 2
    safe = min(victim * per, out2.shape[1], max(r, 0))
    if not np.array_equal(out2[:, :safe], s.T[:, :safe]):
        bad += 1
        print("PCM CORRUPT-PREFIX MISMATCH", it, victim, safe)
# demux: MLP payload through damaged sectors
for it in range(n // 4):
    b, f = syn.stream(syn.make_cfg(assignment=12, rate_code=1, n_aus=int(rng.randint(4, 60))), 900 + it)
    secs = np.frombuffer(b"".join(disc.mlp_track_sectors(b)), np.uint8).copy()
    got, nb = hip.mlp_demux_sectors(secs)
    if nb or len(got) != len(b) or not np.array_equal(got, b):
        bad += 1
        print("DEMUX MISMATCH", it)
    d2 = secs.copy()
    victim = int(rng.randint(0, len(d2) // 2048))
    d2[victim * 2048 + int(rng.randint(0, 30))] ^= 0xFF
    got2, nb2 = hip.mlp_demux_sectors(d2)
    room = 2048 - 14 - 6 - 7
    if not np.array_equal(got2[:victim * room], b[:victim * room]):
        bad += 1
        print("DEMUX CORRUPT-PREFIX MISMATCH", it, victim)
print("soak_pcm: %d cases, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
