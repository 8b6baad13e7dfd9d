#!/usr/bin/env python3
"""tools/soak_stream.py [N] -- soak of the streaming tier (mlp.h mirror) and of the disc tier on the
GPU box: N random generator configurations, each fed to dvda_hip_mlpdecoder_decode_packet in
random-sized packets (per-call PCM-frame counts and PCM against the oracle fed the same packets),
and cut into random tracks on a synthetic disc (the tracks must give back the stream).  Diagnostic."""
import ctypes
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libdvd_audio_amd as pkg  # noqa: E402
from tests import oracle_lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
oracle = oracle_lib.Oracle()
syn, hip, disc = pkg.synth, pkg.hipdec, pkg.disc
ol = oracle.lib
ol.mlp_oracle_open.restype = ctypes.c_void_p
ol.mlp_oracle_decode_packet.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
ol.mlp_oracle_decode_packet.restype = ctypes.c_uint
ol.mlp_oracle_close.argtypes = [ctypes.c_void_p]
rng = np.random.RandomState(777)
two = [12, 1, 0x14, 6, 9, 3, 17, 20]
bad = 0
for i in range(n):
    S = 1 + (i & 1)
    asg = int(rng.choice(two)) if S == 2 else int(rng.randint(0, 21))
    feats = int(rng.randint(0, 1 << 18)) if i % 3 else syn.SF_ALL
    if i % 7 == 3:
        feats |= syn.SF["SYNCONLY"]          # major syncs that restart nothing (src/mlp.c:449-460)
    cfg = syn.make_cfg(assignment=asg, rate_code=int(rng.randint(0, 3)), n_substreams=S,
                       n_aus=int(rng.randint(6, 48)), profile=1, features=feats,
                       restart_interval=int(rng.randint(1, 9)))
    data, frames = syn.stream(cfg, 20000 + i)
    nch = syn.channels(asg)
    want, r, st = oracle.decode(data, nch, frames)
    if st != 0 or r != frames:
        continue                    # (the generator can split the two substreams of an access unit differently: reported by both sides)
    # ---- tier B, random packet sizes
    od = ol.mlp_oracle_open(nch)
    dec = hip.MLPDecoder(cfg.bps_code, cfg.bps_code, cfg.rate_code, cfg.rate_code, cfg.assignment)
    samples = [[] for _ in range(nch)]
    off, ok = 0, True
    while off < len(data):
        m = int(rng.choice([1, 7, 100, 777, 2013, 2013, 5000, 20000]))
        piece = np.ascontiguousarray(data[off:off + m])
        want_n = ol.mlp_oracle_decode_packet(od, piece.ctypes.data, len(piece))
        got_n = dec.decode_packet(piece, samples)
        if got_n != want_n or dec.status & ~hip.ST_BENIGN:
            ok = False
            print("STREAM MISMATCH case", i, "asg", asg, "S", S, "feats", hex(feats), "at", off, got_n, want_n, hex(dec.status))
            break
        off += m
    ol.mlp_oracle_close(od)
    dec.close()
    if ok:
        got = np.asarray(samples, np.int32).reshape(nch, -1)
        if got.shape != want.shape or not np.array_equal(got, want):
            ok = False
            print("STREAM PCM MISMATCH case", i, got.shape, want.shape)
    # ---- disc tier, random track cuts (every 4th case: the file I/O dominates)
    if ok and i % 4 == 0:
        secs = disc.mlp_track_sectors(data)
        if len(secs) >= 4:
            k = int(rng.randint(1, min(4, len(secs) - 1) + 1))
            cuts = sorted(set(int(c) for c in rng.randint(1, len(secs), size=k)))
            with tempfile.TemporaryDirectory() as tmp:
                tracks = disc.split_tracks(secs, cuts, [frames // (len(cuts) + 1)] * (len(cuts) + 1), cfg.rate_code)
                ats = disc.write_disc_titles(tmp, [tracks])
                parts = []
                for t in range(1, len(tracks) + 1):
                    try:
                        parts.append(pkg.discdec.read_track(ats, 1, 1, t)["pcm"])
                    except RuntimeError:
                        parts.append(None)          # a track without a major sync of its own
                got = [p for p in parts if p is not None]
                cat = np.concatenate(got).T if got else np.zeros((nch, 0), np.int32)
                # tracks with no sync inside are unreadable in the reference too; the readable ones
                # must tile the stream when every track has one
                if all(p is not None for p in parts) and (cat.shape != want.shape or not np.array_equal(cat, want)):
                    ok = False
                    print("DISC MISMATCH case", i, "cuts", cuts, cat.shape, want.shape)
    bad += 0 if ok else 1
print("soak_stream: %d cases, %d mismatches" % (n, bad))
sys.exit(1 if bad else 0)
