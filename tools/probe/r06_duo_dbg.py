#!/usr/bin/env python3
"""tools/probe/r06_duo_dbg.py -- per-segment status of two-substream streams decoded by the one-lane two-substream
kernels (k_decode<.., DUO>), against the oracle.  Diagnostic."""
import ctypes
import glob
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("libdvd-audio_amd")
hip, syn = pkg.hipdec, pkg.synth


class SegInfo(ctypes.Structure):
    _fields_ = [("offset", ctypes.c_uint64), ("end", ctypes.c_uint64), ("stream", ctypes.c_uint32),
                ("mlp_frames", ctypes.c_uint32), ("pcm_frames", ctypes.c_uint32), ("status", ctypes.c_uint32)]


def run(name, streams, lanes):
    L = hip.lib()
    total = sum((len(s) + 15) & ~15 for s in streams)
    ctx = hip.Context(0, len(streams), max(64, total // 64), lanes, hip.PCM_PLANAR)
    pcm, infos = hip.decode_streams(streams, lanes_per_segment=lanes, ctx=ctx)
    n = ctx.segment_count()
    print("== %s: %d streams, %d segments, lanes=%d" % (name, len(streams), n, lanes))
    for i, inf in enumerate(infos[:8]):
        print("  stream %d status %#x frames %d pcm %d" % (i, inf.status, inf.mlp_frames, inf.pcm_frames))
    si = SegInfo()
    L.dvda_mlp_hip_segment_info.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    for s in range(min(n, 12)):
        L.dvda_mlp_hip_segment_info(ctx._h, s, ctypes.byref(si), None)
        print("    seg %d stream %d off %d frames %d pcm %d status %#x" % (s, si.stream, si.offset, si.mlp_frames, si.pcm_frames, si.status))
    ctx.close()
    return pcm, infos


SF = syn.SF
for feats, nm in ((0, "plain_2ss"), (SF["CHAINED"], "chained_2ss")):
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=24, profile=1, features=feats)
    flat, offs, sizes, frames = syn.batch(cfg, 1, 3)
    streams = [flat[int(o):int(o) + int(z)] for o, z in zip(offs, sizes)]
    pcm, infos = run(nm, streams, 2)
    from importlib import import_module
    ora = import_module("tests.oracle_util") if os.path.exists(os.path.join(ROOT, "tests", "oracle_util.py")) else None
for p in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*2ss*.npz")))[:6]:
    z = np.load(p)
    pcm, infos = run(os.path.basename(p), [z["mlp"]], 2)
    exp = z["pcm"] if "pcm" in z else None
    if exp is not None and pcm[0].shape == exp.shape:
        print("   pcm equal:", bool((pcm[0] == exp).all()))
    elif exp is not None:
        print("   pcm shape", pcm[0].shape, "expected", exp.shape)

# a larger plain two-substream batch: does anything defer?
cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=64)
flat, offs, sizes, frames = syn.batch(cfg, 1, 512)
streams = [flat[int(o):int(o) + int(z)] for o, z in zip(offs, sizes)]
pcm, infos = hip.decode_streams(streams, lanes_per_segment=0)
import collections
print("plain 512 x 64 two-substream titles: status histogram", collections.Counter(hex(i.status) for i in infos))
