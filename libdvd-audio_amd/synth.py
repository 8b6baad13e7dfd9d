"""ctypes binding of the synthetic MLP stream generator (synth/mlp_synth.c)."""
import ctypes
import os

import numpy as np

from . import _build

SF = dict(IIR=1 << 0, QSS=1 << 1, OUTSHIFT=1 << 2, HUFFOFF=1 << 3, VARBLOCK=1 << 4, VARROWS=1 << 5,
          MIDMATRIX=1 << 6, CHAINED=1 << 7, EXTRAWORD=1 << 8, MIDRESTART=1 << 9, FLAGS=1 << 10,
          MIXBOOKS=1 << 11, NOCHECK=1 << 12, PARAMBLOCKS=1 << 13, MATRIXRAND=1 << 14,
          TERMINATOR=1 << 15, FIRRAND=1 << 16, NOISE=1 << 17, CHECKQUIRK=1 << 18, DISC=1 << 19, SYNCONLY=1 << 20)
SF_ALL = (1 << 18) - 1
# features the fused kernel decodes without its reporting paths (standard timing,
# raw lead-in per segment, matrix-class parameters constant inside a frame)
SF_FAST = SF_ALL & ~(SF["VARROWS"] | SF["MIDMATRIX"] | SF["CHAINED"] | SF["MIDRESTART"])


class Cfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in
                ("profile features assignment rate_code bps_code n_substreams ss0_channels n_aus "
                 "restart_interval blocks_per_au fir_order codebook huffman_lsbs n_matrices r0 r1").split()]


_lib = None


def lib():
    global _lib
    if _lib is None:
        so = _build.build_synth()
        L = ctypes.CDLL(so)
        L.mlp_synth_bound.restype = ctypes.c_size_t
        L.mlp_synth_bound.argtypes = [ctypes.POINTER(Cfg)]
        L.mlp_synth_stream.restype = ctypes.c_size_t
        L.mlp_synth_stream.argtypes = [ctypes.POINTER(Cfg), ctypes.c_uint64, ctypes.c_void_p,
                                       ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint64)]
        L.mlp_synth_batch.restype = ctypes.c_size_t
        L.mlp_synth_batch.argtypes = [ctypes.POINTER(Cfg), ctypes.c_uint64, ctypes.c_uint32,
                                      ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mlp_synth_default.argtypes = [ctypes.POINTER(Cfg)] + [ctypes.c_uint32] * 4
        L.mlp_synth_channels.restype = ctypes.c_uint
        L.mlp_synth_channels.argtypes = [ctypes.c_uint32]
        L.mlp_synth_rows_per_au.restype = ctypes.c_uint
        L.mlp_synth_rows_per_au.argtypes = [ctypes.c_uint32]
        _lib = L
    return _lib


def channels(assignment):
    return int(lib().mlp_synth_channels(assignment))


def rows_per_au(rate_code):
    return int(lib().mlp_synth_rows_per_au(rate_code))


def make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=64, profile=0, features=0, **kw):
    c = Cfg()
    lib().mlp_synth_default(ctypes.byref(c), assignment, rate_code, n_substreams, n_aus)
    c.profile = profile
    c.features = features
    if n_substreams == 2 and "ss0_channels" not in kw:
        c.ss0_channels = max(1, min(2, channels(assignment) - 1))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def stream(cfg, seed):
    """-> (bytes as uint8 ndarray, pcm_frames)"""
    L = lib()
    cap = L.mlp_synth_bound(ctypes.byref(cfg))
    buf = np.zeros(cap, np.uint8)
    fr = ctypes.c_uint64()
    n = L.mlp_synth_stream(ctypes.byref(cfg), seed, buf.ctypes.data, cap, ctypes.byref(fr))
    if n == 0:
        raise RuntimeError("mlp_synth_stream failed")
    return buf[:n].copy(), int(fr.value)


def batch(cfg, seed0, n, threads=None):
    """-> (bytes ndarray with 64 spare zero bytes, offsets[n], sizes[n], frames[n]) ; streams 16-byte aligned"""
    L = lib()
    if threads is None:
        threads = min(os.cpu_count() or 1, 32)
    bound = ((L.mlp_synth_bound(ctypes.byref(cfg)) + 15) & ~15) * n + 64
    buf = np.zeros(bound, np.uint8)
    off = np.zeros(n, np.uint64)
    siz = np.zeros(n, np.uint64)
    frm = np.zeros(n, np.uint64)
    tot = L.mlp_synth_batch(ctypes.byref(cfg), seed0, n, threads, buf.ctypes.data, bound - 64,
                            off.ctypes.data, siz.ctypes.data, frm.ctypes.data)
    if tot == 0:
        raise RuntimeError("mlp_synth_batch failed")
    return buf[:tot + 64], off, siz, frm
