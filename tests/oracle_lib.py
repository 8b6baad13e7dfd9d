"""Loader for the CPU oracle (oracle/libmlp_oracle.so) and, when it has been built in
the dev container, the compiled reference (oracle/_ref/libdvda_ref.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libmlp_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libdvda_ref.so")

CHANNELS = [1, 2, 3, 4, 3, 4, 5, 3, 4, 5, 4, 5, 6, 4, 5, 4, 5, 6, 5, 5, 6]


def build_oracle():
    src = os.path.join(ORACLE_DIR, "mlp_oracle.c")
    if (not os.path.exists(ORACLE_SO)) or os.path.getmtime(src) > os.path.getmtime(ORACLE_SO):
        subprocess.run(["make", "-C", ORACLE_DIR, "libmlp_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return ORACLE_SO


class Oracle:
    def __init__(self):
        self.lib = ctypes.CDLL(build_oracle())
        self.lib.mlp_oracle_decode.restype = ctypes.c_long
        self.lib.mlp_oracle_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t,
                                               ctypes.c_uint, ctypes.c_void_p, ctypes.c_size_t,
                                               ctypes.POINTER(ctypes.c_uint)]

    def decode(self, data, nch, max_frames, chunk=0):
        """-> (pcm int32 [nch, frames], frames, status)"""
        data = np.ascontiguousarray(data, np.uint8)
        cap = int(max_frames) + 16
        out = np.zeros((nch, cap), np.int32)
        st = ctypes.c_uint()
        r = self.lib.mlp_oracle_decode(data.ctypes.data, len(data), chunk, nch, out.ctypes.data, cap,
                                       ctypes.byref(st))
        if r < 0:
            raise RuntimeError("oracle output capacity too small")
        return out[:, :r].copy(), int(r), int(st.value)

    def wav_pack(self, planar, bits):
        """planar int32 [channels, frames] -> the WAV payload bytes dvda2wav writes (oracle/pcm_oracle.c)"""
        planar = np.ascontiguousarray(planar, np.int32)
        ch, frames = planar.shape
        self.lib.wav_oracle_pack.restype = ctypes.c_long
        self.lib.wav_oracle_pack.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_size_t,
                                             ctypes.c_uint, ctypes.c_void_p]
        out = np.zeros(ch * frames * 3 + 8, np.uint8)
        n = self.lib.wav_oracle_pack(planar.ctypes.data, frames, ch, frames, bits, out.ctypes.data)
        return out[:n].tobytes()


class Reference:
    """The real reference decoder; exists only where oracle/_ref has been built."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self):
        self.lib = ctypes.CDLL(REF_SO)
        self.lib.ref_mlp_decode.restype = ctypes.c_long
        self.lib.ref_mlp_decode.argtypes = ([ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t] +
                                            [ctypes.c_uint] * 6 + [ctypes.c_void_p, ctypes.c_size_t])

    def decode(self, data, assignment, rate_code, bps_code, max_frames, chunk=0):
        data = np.ascontiguousarray(data, np.uint8)
        nch = CHANNELS[assignment]
        cap = int(max_frames) + 16
        out = np.zeros((nch, cap), np.int32)
        r = self.lib.ref_mlp_decode(data.ctypes.data, len(data), chunk, bps_code, bps_code, rate_code,
                                    rate_code, assignment, nch, out.ctypes.data, cap)
        if r < 0:
            raise RuntimeError("reference output capacity too small")
        return out[:, :r].copy(), int(r)
