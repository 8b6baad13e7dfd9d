"""ctypes stub over libdvd_audio_hip.so (include/dvd-audio-hip.h): the disc-level API.

Reads like a program written against the reference's public header (reference
include/dvd-audio.h:59-201): open the disc, a title set, a title, a track, a track reader, then
dvda_read() until it returns 0.  Opening a track reader is where the GPU batch runs; there is no
CPU decode path behind it.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DISC_SO = os.path.join(HERE, "libdvd_audio_hip.so")

EXPORTS = [
    "dvda_open", "dvda_close", "dvda_titleset_count",
    "dvda_open_titleset", "dvda_close_titleset", "dvda_titleset_number", "dvda_title_count",
    "dvda_open_title", "dvda_close_title", "dvda_title_number", "dvda_track_count", "dvda_title_pts_length",
    "dvda_open_track", "dvda_close_track", "dvda_track_number", "dvda_track_pts_index",
    "dvda_track_pts_length", "dvda_track_first_sector", "dvda_track_last_sector",
    "dvda_open_track_reader", "dvda_close_track_reader", "dvda_codec", "dvda_bits_per_sample",
    "dvda_sample_rate", "dvda_channel_count", "dvda_riff_wave_channel_mask", "dvda_read",
    "dvda_hip_set_device", "dvda_hip_set_wav_output", "dvda_hip_reader_status", "dvda_hip_reader_total_frames",
    "dvda_hip_reader_wav_payload", "dvda_hip_open_track_reader_on", "dvda_hip_reader_wav_only",
    "dvda_hip_reader_wav_next", "dvda_hip_reader_windowed", "dvda_hip_reader_memory", "dvda_hip_reader_failed",
    "dvda_hip_release_cached_buffers",
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        from . import _build, hipdec
        _build.build_all()
        hipdec.lib()                      # torch first, then libdvda_mlp_hip.so (one HIP runtime per process)
        L = ctypes.CDLL(DISC_SO)
        vp, u, cp = ctypes.c_void_p, ctypes.c_uint, ctypes.c_char_p
        for name in ("dvda_open",):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [cp, cp]
        for name in ("dvda_open_titleset", "dvda_open_title", "dvda_open_track"):
            getattr(L, name).restype = vp
            getattr(L, name).argtypes = [vp, u]
        L.dvda_open_track_reader.restype = vp
        L.dvda_open_track_reader.argtypes = [vp]
        for name in ("dvda_close", "dvda_close_titleset", "dvda_close_title", "dvda_close_track",
                     "dvda_close_track_reader"):
            getattr(L, name).restype = None
            getattr(L, name).argtypes = [vp]
        for name in ("dvda_titleset_count", "dvda_titleset_number", "dvda_title_count", "dvda_title_number",
                     "dvda_track_count", "dvda_title_pts_length", "dvda_track_number", "dvda_track_pts_index",
                     "dvda_track_pts_length", "dvda_track_first_sector", "dvda_track_last_sector",
                     "dvda_bits_per_sample", "dvda_sample_rate", "dvda_channel_count",
                     "dvda_riff_wave_channel_mask", "dvda_hip_reader_status"):
            getattr(L, name).restype = u
            getattr(L, name).argtypes = [vp]
        L.dvda_codec.restype = ctypes.c_int
        L.dvda_codec.argtypes = [vp]
        L.dvda_read.restype = u
        L.dvda_read.argtypes = [vp, u, ctypes.POINTER(ctypes.c_int)]
        L.dvda_hip_set_device.restype = None
        L.dvda_hip_set_device.argtypes = [ctypes.c_int]
        L.dvda_hip_set_wav_output.restype = None
        L.dvda_hip_set_wav_output.argtypes = [ctypes.c_int]
        L.dvda_hip_open_track_reader_on.restype = ctypes.c_void_p
        L.dvda_hip_open_track_reader_on.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.dvda_hip_reader_wav_only.restype = ctypes.c_int
        L.dvda_hip_reader_wav_only.argtypes = [ctypes.c_void_p]
        L.dvda_hip_reader_total_frames.restype = ctypes.c_ulonglong
        L.dvda_hip_reader_total_frames.argtypes = [vp]
        L.dvda_hip_reader_wav_payload.restype = ctypes.c_ulonglong
        L.dvda_hip_reader_wav_payload.argtypes = [vp, ctypes.POINTER(ctypes.POINTER(ctypes.c_ubyte))]
        L.dvda_hip_reader_wav_next.restype = ctypes.c_ulonglong
        L.dvda_hip_reader_wav_next.argtypes = [vp, ctypes.POINTER(ctypes.POINTER(ctypes.c_ubyte))]
        L.dvda_hip_reader_windowed.restype = ctypes.c_int
        L.dvda_hip_reader_windowed.argtypes = [vp]
        L.dvda_hip_reader_failed.restype = ctypes.c_int
        L.dvda_hip_reader_failed.argtypes = [vp]
        L.dvda_hip_reader_memory.restype = ctypes.c_int
        L.dvda_hip_reader_memory.argtypes = [vp, ctypes.POINTER(ctypes.c_ulonglong), ctypes.POINTER(ctypes.c_ulonglong)]
        _lib = L
    return _lib


def layout(audio_ts, titleset=1):
    """[(title, track, pts_index, pts_length, first_sector, last_sector), ...] of one title set;
    host only (no GPU needed)."""
    L = lib()
    d = L.dvda_open(audio_ts.encode(), None)
    if not d:
        raise IOError("not an AUDIO_TS directory: %s" % audio_ts)
    out = []
    try:
        ts = L.dvda_open_titleset(d, titleset)
        if not ts:
            raise IOError("title set %d not found" % titleset)
        for ti in range(1, L.dvda_title_count(ts) + 1):
            t = L.dvda_open_title(ts, ti)
            for ki in range(1, L.dvda_track_count(t) + 1):
                k = L.dvda_open_track(t, ki)
                out.append((ti, ki, L.dvda_track_pts_index(k), L.dvda_track_pts_length(k),
                            L.dvda_track_first_sector(k), L.dvda_track_last_sector(k)))
                L.dvda_close_track(k)
            L.dvda_close_title(t)
        L.dvda_close_titleset(ts)
    finally:
        L.dvda_close(d)
    return out


def read_track(audio_ts, titleset, title, track, chunk=4096, wav=False, device=0, fused=False, pieces=False):
    """Decodes one track on the GPU.  Returns a dict: codec ("PCM"/"MLP"), bits, rate, channels,
    mask, status, and pcm = int32 [frames, channels] (interleaved, RIFF-WAVE order) read with
    dvda_read() in `chunk`-frame calls -- or, with wav=True, payload = the WAV data bytes packed
    on the GPU; fused=True (with wav=True) opens the reader as a WAV-payload-only one (dvda_hip_open_track_reader_on):
    MLP tracks are decoded straight into that payload, no int32 PCM and no packing pass.  Device and output form are
    the READER's: no process-wide switch is left behind.  pieces=True (with wav=True) takes the payload piece by piece
    (dvda_hip_reader_wav_next: a long track's windows).  A track read in windows (info["windowed"]) reports the peaks of
    what it held in info["host_peak"] / info["device_peak"]; info["frames"] is then the count at the END of the read."""
    L = lib()
    d = L.dvda_open(audio_ts.encode(), None)
    if not d:
        raise IOError("not an AUDIO_TS directory: %s" % audio_ts)
    ts = t = k = r = None
    try:
        ts = L.dvda_open_titleset(d, titleset)
        t = L.dvda_open_title(ts, title) if ts else None
        k = L.dvda_open_track(t, track) if t else None
        r = L.dvda_hip_open_track_reader_on(k, device, 1 if (fused and wav) else 0) if k else None
        if not r:
            raise RuntimeError("track %d/%d/%d cannot be opened for reading" % (titleset, title, track))
        ch = L.dvda_channel_count(r)
        info = {"codec": "MLP" if L.dvda_codec(r) == 1 else "PCM", "bits": L.dvda_bits_per_sample(r),
                "rate": L.dvda_sample_rate(r), "channels": ch, "mask": L.dvda_riff_wave_channel_mask(r),
                "status": L.dvda_hip_reader_status(r), "frames": int(L.dvda_hip_reader_total_frames(r)),
                "wav_only": bool(L.dvda_hip_reader_wav_only(r))}
        info["windowed"] = bool(L.dvda_hip_reader_windowed(r))
        if wav and pieces:
            got, sizes = [], []
            while True:
                p = ctypes.POINTER(ctypes.c_ubyte)()
                n = L.dvda_hip_reader_wav_next(r, ctypes.byref(p))
                if n == 0:
                    break
                got.append(bytes(ctypes.string_at(p, n)))
                sizes.append(int(n))
            info["payload"] = b"".join(got)
            info["piece_sizes"] = sizes
        elif wav:
            p = ctypes.POINTER(ctypes.c_ubyte)()
            n = L.dvda_hip_reader_wav_payload(r, ctypes.byref(p))
            info["payload"] = bytes(ctypes.string_at(p, n)) if n else b""
        else:
            parts = []
            buf = (ctypes.c_int * (chunk * ch))()
            while True:
                n = L.dvda_read(r, chunk, buf)
                if n == 0:
                    break
                parts.append(np.frombuffer(buf, dtype=np.int32, count=n * ch).reshape(n, ch).copy())
            info["pcm"] = np.concatenate(parts) if parts else np.zeros((0, ch), np.int32)
        if info["windowed"]:
            hp, dp = ctypes.c_ulonglong(), ctypes.c_ulonglong()
            L.dvda_hip_reader_memory(r, ctypes.byref(hp), ctypes.byref(dp))
            info.update(host_peak=int(hp.value), device_peak=int(dp.value), frames=int(L.dvda_hip_reader_total_frames(r)),
                        failed=bool(L.dvda_hip_reader_failed(r)), status=L.dvda_hip_reader_status(r))
        return info
    finally:
        if r:
            L.dvda_close_track_reader(r)
        if k:
            L.dvda_close_track(k)
        if t:
            L.dvda_close_title(t)
        if ts:
            L.dvda_close_titleset(ts)
        L.dvda_close(d)
