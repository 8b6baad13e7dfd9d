#!/usr/bin/env python3
"""tools/fuzz_ifo.py -- CPU-only robustness run of the disc tier's IFO walk (csrc/dvda_disc.c) under
AddressSanitizer + UBSan: truncated, bit-flipped and padded AUDIO_TS.IFO / ATS_01_0.IFO files, every
title / track that still opens is walked.  Build and run:

    gcc -O1 -g -fPIC -shared -fsanitize=address,undefined -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include \
        -o /tmp/asan/libdvd_audio_hip.so libdvd-audio_amd/csrc/dvda_disc.c -Llibdvd-audio_amd \
        -ldvda_mlp_hip -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/libdvd-audio_amd -Wl,-rpath,/opt/rocm/lib
    LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
        ASAN_OPTIONS=detect_leaks=0 python tools/fuzz_ifo.py

(GPU sanitizers are not available on the pool; this covers the host-side parser only.)"""
import ctypes, os, sys, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libdvd_audio_amd as pkg
syn, disc = pkg.synth, pkg.disc
L = ctypes.CDLL("/tmp/asan/libdvd_audio_hip.so")
vp, u, cp = ctypes.c_void_p, ctypes.c_uint, ctypes.c_char_p
L.dvda_open.restype = vp; L.dvda_open.argtypes = [cp, cp]
for n in ("dvda_open_titleset", "dvda_open_title", "dvda_open_track"):
    getattr(L, n).restype = vp; getattr(L, n).argtypes = [vp, u]
for n in ("dvda_close", "dvda_close_titleset", "dvda_close_title", "dvda_close_track"):
    getattr(L, n).restype = None; getattr(L, n).argtypes = [vp]
for n in ("dvda_titleset_count", "dvda_title_count", "dvda_track_count", "dvda_track_first_sector", "dvda_track_last_sector"):
    getattr(L, n).restype = u; getattr(L, n).argtypes = [vp]
rng = np.random.RandomState(3)
b, f = syn.stream(syn.make_cfg(assignment=1, rate_code=0, n_aus=40), 1)
secs = disc.mlp_track_sectors(b)
tmp = tempfile.mkdtemp()
ats = disc.write_disc_titles(tmp, [disc.split_tracks(secs, [3], [f // 2, f - f // 2], 0), disc.split_tracks(secs, [], [f], 0)])
good = {n: open(os.path.join(ats, n), "rb").read() for n in ("AUDIO_TS.IFO", "ATS_01_0.IFO")}
walked = 0
for it in range(3000):
    for n, data in good.items():
        d = bytearray(data)
        mode = it % 4
        if mode == 0:
            d = d[:rng.randint(0, len(d))]
        elif mode == 1:
            for _ in range(rng.randint(1, 8)):
                d[rng.randint(0, len(d))] = rng.randint(0, 256)
        elif mode == 2:
            lo = 2048 if n.startswith("ATS") else 0
            for _ in range(rng.randint(1, 6)):
                d[lo + rng.randint(0, 400)] = rng.randint(0, 256)
        else:
            d = d + bytes(rng.randint(0, 256, size=rng.randint(0, 64)).astype(np.uint8))
        open(os.path.join(ats, n), "wb").write(bytes(d))
    h = L.dvda_open(ats.encode(), None)
    if not h:
        continue
    for ts_n in range(0, min(L.dvda_titleset_count(h), 3) + 1):
        ts = L.dvda_open_titleset(h, ts_n)
        if not ts:
            continue
        for ti in range(0, min(L.dvda_title_count(ts), 300) + 2):
            t = L.dvda_open_title(ts, ti)
            if not t:
                continue
            for ki in range(0, min(L.dvda_track_count(t), 260) + 2):
                k = L.dvda_open_track(t, ki)
                if k:
                    L.dvda_track_first_sector(k); L.dvda_track_last_sector(k)
                    walked += 1
                    L.dvda_close_track(k)
            L.dvda_close_title(t)
        L.dvda_close_titleset(ts)
    L.dvda_close(h)
shutil.rmtree(tmp)
print("fuzzed IFO walks done, tracks opened:", walked)
