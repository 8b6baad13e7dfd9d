#!/usr/bin/env python3
"""Bandwidth of the PCM un-swizzle path on one GPU: raw 6-ch / 24-bit AOB
sectors resident in HBM -> planar int32.  Prints one JSON line; bytes = sectors read + PCM written."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import libdvd_audio_amd as pkg
hip, disc = pkg.hipdec, pkg.disc
bps, ch, asg = 24, 6, 12
rng = np.random.RandomState(1)
frames_unit = 110 * 2048                      # 2048 sectors of unique data
s = rng.randint(-(1 << 23), 1 << 23, size=(frames_unit, ch))
unit = np.frombuffer(b"".join(disc.pcm_track_sectors(s, 2, 1, asg)), np.uint8)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
d_unit = torch.from_numpy(unit.copy()).to(dev)
d_sec = d_unit.repeat(reps)
n = d_sec.numel() // 2048
cap = n * 110 + 2
d_pcm = torch.empty(ch * cap, dtype=torch.int32, device=dev)
d_work = torch.zeros(int(hip.lib().dvda_pcm_hip_workspace_words(n)), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
def step():
    hip._check(hip.lib().dvda_pcm_hip_decode_sectors(d_sec.data_ptr(), n, bps, ch, d_pcm.data_ptr(), cap,
                                                     d_work.data_ptr(), st), "pcm")
for _ in range(2):
    step()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
frames, bad = ctypes.c_uint64(), ctypes.c_uint32()
hip.lib().dvda_pcm_hip_result(d_work.data_ptr(), n, ctypes.byref(frames), ctypes.byref(bad), st)
got = d_pcm.view(ch, cap)[:, :frames_unit].cpu().numpy()
ok = bool(np.array_equal(got, s.T)) and bad.value == 0 and frames.value == n * 110
nbytes = n * 2048 + frames.value * ch * 4
print(json.dumps({"metric": "PCM un-swizzle", "sectors": n, "ms_per_step": round(dt * 1e3, 3),
                  "Gsamples_per_s": round(frames.value * ch / dt / 1e9, 1), "GB_per_s": round(nbytes / dt / 1e9, 1),
                  "frac_of_8TBs": round(nbytes / dt / 8e12, 3), "bit_exact": ok}))
