#!/usr/bin/env python3
"""Bandwidth of the WAV packing kernel: planar int32 6-ch -> 24-bit interleaved."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import libdvd_audio_amd as pkg
hip = pkg.hipdec
ch, frames, bits = 6, 64 * 1024 * 1024, 24
dev = torch.device("cuda", 0)
d_pcm = torch.randint(-(1 << 23), 1 << 23, (ch * frames,), dtype=torch.int32, device=dev)
d_out = torch.empty(frames * ch * 3 + 4, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
def step():
    hip._check(hip.lib().dvda_mlp_hip_pack_wav(d_pcm.data_ptr(), frames, ch, frames, bits, d_out.data_ptr(), st), "pack")
for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
nbytes = frames * ch * 7
print(json.dumps({"metric": "WAV pack 6ch/24b", "ms": round(dt * 1e3, 3), "Gsamples_per_s": round(frames * ch / dt / 1e9, 1),
                  "GB_per_s": round(nbytes / dt / 1e9, 1), "frac_of_8TBs": round(nbytes / dt / 8e12, 3)}))
