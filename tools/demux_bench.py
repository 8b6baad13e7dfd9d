#!/usr/bin/env python3
"""Bandwidth of the MLP track demux on one GPU: AOB sectors resident in HBM ->
contiguous MLP payload (sector walk + scan + gather).  Prints one JSON line; bytes = sectors read +
payload written."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import libdvd_audio_amd as pkg
hip, disc, syn = pkg.hipdec, pkg.disc, pkg.synth
b, f = syn.stream(syn.make_cfg(assignment=12, rate_code=1, n_aus=4096), 7)
unit = np.frombuffer(b"".join(disc.mlp_track_sectors(b)), np.uint8).copy()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda", 0)
d_sec = torch.from_numpy(unit).to(dev).repeat(reps)
n = d_sec.numel() // 2048
d_out = torch.empty(d_sec.numel() + 64, dtype=torch.uint8, device=dev)
d_work = torch.zeros(int(hip.lib().dvda_pcm_hip_workspace_words(n)), dtype=torch.int32, device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
def step():
    hip._check(hip.lib().dvda_mlp_hip_demux_sectors(d_sec.data_ptr(), n, d_out.data_ptr(), d_sec.numel(),
                                                    d_work.data_ptr(), st), "demux")
for _ in range(2):
    step()
torch.cuda.synchronize()
K = 10
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
total, bad = ctypes.c_uint64(), ctypes.c_uint32()
hip.lib().dvda_pcm_hip_result(d_work.data_ptr(), n, ctypes.byref(total), ctypes.byref(bad), st)
ok = bad.value == 0 and total.value == len(b) * reps and bool(np.array_equal(d_out[:len(b)].cpu().numpy(), b))
nbytes = n * 2048 + total.value
print(json.dumps({"metric": "MLP track demux", "sectors": n, "ms": round(dt * 1e3, 3), "GB_per_s": round(nbytes / dt / 1e9, 1),
                  "frac_of_8TBs": round(nbytes / dt / 8e12, 3), "payload_correct": ok}))
