#!/usr/bin/env python3
"""Diagnostic: runs one bench-sized decode with the DVDA_EXP_STAMP build and prints where a
wave's loop iteration spends its cycles (shares, not absolute times)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DVDA_MLP_HIP_LIB"] = os.path.join(ROOT, "libdvd-audio_amd", os.environ.get("STAMP_LIB", "exp_stamp.so"))
import numpy as np, torch
import libdvd_audio_amd as pkg
syn, hip = pkg.synth, pkg.hipdec
SS = 1
if len(sys.argv) > 1 and sys.argv[1] in ("one", "c4"):
    # small batches (the cooperative kernel): ONE title of 512 units / configs[3]'s 1 024 single units
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=512 if sys.argv[1] == "one" else 1)
    n = 1 if sys.argv[1] == "one" else 1024
    flat, offs, sizes, frames = syn.batch(cfg, 1, n)
    nchs = np.full(n, 6, np.int64)
    nseg = 1024
elif len(sys.argv) > 1 and sys.argv[1] in ("fuzz_fast", "fuzz_all"):
    # the bench's fuzz sub-record shape (8 configurations x 512 titles of 64 access units)
    from bench import gen_mixed
    SF = syn.SF
    feats = syn.SF_FAST & ~(SF["IIR"] | SF["MATRIXRAND"]) if sys.argv[1] == "fuzz_fast" else syn.SF_FAST
    shapes = [(12, 1, 8), (1, 1, 5), (12, 2, 16), (0x12, 0, 3), (12, 0, 8), (6, 1, 4), (0, 2, 8), (12, 1, 2)]
    specs = [(syn.make_cfg(assignment=a, rate_code=rc, n_substreams=1, n_aus=64, profile=1, features=feats,
                           restart_interval=ri), 512) for a, rc, ri in shapes]
    flat, offs, sizes, frames, nchs, nseg = gen_mixed(syn, specs, 90000)
    n = len(sizes)
else:
    SS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048       # titles: 2 048 = two waves per SIMD, 512 = half of the SIMDs one
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=512, n_substreams=SS)
    flat, offs, sizes, frames = syn.batch(cfg, 1, n)
    nchs = np.full(n, 6, np.int64)
    nseg = n * 64
dev = torch.device("cuda", 0)
d_bytes = torch.from_numpy(flat).to(dev)
d_off = torch.from_numpy(offs.astype(np.int64)).to(dev); d_len = torch.from_numpy(sizes.astype(np.int64)).to(dev)
out_off = np.zeros(n, np.int64); out_off[1:] = np.cumsum(frames[:-1].astype(np.int64) * nchs[:-1])
d_oo = torch.from_numpy(out_off).to(dev); d_st = torch.from_numpy(frames.astype(np.int64)).to(dev)
d_pcm = torch.empty(int((frames.astype(np.int64) * nchs).sum()), dtype=torch.int32, device=dev)
ctx = hip.Context(0, n, nseg, lanes_per_segment=0)
for it in range(2):
    ctx.index(d_bytes.data_ptr(), len(flat) - 64, d_off.data_ptr(), d_len.data_ptr(), n, 0)
    ctx.decode(d_pcm.data_ptr(), d_oo.data_ptr(), d_st.data_ptr(), 0)
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    hip.lib().dvda_mlp_hip_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    hip.lib().dvda_mlp_hip_debug_counters(ctx._h, out)
names = ["header phase", "prefetch issue / sync fill", "parse+filter (row)", "exchange+rematrix+stage", "ring commit+flush",
         "loop top", "wait for the chunk", "ring top-up test"]
for role in range(1):       # (rounds 2-5: two roles for the two-wave kernel; since round 6 one lane reads both substreams)
    v = np.array(list(out)[8 * role:8 * role + 8], dtype=np.float64)
    print("substreams per title:", SS)
    for nme, x in zip(names, v):
        print("  %-28s %6.2f %%  (%.3g cycles)" % (nme, 100 * x / max(v.sum(), 1), x))
v = np.array(list(out)[8:16], dtype=np.float64)
if SS == 1 and v.sum() > 0 and n > 1024:
    t = np.array(list(out)[0:8], dtype=np.float64).sum()
    print("inside the header phase (lane 0's clock; shares of the wave's whole time):")
    for nme, x in zip(["frame header + substream info", "restart header", "parameters up to the channels", "channels' parameters",
                       "behind the parse"], v):
        print("  %-32s %6.2f %%  (%.3g cycles)" % (nme, 100 * x / max(t, 1), x))
elif v.sum() > 0:
    print("cooperative kernel (k_coop), shares of its waves' time:")
    for nme, x in zip(["staging + framing", "block headers", "symbol scan", "residuals", "filter", "meeting + rematrix + output",
                       "loop frame", "-"], v):
        print("  %-28s %6.2f %%  (%.3g cycles)" % (nme, 100 * x / max(v.sum(), 1), x))
print(ctx.kernel_time())
