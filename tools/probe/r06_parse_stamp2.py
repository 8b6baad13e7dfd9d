#!/usr/bin/env python3
"""tools/probe/r06_parse_stamp2.py -- where the two-substream lane's waves spend their cycles in the chain PARSE pass
(DVDA_EXP_STAMP build: tools/ab_build.py stamp=DVDA_EXP_STAMP): chained and disc-profile batches, one and two substreams,
with the header phase's own split.  Diagnostic."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["DVDA_MLP_HIP_LIB"] = os.path.join(ROOT, "libdvd-audio_amd", "exp_stamp.so")
import numpy as np, torch
import libdvd_audio_amd as pkg
syn, hip = pkg.synth, pkg.hipdec
dev = torch.device("cuda", 0)
L = hip.lib(); L.dvda_mlp_hip_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
SF = syn.SF
names = ["header phase", "prefetch issue / sync fill", "parse(+filter) row", "exchange+rematrix+stage", "ring commit+flush",
         "loop top", "wait for the chunk", "ring top-up test"]
hnames = ["frame header + substream info", "restart header", "parameters up to the channels", "channels' parameters",
          "behind the parse", "(of these: inside synchronous fills)"]
for S in (1, 2):
    for name, kw in (("chained", dict(profile=1, features=SF["CHAINED"])),
                     ("disc profile", dict(profile=1, features=SF["CHAINED"] | SF["DISC"] | SF["FIRRAND"] | SF["MIXBOOKS"]))):
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=512, n_substreams=S, **kw)
        n = 2048
        flat, offs, sizes, frames = syn.batch(cfg, 1, n)
        d_bytes = torch.from_numpy(flat).to(dev)
        d_off = torch.from_numpy(offs.astype(np.int64)).to(dev); d_len = torch.from_numpy(sizes.astype(np.int64)).to(dev)
        out_off = np.zeros(n, np.int64); out_off[1:] = np.cumsum(frames[:-1].astype(np.int64) * 6)
        d_oo = torch.from_numpy(out_off).to(dev); d_st = torch.from_numpy(frames.astype(np.int64)).to(dev)
        d_pcm = torch.empty(int(frames.sum()) * 6, dtype=torch.int32, device=dev)
        ctx = hip.Context(0, n, n * 66, lanes_per_segment=0)
        for it in range(2):
            ctx.index(d_bytes.data_ptr(), len(flat) - 64, d_off.data_ptr(), d_len.data_ptr(), n, 0)
            ctx.decode(d_pcm.data_ptr(), d_oo.data_ptr(), d_st.data_ptr(), 0)
            torch.cuda.synchronize()
            out = (ctypes.c_ulonglong * 16)()
            L.dvda_mlp_hip_debug_counters(ctx._h, out)
        v = np.array(list(out)[:8], dtype=np.float64)
        h = np.array(list(out)[8:14], dtype=np.float64)
        print("%s, %d substream(s): %.3g cycles of waves, %d bytes in" % (name, S, v.sum(), int(sizes.sum())))
        for nme, x in zip(names, v):
            print("  %-32s %6.2f %%  (%.3g)" % (nme, 100 * x / max(v.sum(), 1), x))
        for nme, x in zip(hnames, h):
            print("    %-30s %6.2f %%  (%.3g)" % (nme, 100 * x / max(v.sum(), 1), x))
        print("    header phases (lane 0's): %d, synchronous fills in them: %d" % (out[15], out[14]))
        ctx.close()
        del d_bytes, d_pcm
