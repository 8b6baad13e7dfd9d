#!/usr/bin/env python3
"""tools/probe/fused_stamp.py [substreams] -- where the two waves of k_chain_fused spend a turn (DVDA_EXP_STAMP build,
shares of wave time; diagnostic).  Build the variant first: python tools/probe/fused_stamp.py build"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
SO = os.path.join(ROOT, "libdvd-audio_amd", "exp_stamp.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    import importlib.util
    spec = importlib.util.spec_from_file_location("b", os.path.join(ROOT, "libdvd-audio_amd", "_build.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    m.build_hip(force=True, defines=["DVDA_EXP_STAMP=1"], out=SO)
    sys.exit(0)
os.environ["DVDA_MLP_HIP_LIB"] = SO
import numpy as np, torch
import libdvd_audio_amd as pkg
syn, hip = pkg.synth, pkg.hipdec
SS = int(sys.argv[1]) if len(sys.argv) > 1 else 1
feats = syn.SF["CHAINED"] | (int(sys.argv[2], 0) if len(sys.argv) > 2 else 0)
n = 4096
cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=512, n_substreams=SS, profile=1, features=feats)
flat, offs, sizes, frames = syn.batch(cfg, 1, n)
dev = torch.device("cuda", 0)
d_bytes = torch.from_numpy(flat).to(dev)
d_off = torch.from_numpy(offs.astype(np.int64)).to(dev); d_len = torch.from_numpy(sizes.astype(np.int64)).to(dev)
out_off = np.zeros(n, np.int64); out_off[1:] = np.cumsum(frames[:-1].astype(np.int64) * 6)
d_oo = torch.from_numpy(out_off).to(dev); d_st = torch.from_numpy(frames.astype(np.int64)).to(dev)
d_pcm = torch.empty(int(frames.sum()) * 6, dtype=torch.int32, device=dev)
ctx = hip.Context(0, n, n * 66, lanes_per_segment=0, layout=1)
hip.lib().dvda_mlp_hip_debug_counters2.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
out = (ctypes.c_ulonglong * 16)()
for it in range(2):
    ctx.index(d_bytes.data_ptr(), len(flat) - 64, d_off.data_ptr(), d_len.data_ptr(), n, 0)
    ctx.decode(d_pcm.data_ptr(), d_oo.data_ptr(), d_st.data_ptr(), 0)
    torch.cuda.synchronize()
    hip.lib().dvda_mlp_hip_debug_counters2(ctx._h, out)
for role, names in (("filter wave", ["ring read (per unit: stamp to stamp)", "recursion (filter_unit)", "publish + next loads", "barrier",
                                     "between units (segment end / set-up)", "loop top", "count written", "record load"]),
                    ("output wave", ["barrier", "read tile + record", "rematrix", "order + store + loop", "-", "-", "-", "-"])):
    v = np.array(list(out)[(8 if role == "output wave" else 0):][:8], dtype=np.float64)
    print(role)
    for nme, x in zip(names, v):
        if x:
            print("  %-40s %6.2f %%  (%.3g cycles)" % (nme, 100 * x / max(v.sum(), 1), x))
