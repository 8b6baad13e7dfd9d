#!/usr/bin/env python3
"""tools/coverage_run.py -- which rarely taken paths of k_decode do the parity cases really execute?
Builds the DVDA_EXP_COUNT variant (tools/ab_build.py count=DVDA_EXP_COUNT), decodes the soak's
random configurations plus the special cases of tests/test_gpu_parity.py with it, checks each
against the oracle and prints the per-path execution counts.  Diagnostic."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DVDA_MLP_HIP_LIB"] = os.path.join(ROOT, "libdvd-audio_amd", "exp_count.so")
import torch  # noqa: E402
import libdvd_audio_amd as pkg  # noqa: E402
from tests import oracle_lib  # noqa: E402

NAMES = {8: "two-dword window step", 9: "matrix 2..5 (workspace)", 10: "bypassed LSBs of matrices 2..5",
         11: "IIR taps", 13: "unaligned output (scalar stores)",
         14: "synchronous ring top-up in the row loop"}
oracle = oracle_lib.Oracle()
syn, hip = pkg.synth, pkg.hipdec
L = hip.lib()
L.dvda_mlp_hip_debug_counters.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
tot = np.zeros(16, np.uint64)
dev = torch.device("cuda", 0)


def run(b, f, asg, lanes, misalign=0):
    nch = syn.channels(asg)
    flat, offs, lens = hip.pack_streams([b])
    ctx = hip.Context(0, 1, 1024, lanes)
    d_bytes = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
    ctx.index(d_bytes.data_ptr(), len(flat) - 64, d_off.data_ptr(), d_len.data_ptr(), 1, 0)
    stride = f + 64 + misalign
    d_pcm = torch.zeros(stride * nch + 8, dtype=torch.int32, device=dev)
    d_oo = torch.tensor([misalign], dtype=torch.int64, device=dev)
    d_st = torch.tensor([stride], dtype=torch.int64, device=dev)
    ctx.decode(d_pcm.data_ptr(), d_oo.data_ptr(), d_st.data_ptr(), 0)
    inf = ctx.stream_info()[0]
    out = (ctypes.c_ulonglong * 16)()
    L.dvda_mlp_hip_debug_counters(ctx._h, out)
    tot[:] += np.array(list(out), np.uint64)
    want, r, st = oracle.decode(b, nch, f)
    got = d_pcm.cpu().numpy()[misalign:misalign + stride * nch].reshape(nch, stride)[:, :inf.pcm_frames]
    ok = st == 0 and (inf.status & ~hip.ST_BENIGN) == 0 and got.shape == want.shape and np.array_equal(got, want)
    ctx.close()
    return ok


rng = np.random.RandomState(4242)
bad = 0
for i in range(300):
    S = 1 + (i & 1)
    asg = [12, 1, 0x14, 6, 9, 3, 17, 20][i % 8] if S == 2 else int(rng.randint(0, 21))
    feats = int(rng.randint(0, 1 << 18)) if i % 3 else syn.SF_ALL
    kw = {}
    if i % 7 == 0:
        feats &= ~syn.SF["MIXBOOKS"]
        kw = dict(codebook=1 + i % 3, huffman_lsbs=24)
    cfg = syn.make_cfg(assignment=asg, rate_code=int(rng.randint(0, 3)), n_substreams=S, n_aus=int(rng.randint(4, 30)),
                       profile=1, features=feats, restart_interval=int(rng.randint(1, 9)), **kw)
    b, f = syn.stream(cfg, 50000 + i)
    lanes = 2 if S == 2 or i % 4 == 0 else 1
    bad += 0 if run(b, f, asg, lanes, misalign=(i % 5 == 0) * 1) else 1
print("cases with a mismatch:", bad)
for k in sorted(NAMES):
    print("%-52s %d" % (NAMES[k], tot[k]))
sys.exit(1 if bad else 0)
