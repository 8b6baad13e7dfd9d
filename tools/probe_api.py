"""tools/probe_api.py -- batch-tier calls a careful caller would not make (capacities too small, empty and one-byte
streams, odd lengths, garbage between streams ...), each in its own process (a GPU memory fault kills the process):
per case the return code, the streams' status and PCM-frame counts, and whether the words behind the PCM buffer are
untouched.  Diagnostic."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["short_capacity", "zero_len_streams", "tiny_total", "one_byte_stream", "odd_len", "wav_small_capacity",
         "two_substreams_short_capacity", "garbage_between", "many_empty", "chained_short_capacity"]
if len(sys.argv) == 1:
    for c in CASES:
        r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=300)
        tail = [l for l in r.stdout.splitlines() if l.strip()][-1:] or [""]
        print("%-32s rc=%d %s" % (c, r.returncode, tail[0][:150]), flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np, torch
import libdvd_audio_amd as pkg
syn, hip = pkg.synth, pkg.hipdec
case = sys.argv[1]
dev = torch.device("cuda", 0)
def run(streams_bytes, offs, lens, total, nch, cap_frames, layout=hip.PCM_PLANAR, words_per_frame=None):
    n = len(offs)
    flat = streams_bytes
    d_bytes = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(np.asarray(offs, np.int64)).to(dev); d_len = torch.from_numpy(np.asarray(lens, np.int64)).to(dev)
    wpf = words_per_frame or nch
    oo = torch.from_numpy(np.arange(n, dtype=np.int64) * (cap_frames * wpf + 4)).to(dev)
    stride = torch.from_numpy(np.full(n, cap_frames, np.int64)).to(dev)
    guard = 4096
    pcm = torch.full((n * (cap_frames * wpf + 4) + guard,), 0x5A5A5A5A, dtype=torch.int32, device=dev)
    ctx = hip.Context(0, n, 4096, layout=layout)
    ctx.index(d_bytes.data_ptr(), total, d_off.data_ptr(), d_len.data_ptr(), n, 0)
    ctx.decode(pcm.data_ptr(), oo.data_ptr(), stride.data_ptr(), 0)
    infos = ctx.stream_info(n)
    tail_ok = bool((pcm[-guard:] == 0x5A5A5A5A).all().item())
    print([(hex(i.status), int(i.pcm_frames)) for i in infos][:6], "guard intact" if tail_ok else "GUARD OVERWRITTEN")
    ctx.close()
cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=24)
b, f = syn.stream(cfg, 5)
if case == "short_capacity":
    flat, offs, lens = hip.pack_streams([b, b]); run(flat, offs, lens, len(flat) - 64, 6, f // 3)
elif case == "zero_len_streams":
    flat, offs, lens = hip.pack_streams([b, b, b]); run(flat, [offs[0], offs[1], offs[1], offs[2]], [lens[0], 0, lens[1], lens[2]], len(flat) - 64, 6, f)
elif case == "tiny_total":
    flat = np.zeros(10 + 64, np.uint8); run(flat, [0], [10], 10, 6, 64)
elif case == "one_byte_stream":
    flat, offs, lens = hip.pack_streams([b[:1], b, b[:3]]); run(flat, offs, lens, len(flat) - 64, 6, f)
elif case == "odd_len":
    flat, offs, lens = hip.pack_streams([b[:len(b) - 1], b[:4001]]); run(flat, offs, lens, len(flat) - 64, 6, f)
elif case == "wav_small_capacity":
    flat, offs, lens = hip.pack_streams([b, b]); run(flat, offs, lens, len(flat) - 64, 6, f // 2, layout=hip.PCM_WAV24, words_per_frame=5)
elif case == "two_substreams_short_capacity":
    c2 = syn.make_cfg(assignment=12, rate_code=1, n_aus=24, n_substreams=2); b2, f2 = syn.stream(c2, 6)
    flat, offs, lens = hip.pack_streams([b2, b, b2]); run(flat, offs, lens, len(flat) - 64, 6, f2 // 3)
elif case == "garbage_between":
    rng = np.random.RandomState(3); g = rng.randint(0, 256, 5000).astype(np.uint8)
    g[100:104] = [0xF8, 0x72, 0x6F, 0xBB]
    flat, offs, lens = hip.pack_streams([b, g, b]); run(flat, [offs[0], offs[2]], [lens[0], lens[2]], len(flat) - 64, 6, f)
elif case == "many_empty":
    flat, offs, lens = hip.pack_streams([b] + [b[:0]] * 50 + [b]); run(flat, offs, lens, len(flat) - 64, 6, f)
elif case == "chained_short_capacity":
    cc = syn.make_cfg(assignment=12, rate_code=1, n_aus=48, profile=1, features=syn.SF["CHAINED"]); bc, fc = syn.stream(cc, 8)
    flat, offs, lens = hip.pack_streams([bc, bc]); run(flat, offs, lens, len(flat) - 64, 6, fc // 2)
