"""tools/probe_stream.py -- the mlp.h mirror (tier B) fed garbage and odd packet sequences, each case in its own process:
no crash, no hang; prints frames returned, status, queued bytes.  Diagnostic."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = ["garbage_then_stream", "empty_packets", "one_byte_packets", "huge_packet", "flip_mid", "garbage_only", "wrong_params", "sync_in_garbage"]
if len(sys.argv) == 1:
    for c in CASES:
        try:
            r = subprocess.run([sys.executable, __file__, c], capture_output=True, text=True, timeout=240)
            tail = [l for l in r.stdout.splitlines() if l.strip()][-1:] or [""]
            print("%-24s rc=%d %s" % (c, r.returncode, tail[0][:160]), flush=True)
        except subprocess.TimeoutExpired:
            print("%-24s TIMEOUT" % c, flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import numpy as np
import libdvd_audio_amd as pkg
from tests import oracle_lib
syn, hip = pkg.synth, pkg.hipdec
case = sys.argv[1]
cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=40)
b, f = syn.stream(cfg, 11)
rng = np.random.RandomState(9)
def feed(dec, packets):
    samples = [[] for _ in range(6)]
    rets = []
    for p in packets:
        rets.append(dec.decode_packet(p, samples))
    return samples, rets
def chunks(x, n):
    return [x[i:i + n] for i in range(0, len(x), n)]
d = hip.MLPDecoder(2, 2, 1, 1, 12)
if case == "garbage_then_stream":
    g = rng.randint(0, 256, 3000).astype(np.uint8)
    s, r = feed(d, chunks(g, 700) + chunks(b, 2013))
elif case == "empty_packets":
    s, r = feed(d, [b[:0], b[:0]] + chunks(b, 2013) + [b[:0]])
elif case == "one_byte_packets":
    s, r = feed(d, chunks(b[:1500], 1) + chunks(b[1500:], 2013))
elif case == "huge_packet":
    s, r = feed(d, [np.concatenate([b] * 1)])
elif case == "flip_mid":
    x = b.copy(); x[len(x) // 2] ^= 0x40
    s, r = feed(d, chunks(x, 2013))
elif case == "garbage_only":
    g = rng.randint(0, 256, 20000).astype(np.uint8)
    s, r = feed(d, chunks(g, 2013))
elif case == "wrong_params":
    d.close(); d = hip.MLPDecoder(0, 0, 0, 0, 1)
    s, r = feed(d, chunks(b, 2013))
elif case == "sync_in_garbage":
    g = rng.randint(0, 256, 6000).astype(np.uint8)
    g[1000:1004] = [0xF8, 0x72, 0x6F, 0xBB]; g[996:1000] = [0xF0, 0x10, 0, 0]
    s, r = feed(d, chunks(g, 2013) + chunks(b, 2013))
want = None
print(case, "frames", sum(r), "of", f, "status", hex(d.status), "queued", d.queued_bytes)
d.close()
