#!/usr/bin/env python3
"""tools/fuzz_oracle.py -- the oracle (oracle/mlp_oracle.c) and the stream generator under
AddressSanitizer + UBSan on the CPU: valid random configurations, then the same streams with a bit
flipped, cut short, and pure noise.  The oracle is what every parity claim rests on and the tests
feed it corrupted streams too, so it must stay memory-safe on them.  Build and run:

    mkdir -p /tmp/asan
    gcc -O1 -g -fPIC -shared -fsanitize=address,undefined -o /tmp/asan/libmlp_oracle.so oracle/mlp_oracle.c oracle/pcm_oracle.c
    gcc -O1 -g -fPIC -shared -fsanitize=address,undefined -o /tmp/asan/libmlp_synth.so libdvd-audio_amd/synth/mlp_synth.c -lpthread
    LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
        python tools/fuzz_oracle.py
"""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# swap in the sanitized builds
import libdvd_audio_amd as pkg
from tests import oracle_lib
pkg.synth._lib = None
orig_cdll = ctypes.CDLL
def cdll(path, *a, **k):
    base = os.path.basename(str(path))
    if base in ("libmlp_oracle.so", "libmlp_synth.so"):
        path = "/tmp/asan/" + base
    return orig_cdll(path, *a, **k)
ctypes.CDLL = cdll
syn = pkg.synth
ora = oracle_lib.Oracle()
rng = np.random.RandomState(9)
n = 0
for i in range(2500):
    S = 1 + (i & 1)
    asg = [12, 1, 0x14, 6, 9, 3, 17, 20][i % 8] if S == 2 else int(rng.randint(0, 21))
    feats = int(rng.randint(0, 1 << 18)) if i % 3 else syn.SF_ALL
    cfg = syn.make_cfg(assignment=asg, rate_code=int(rng.randint(0, 3)), n_substreams=S, n_aus=int(rng.randint(2, 30)),
                       profile=1, features=feats, restart_interval=int(rng.randint(1, 9)))
    b, f = syn.stream(cfg, 70000 + i)
    want, r, st = ora.decode(b, syn.channels(asg), f)
    assert st == 0 and r == f, (i, hex(st), r, f)
    # corrupted and truncated copies must not trip the sanitizer either
    c = b.copy(); c[rng.randint(0, len(c))] ^= 1 << rng.randint(0, 8)
    for bad in (c, b[:rng.randint(0, len(b))], rng.randint(0, 256, size=600).astype(np.uint8)):
        try:
            ora.decode(bad, syn.channels(asg), f)
        except RuntimeError:
            pass                      # more PCM frames than the caller's capacity: reported, fine
    n += 1
print("oracle + generator under ASan/UBSan:", n, "configurations clean")
