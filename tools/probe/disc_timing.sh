#!/bin/bash
# tools/probe/disc_timing.sh -- where build/dvda2wav_hip spends its time on a synthetic 2-track disc (DVDA_DISC_TIMING). Diagnostic.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"
python - <<'PY'
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import libdvd_audio_amd as pkg
syn, disc = pkg.synth, pkg.disc
tool = pkg._build.build_tool()
with tempfile.TemporaryDirectory() as tmp:
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=65536)
    tracks = []
    for t in range(2):
        b, f = syn.stream(cfg, 100 + t)
        tracks.append({"sectors": disc.mlp_track_sectors(b), "pcm_frames": f, "rate_code": 1})
    ats = disc.write_disc_titles(tmp, [tracks])
    env = dict(os.environ, DVDA_DISC_TIMING="1")
    for rep in range(2):
        out = os.path.join(tmp, "o%d" % rep); os.makedirs(out)
        t0 = time.time()
        r = subprocess.run([tool, "-A", ats, "-d", out], capture_output=True, text=True, env=env)
        print("run %d: %.3f s wall" % (rep, time.time() - t0))
    print(r.stderr[-1500:])
    t0 = time.time(); subprocess.run([tool, "-h"], capture_output=True); print("tool -h: %.3f s" % (time.time() - t0))
PY
