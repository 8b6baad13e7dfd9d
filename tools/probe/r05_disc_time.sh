#!/bin/bash
# tools/probe/r05_disc_time.sh -- where the 8-track disc's wall-clock time goes: the tool's own timing lines, 1..4 workers.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$ROOT"; mkdir -p gpurun_out/r05
for d in 0 0,0 0,0,0 0,0,0,0; do
  echo "== devices $d"
  python tools/disc_bench.py --no-tier-b --devices $d 2>&1 | tail -1
done
python - <<'PY'
import os, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
import libdvd_audio_amd as pkg
syn, disc = pkg.synth, pkg.disc
tool = pkg._build.build_tool()
with tempfile.TemporaryDirectory() as tmp:
    ch = bool(os.environ.get("CHAINED"))
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=int(os.environ.get("AUS", "16384")), profile=1 if ch else 0, features=syn.SF["CHAINED"] if ch else 0)
    tracks = []
    for t in range(8):
        b, f = syn.stream(cfg, 100 + t)
        tracks.append({"sectors": disc.mlp_track_sectors(b), "pcm_frames": f, "rate_code": 1})
    ats = disc.write_disc_titles(tmp, [tracks])
    env = dict(os.environ, DVDA_TOOL_TIMING="1", LD_LIBRARY_PATH="libdvd-audio_amd:/opt/rocm/lib")
    for dev in ("0", "0,0,0"):
        out = os.path.join(tmp, "o" + dev.replace(",", ""))
        os.makedirs(out)
        for rep in range(2):
            t0 = time.time()
            r = subprocess.run([tool, "-A", ats, "-d", out, "--devices", dev], capture_output=True, text=True, env=env)
            dt = time.time() - t0
        print("devices", dev, "wall %.3f s, started at %.1f ms" % (dt, (t0 % 1e6) * 1e3))
        print(r.stderr)
PY
