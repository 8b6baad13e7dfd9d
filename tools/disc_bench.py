#!/usr/bin/env python3
"""tools/disc_bench.py [--tracks N] [--aus A] -- end-to-end extraction of a synthetic disc:
build/dvda2wav_hip (disc tier, everything on the GPU but file I/O) beside the reference's own
dvda2wav where oracle/_ref holds it.  Prints one JSON line with wall-clock seconds and
Msamples/s for each, and checks that the files are identical.  Diagnostic, not bench.py."""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libdvd_audio_amd as pkg  # noqa: E402


def digest(d):
    out = {}
    for f in sorted(os.listdir(d)):
        h = hashlib.sha256()
        with open(os.path.join(d, f), "rb") as fh:
            for blk in iter(lambda: fh.read(1 << 22), b""):
                h.update(blk)
        out[f] = h.hexdigest()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tracks", type=int, default=8)
    ap.add_argument("--aus", type=int, default=16384)
    ap.add_argument("--tmp", default=None)
    ap.add_argument("--no-tier-b", action="store_true")
    ap.add_argument("--devices", default="", help="passed to dvda2wav_hip: comma-separated device entries, one worker "
                                                   "thread each (e.g. 0,0,0: three workers on one GPU)")
    ap.add_argument("--pcm-sectors", type=int, default=0,
                    help="instead of MLP tracks: ONE raw-PCM track (6 ch / 96 kHz / 24 bit) of this many sectors, read in "
                         "windows of bounded memory when longer than one (1 050 000 sectors = 2.15 GB of AOB)")
    ap.add_argument("--chained", action="store_true",
                    help="tracks as an encoder writes them: no raw lead-in at the restart points, the FIR history runs "
                         "through each track (the chain passes decode them)")
    a = ap.parse_args()
    syn, disc = pkg.synth, pkg.disc
    tool = pkg._build.build_tool()
    ref = os.path.join(ROOT, "oracle", "_ref", "dvda2wav_ref")
    with tempfile.TemporaryDirectory(dir=a.tmp) as tmp:
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=a.aus, profile=1 if a.chained else 0,
                           features=syn.SF["CHAINED"] if a.chained else 0)
        tracks, samples = [], 0
        if a.pcm_sectors:
            import numpy as np
            per = (2048 - 14 - 6 - 7 - 9) // 36 * 2
            rng = np.random.RandomState(5)
            # (a block of random sectors, repeated: the extractors do not care, and 10^6 sectors are not built one by one)
            blk = 4096
            secs = disc.pcm_track_sectors(rng.randint(-(1 << 23), 1 << 23, size=(per * blk, 6)), 2, 1, 12)
            secs = (secs * ((a.pcm_sectors + blk - 1) // blk))[:a.pcm_sectors]
            tracks.append({"sectors": secs, "pcm_frames": per * a.pcm_sectors, "rate_code": 1})
            samples = per * a.pcm_sectors * 6
            a.tracks = 0
            a.no_tier_b = True
        for t in range(a.tracks):
            b, f = syn.stream(cfg, 100 + t)
            tracks.append({"sectors": disc.mlp_track_sectors(b), "pcm_frames": f, "rate_code": 1})
            samples += f * 6
        ats = disc.write_disc_titles(tmp, [tracks])
        aob = os.path.getsize(os.path.join(ats, "ATS_01_1.AOB"))
        res = {"tracks": a.tracks, "samples": samples, "aob_bytes": aob, "chained": bool(a.chained), "devices": a.devices or "0"}
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "libdvd-audio_amd") + ":/opt/rocm/lib:" +
                   os.environ.get("LD_LIBRARY_PATH", ""))
        outs = {}
        for name, exe in (("gpu", tool), ("reference", ref)):
            if not os.path.exists(exe):
                continue
            out = os.path.join(tmp, name)
            os.makedirs(out)
            for rep in range(2 if name == "gpu" else 1):          # second GPU run: page cache + driver warm
                t0 = time.time()
                cmd = [exe, "-A", ats, "-d", out] + (["--devices", a.devices] if a.devices and name == "gpu" else [])
                r = subprocess.run(cmd, capture_output=True, text=True, env=env)
                dt = time.time() - t0
                assert r.returncode == 0, r.stderr[-2000:]
            if name == "gpu":
                import re
                pk = re.findall(r"host peak ([0-9.]+) MB, device peak ([0-9.]+) MB, payload ([0-9.]+) MB", r.stdout)
                if pk:
                    res["windowed_tracks"] = len(pk)
                    res["host_peak_mb_max"] = max(float(x[0]) for x in pk)
                    res["device_peak_mb_max"] = max(float(x[1]) for x in pk)
                    res["payload_mb_max"] = max(float(x[2]) for x in pk)
            res[name + "_seconds"] = round(dt, 3)
            res[name + "_msamples_per_s"] = round(samples / dt / 1e6, 1)
            outs[name] = digest(out)
        if len(outs) == 2:
            res["identical_files"] = outs["gpu"] == outs["reference"]
        # tier B, "dvda2wav links unchanged": the reference's own dvda2wav + dvd-audio.c with only src/mlp.c
        # replaced by integration/mlp_hip_shim.c -- one small GPU batch and synchronous copies per <= 2 KB PES
        # payload.  Compatibility, not throughput: first track only, stated plainly.
        shim = os.path.join(ROOT, "oracle", "_ref", "dvda2wav_hip")
        if os.path.exists(shim) and not a.no_tier_b:
            out = os.path.join(tmp, "tier_b")
            os.makedirs(out)
            t0 = time.time()
            r = subprocess.run([shim, "-A", ats, "-T", "1", "-t", "1", "-d", out], capture_output=True, text=True, env=env)
            dt = time.time() - t0
            assert r.returncode == 0, r.stderr[-2000:]
            one = tracks[0]["pcm_frames"] * 6
            res["tier_b_first_track_seconds"] = round(dt, 3)
            res["tier_b_msamples_per_s"] = round(one / dt / 1e6, 2)
            if "reference" in outs:
                res["tier_b_identical_to_reference"] = digest(out) == {k: v for k, v in outs["reference"].items()
                                                                        if k in os.listdir(out)}
        print(json.dumps(res))


if __name__ == "__main__":
    main()
