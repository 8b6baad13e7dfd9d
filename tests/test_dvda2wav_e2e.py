"""End to end through the REFERENCE's own dvda2wav (BASELINE configs[0] and [1]).

oracle/_ref/dvda2wav_ref = the reference utility, all reference sources.
oracle/_ref/dvda2wav_hip = the same reference sources with src/mlp.c replaced by our binding
                           (integration/mlp_hip_shim.c -> libdvda_mlp_hip.so).
Both are built in the dev container by `make -C oracle ref_tools` and travel to the GPU box as
prebuilt binaries; the tests skip where they are absent."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TOOL = os.path.join(ROOT, "oracle", "_ref", "dvda2wav_ref")
HIP_TOOL = os.path.join(ROOT, "oracle", "_ref", "dvda2wav_hip")


def _wav_pcm(path, nch, bps):
    data = open(path, "rb").read()
    assert data[:4] == b"RIFF" and data[8:12] == b"WAVE"
    nb = bps // 8
    pay = np.frombuffer(data[68:], np.uint8).reshape(-1, nb).astype(np.int64)   # 68-byte EXTENSIBLE header
    v = sum(pay[:, i] << (8 * i) for i in range(nb))
    v = np.where(v & (1 << (bps - 1)), v - (1 << bps), v)
    return v.reshape(-1, nch).T.astype(np.int32)


def _run(tool, ats, out):
    os.makedirs(out, exist_ok=True)
    r = subprocess.run([tool, "-A", ats, "-d", out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return sorted(os.path.join(out, f) for f in os.listdir(out))


def _mlp_disc(pkg, oracle, tmp, layouts):
    syn, disc = pkg.synth, pkg.disc
    tracks, want = [], []
    for i, (asg, rate, S, naus, feat) in enumerate(layouts):
        cfg = syn.make_cfg(assignment=asg, rate_code=rate, n_substreams=S, n_aus=naus,
                           profile=1 if feat else 0, features=feat)
        b, f = syn.stream(cfg, 20 + i)
        pcm, r, st = oracle.decode(b, syn.channels(asg), f)
        assert st == 0 and r == f
        tracks.append({"sectors": disc.mlp_track_sectors(b), "pcm_frames": f, "rate_code": rate})
        want.append(pcm)
    return disc.write_disc(tmp, tracks), want


@pytest.mark.skipif(not os.path.exists(REF_TOOL), reason="reference tools not built (dev container only)")
def test_pcm_disc_through_reference_dvda2wav(pkg):
    """configs[0]: 2-ch / 48 kHz / 16-bit uncompressed PCM AOB, reference CPU path (plumbing)."""
    disc = pkg.disc
    rng = np.random.RandomState(5)
    samples = rng.randint(-32768, 32768, size=(5020, 2))
    with tempfile.TemporaryDirectory() as tmp:
        ats = disc.write_disc(tmp, [{"sectors": disc.pcm_track_sectors(samples, 0, 0, 1),
                                     "pcm_frames": len(samples), "rate_code": 0}])
        wavs = _run(REF_TOOL, ats, os.path.join(tmp, "out"))
        assert len(wavs) == 1
        got = _wav_pcm(wavs[0], 2, 16)
        assert np.array_equal(got, samples.T)


@pytest.mark.skipif(not os.path.exists(REF_TOOL), reason="reference tools not built (dev container only)")
def test_mlp_disc_reference_equals_oracle(pkg, oracle):
    with tempfile.TemporaryDirectory() as tmp:
        ats, want = _mlp_disc(pkg, oracle, tmp, [(1, 1, 1, 64, 0), (12, 1, 2, 48, 0)])
        wavs = _run(REF_TOOL, ats, os.path.join(tmp, "out"))
        assert len(wavs) == 2
        for w, pcm in zip(wavs, want):
            assert np.array_equal(_wav_pcm(w, pcm.shape[0], 24), pcm)


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(REF_TOOL) and os.path.exists(HIP_TOOL)),
                    reason="reference tools not built (dev container only)")
def test_dvda2wav_links_unchanged_and_matches(pkg, oracle):
    """configs[1] and [2] end to end: the reference's dvda2wav + dvd-audio.c, linked against the HIP
    decoder through the mlp.h binding, must write byte-identical WAV files."""
    with tempfile.TemporaryDirectory() as tmp:
        ats, want = _mlp_disc(pkg, oracle, tmp, [(1, 1, 1, 96, 0), (12, 1, 1, 64, 0), (12, 1, 2, 48, 0),
                                                 (12, 1, 1, 40, pkg.synth.SF["CHAINED"] | pkg.synth.SF["FIRRAND"])])
        ref = _run(REF_TOOL, ats, os.path.join(tmp, "ref"))
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "libdvd-audio_amd") + ":/opt/rocm/lib:" +
                   os.environ.get("LD_LIBRARY_PATH", ""))
        out = os.path.join(tmp, "hip")
        os.makedirs(out)
        r = subprocess.run([HIP_TOOL, "-A", ats, "-d", out], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
        hip = sorted(os.path.join(out, f) for f in os.listdir(out))
        assert len(ref) == len(hip) == 4
        for a, b, pcm in zip(ref, hip, want):
            assert open(a, "rb").read() == open(b, "rb").read(), os.path.basename(a)
            assert np.array_equal(_wav_pcm(b, pcm.shape[0], 24), pcm)
