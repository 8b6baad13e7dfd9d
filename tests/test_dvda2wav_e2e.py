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


NATIVE_TOOL = os.path.join(ROOT, "oracle", "_ref", "dvda2wav_native")


def _mixed_disc(pkg, tmp):
    """Title 1: ONE 6-ch MLP stream cut into three tracks at sector boundaries (a track's last
    frames sit in the next track's first sector: the end-of-track rule); title 2: a 2-substream
    stream in two tracks; title 3: 16-bit stereo PCM in two tracks; title 4: 24-bit 6-ch PCM."""
    syn, disc = pkg.synth, pkg.disc
    titles = []
    b, f = syn.stream(syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=96), 3)
    secs = disc.mlp_track_sectors(b)
    titles.append(disc.split_tracks(secs, [len(secs) // 3, 2 * len(secs) // 3 + 1], [f // 3, f // 3, f - 2 * (f // 3)], 1))
    b, f = syn.stream(syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=64), 4)
    secs = disc.mlp_track_sectors(b)
    titles.append(disc.split_tracks(secs, [len(secs) // 2], [f // 2, f - f // 2], 1))
    rng = np.random.RandomState(11)
    pcm = rng.randint(-32768, 32768, size=(3000, 2))
    secs = disc.pcm_track_sectors(pcm, 0, 0, 1)
    per = 2008 // 4
    titles.append(disc.split_tracks(secs, [3], [3 * per, len(pcm) - 3 * per], 0))
    pcm6 = rng.randint(-(1 << 23), 1 << 23, size=(1000, 6))
    titles.append(disc.split_tracks(disc.pcm_track_sectors(pcm6, 2, 1, 12), [], [len(pcm6)], 1))
    return disc.write_disc_titles(tmp, titles)


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(REF_TOOL) and os.path.exists(NATIVE_TOOL)),
                    reason="reference tools not built (dev container only)")
def test_disc_library_replaces_libdvd_audio_under_reference_dvda2wav(pkg):
    """SURVEY 8(b) outer boundary, rows f-1/f-4: the reference's utils/dvda2wav.c linked against
    libdvd_audio_hip.so (IFO walk, sector demux, end-of-track rule and decode all ours, on the
    GPU) writes the same files as the all-reference build."""
    with tempfile.TemporaryDirectory() as tmp:
        ats = _mixed_disc(pkg, tmp)
        ref = _run(REF_TOOL, ats, os.path.join(tmp, "ref"))
        env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "libdvd-audio_amd") + ":/opt/rocm/lib:" +
                   os.environ.get("LD_LIBRARY_PATH", ""))
        out = os.path.join(tmp, "native")
        os.makedirs(out)
        r = subprocess.run([NATIVE_TOOL, "-A", ats, "-d", out], capture_output=True, text=True, timeout=900, env=env)
        assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
        got = sorted(os.path.join(out, f) for f in os.listdir(out))
        assert [os.path.basename(p) for p in got] == [os.path.basename(p) for p in ref]
        assert len(ref) == 8
        for a, b in zip(ref, got):
            assert open(a, "rb").read() == open(b, "rb").read(), os.path.basename(a)
        # single title / single track selection takes the same path
        one = os.path.join(tmp, "one")
        os.makedirs(one)
        r = subprocess.run([NATIVE_TOOL, "-A", ats, "-T", "1", "-t", "2", "-d", one], capture_output=True,
                           text=True, timeout=900, env=env)
        assert r.returncode == 0 and os.listdir(one) == ["track-01-02.wav"]
        assert open(os.path.join(one, "track-01-02.wav"), "rb").read() == open(ref[1], "rb").read()


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_TOOL), reason="reference tools not built (dev container only)")
def test_gpu_extractor_writes_the_reference_files(pkg):
    """tools/dvda2wav_hip.c: the whole chain on the GPU (IFO walk, demux, decode, write_signed packing),
    only the header and fwrite() on the host -- same files as the reference's dvda2wav."""
    tool = pkg._build.build_tool()
    with tempfile.TemporaryDirectory() as tmp:
        ats = _mixed_disc(pkg, tmp)
        ref = _run(REF_TOOL, ats, os.path.join(tmp, "ref"))
        got = _run(tool, ats, os.path.join(tmp, "gpu"))
        assert [os.path.basename(p) for p in got] == [os.path.basename(p) for p in ref] and len(ref) == 8
        for a, b in zip(ref, got):
            assert open(a, "rb").read() == open(b, "rb").read(), os.path.basename(a)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_TOOL), reason="reference tools not built (dev container only)")
def test_gpu_extractor_fans_tracks_out_over_device_entries(pkg):
    """dvda2wav_hip --devices 0,0,0: three worker threads, each with its own device entry (all the one GPU of a test
    box), take the disc's tracks in turn -- the same eight files as the reference's tool, whichever worker wrote which."""
    tool = pkg._build.build_tool()
    with tempfile.TemporaryDirectory() as tmp:
        ats = _mixed_disc(pkg, tmp)
        ref = _run(REF_TOOL, ats, os.path.join(tmp, "ref"))
        out = os.path.join(tmp, "gpu")
        os.makedirs(out)
        r = subprocess.run([tool, "-A", ats, "-d", out, "--devices", "0,0,0"], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr
        got = sorted(os.path.join(out, f) for f in os.listdir(out))
        assert [os.path.basename(p) for p in got] == [os.path.basename(p) for p in ref] and len(ref) == 8
        for a, b in zip(ref, got):
            assert open(a, "rb").read() == open(b, "rb").read(), os.path.basename(a)
