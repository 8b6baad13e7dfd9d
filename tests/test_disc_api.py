"""Disc-level API (include/dvd-audio-hip.h, libdvd_audio_hip.so): SURVEY 8(b) outer boundary,
rows f-1 (demux + end-of-track rule) and f-4 (IFO walk, several title sets).

CPU part: the library loads without a GPU, exports the reference's 27 entry points, walks the IFO
tables like the reference (track sector ranges per src/dvd-audio.c:426-492) and refuses to open a
track reader without a device.  GPU part: every track read through dvda_read() / the GPU WAV
packer against the oracle."""
import ctypes
import os
import re
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _titles(pkg, seeds=(3, 4)):
    syn, disc = pkg.synth, pkg.disc
    titles, streams = [], []
    b, f = syn.stream(syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=96), seeds[0])
    secs = disc.mlp_track_sectors(b)
    titles.append(disc.split_tracks(secs, [len(secs) // 3, 2 * len(secs) // 3 + 1], [f // 3, f // 3, f - 2 * (f // 3)], 1))
    streams.append((b, f, 12))
    b, f = syn.stream(syn.make_cfg(assignment=1, rate_code=0, n_substreams=1, n_aus=80), seeds[1])
    secs = disc.mlp_track_sectors(b)
    titles.append(disc.split_tracks(secs, [len(secs) // 2], [f // 2, f - f // 2], 0))
    streams.append((b, f, 1))
    return titles, streams


def test_header_and_library_export_the_reference_api(pkg):
    text = open(os.path.join(ROOT, "include", "dvd-audio-hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(dvda_\w+)\s*\(", text)))
    assert len([n for n in declared if not n.startswith("dvda_hip_")]) == 27
    assert set(declared) == set(pkg.discdec.EXPORTS)
    lib = pkg.discdec.lib()
    for name in declared:
        assert hasattr(lib, name), "missing export: " + name


def test_ifo_walk_and_track_sector_ranges(pkg):
    disc = pkg.disc
    with tempfile.TemporaryDirectory() as tmp:
        titles, _ = _titles(pkg)
        ats = disc.write_disc_titles(tmp, titles)
        lay = pkg.discdec.layout(ats)
        assert [(t, k) for t, k, *_ in lay] == [(1, 1), (1, 2), (1, 3), (2, 1), (2, 2)]
        pos = 0
        for (ti, ki, pts_i, pts_l, first, last), trk in zip(lay, [t for tt in titles for t in tt]):
            n = len(trk["sectors"])
            assert (first, last) == (pos, pos + n - 1)          # contiguous tracks: last = next first - 1
            assert pts_l == int(round(trk["pcm_frames"] * 90000.0 / disc.RATES[trk["rate_code"]]))
            pos += n
        # case-insensitive file lookup (src/audio_ts.c:37-73)
        os.rename(os.path.join(ats, "ATS_01_0.IFO"), os.path.join(ats, "ats_01_0.ifo"))
        assert len(pkg.discdec.layout(ats)) == 5
        # not a disc
        with pytest.raises(IOError):
            pkg.discdec.layout(tmp)


def test_no_cpu_fallback_for_track_readers(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with tempfile.TemporaryDirectory() as tmp:
        titles, _ = _titles(pkg)
        ats = pkg.disc.write_disc_titles(tmp, titles)
        with pytest.raises(RuntimeError):
            pkg.discdec.read_track(ats, 1, 1, 1)


@pytest.mark.gpu
def test_tracks_of_a_title_partition_the_stream(pkg, oracle):
    """The tracks of a title are cut out of ONE MLP stream at sector boundaries: each track runs from
    its first major sync to the first major sync behind its last sector, so together they give back
    the whole stream's PCM, access unit for access unit -- also in title set 2."""
    syn, disc = pkg.synth, pkg.disc
    for titleset in (1, 2):
        with tempfile.TemporaryDirectory() as tmp:
            titles, streams = _titles(pkg, seeds=(5 + titleset, 9 + titleset))
            ats = disc.write_disc_titles(tmp, titles, titlesets=2, titleset=titleset)
            for ti, (tracks, (b, f, asg)) in enumerate(zip(titles, streams), 1):
                nch = syn.channels(asg)
                want, r, st = oracle.decode(b, nch, f)
                assert st == 0 and r == f
                got = []
                for ki in range(1, len(tracks) + 1):
                    info = pkg.discdec.read_track(ats, titleset, ti, ki, chunk=1000)
                    assert info["codec"] == "MLP" and info["channels"] == nch and info["bits"] == 24
                    assert info["status"] & ~pkg.hipdec.ST_BENIGN == 0
                    assert info["frames"] == len(info["pcm"]) and info["frames"] % 40 == 0
                    got.append(info["pcm"])
                assert all(len(g) for g in got)
                assert np.array_equal(np.concatenate(got).T, want)


@pytest.mark.gpu
def test_pcm_tracks_and_gpu_wav_payload(pkg, oracle):
    disc = pkg.disc
    rng = np.random.RandomState(21)
    pcm = rng.randint(-(1 << 23), 1 << 23, size=(2004, 6))
    with tempfile.TemporaryDirectory() as tmp:
        secs = disc.pcm_track_sectors(pcm, 2, 1, 12)
        per = (2048 - 14 - 6 - 7 - 9) // 36 * 2
        ats = disc.write_disc_titles(tmp, [disc.split_tracks(secs, [4], [4 * per, len(pcm) - 4 * per], 1)])
        a = pkg.discdec.read_track(ats, 1, 1, 1)
        b = pkg.discdec.read_track(ats, 1, 1, 2)
        assert a["codec"] == "PCM" and a["bits"] == 24 and a["rate"] == 96000 and a["mask"] == 0x3F
        assert np.array_equal(np.concatenate([a["pcm"], b["pcm"]]), pcm)
        w = pkg.discdec.read_track(ats, 1, 1, 1, wav=True)
        assert w["payload"] == oracle.wav_pack(a["pcm"].T, 24)


@pytest.mark.gpu
def test_mlp_tracks_decoded_straight_into_the_wav_payload(pkg, oracle):
    """dvda_hip_open_track_reader_on(track, device, 1): an MLP track reader holds the WAV payload the decode kernels wrote themselves
    (DVDA_PCM_WAV24) -- no int32 PCM, no packing pass.  Byte for byte what the int32 decode + GPU packer give, and what
    the oracle's packing of the oracle's PCM gives; dvda_read() on such a reader returns nothing."""
    syn, disc = pkg.synth, pkg.disc
    with tempfile.TemporaryDirectory() as tmp:
        titles, streams = _titles(pkg, seeds=(31, 32))
        ats = disc.write_disc_titles(tmp, titles)
        for ti, (tracks, (b, f, asg)) in enumerate(zip(titles, streams), 1):
            nch = syn.channels(asg)
            for ki in range(1, len(tracks) + 1):
                plain = pkg.discdec.read_track(ats, 1, ti, ki)
                packed = pkg.discdec.read_track(ats, 1, ti, ki, wav=True)
                fused = pkg.discdec.read_track(ats, 1, ti, ki, wav=True, fused=True)
                assert fused["status"] & ~pkg.hipdec.ST_BENIGN == 0 and fused["frames"] == plain["frames"]
                assert fused["wav_only"] and not packed["wav_only"] and not plain["wav_only"]
                assert fused["payload"] == packed["payload"] == oracle.wav_pack(plain["pcm"].T, plain["bits"])
                assert len(fused["payload"]) == plain["frames"] * nch * plain["bits"] // 8
    # (the output form is the reader's: a reader opened the reference's way right after is an ordinary one)
    with tempfile.TemporaryDirectory() as tmp:
        titles, streams = _titles(pkg, seeds=(33, 34))
        ats = disc.write_disc_titles(tmp, titles)
        again = pkg.discdec.read_track(ats, 1, 1, 1)
        assert not again["wav_only"] and again["pcm"].shape[0] == again["frames"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["plain", "chained", "two_substreams", "syncs_without_restart"])
def test_long_tracks_are_read_in_windows_of_bounded_memory(pkg, oracle, kind):
    """A track of more sectors than a window (DVDA_WINDOW_SECTORS) is read, demultiplexed and decoded window by window
    (csrc/dvda_disc.c, "MLP track, in windows"; reference: src/dvd-audio.c:751-795 streams a track of any length): the
    windows are cut at major syncs, the bytes behind the cut and the FIR history at it (src/mlp.c:297-304: never cleared --
    the chained title needs it at every cut) are all that crosses a cut.  dvda_read() and the payload pieces give the
    oracle's PCM; what the reader holds does not grow with the track.  syncs_without_restart: most major syncs stand in
    front of access units that carry no restart header (src/mlp.c:449-460, 748-753) -- a window is never cut at one
    (win_unit_restarts), it is decoded inside its window with the state of the units before it."""
    syn, disc = pkg.synth, pkg.disc
    feats = dict(plain={}, chained=dict(profile=1, features=syn.SF["CHAINED"] | syn.SF["FIRRAND"]),
                 two_substreams=dict(n_substreams=2),
                 syncs_without_restart=dict(profile=1, features=syn.SF["SYNCONLY"] | syn.SF["FIRRAND"], n_substreams=2,
                                            restart_interval=16))[kind]
    peaks = {}
    old = os.environ.get("DVDA_WINDOW_SECTORS")
    os.environ["DVDA_WINDOW_SECTORS"] = "128"                   # 256 KiB of sectors: a dozen windows and more
    pkg.discdec.lib().dvda_hip_release_cached_buffers()         # (the memory figures below: from a clean start)
    try:
        for n_aus in (1600, 4800):
            kw = dict(assignment=12, rate_code=1, n_substreams=1, n_aus=n_aus)
            kw.update(feats)
            b, f = syn.stream(syn.make_cfg(**kw), 77)
            want, r, st = oracle.decode(b, 6, f)
            assert st == 0 and r == f
            secs = disc.mlp_track_sectors(b)
            assert len(secs) > 4 * 128
            with tempfile.TemporaryDirectory() as tmp:
                # two tracks: the first one long (windows; its end is the end-of-track rule on its LAST window), the
                # second short enough to be one batch -- together the whole stream
                # (a chained title is one track: a second track would start on FIR taps with a fresh decoder, where the
                #  reference reads out of its arrays -- src/mlp.c:1278-1290 -- and this library refuses)
                # (nor is a second track cut out of the title with the sync-only units: it would begin at whatever major sync
                #  its first sector holds, most likely one that restarts nothing -- a fresh decoder has no parameters there)
                if kind in ("chained", "syncs_without_restart"):
                    ats = disc.write_disc_titles(tmp, [disc.split_tracks(secs, [], [f], 1)])
                    a = pkg.discdec.read_track(ats, 1, 1, 1, chunk=3001)
                    tail = np.zeros((0, 6), np.int32)
                else:
                    cutp = len(secs) - 40
                    ats = disc.write_disc_titles(tmp, [disc.split_tracks(secs, [cutp], [f - 80, 80], 1)])
                    a = pkg.discdec.read_track(ats, 1, 1, 1, chunk=3001)
                    z = pkg.discdec.read_track(ats, 1, 1, 2)
                    assert not z["windowed"]
                    tail = z["pcm"]
                assert a["windowed"] and not a["failed"]
                assert a["status"] & ~pkg.hipdec.ST_BENIGN == 0
                assert a["frames"] == len(a["pcm"]) and len(a["pcm"]) + len(tail) == f
                assert np.array_equal(np.concatenate([a["pcm"], tail]).T, want)
                # the payload, piece by piece: int32 windows packed on the host, and windows decoded straight into it
                for fused in (False, True):
                    w = pkg.discdec.read_track(ats, 1, 1, 1, wav=True, fused=fused, pieces=True)
                    assert w["windowed"] and len(w["piece_sizes"]) >= 4 and not w["failed"]
                    assert w["payload"] == oracle.wav_pack(a["pcm"].T, 24)
                whole = pkg.discdec.read_track(ats, 1, 1, 1, wav=True, fused=True)       # the one-piece interface still works
                assert whole["payload"] == oracle.wav_pack(a["pcm"].T, 24)
                peaks[n_aus] = (a["host_peak"], a["device_peak"], len(a["pcm"]) * 6 * 4)
        # three times the track, the same windows: what the reader held did not grow with it, and is a fraction of the PCM
        (h1, d1, p1), (h2, d2, p2) = peaks[1600], peaks[4800]
        assert p2 > 2.5 * p1
        assert h2 <= 1.35 * h1 + (1 << 20) and d2 <= 1.35 * d1 + (8 << 20)
        assert h2 < p2 / 2
    finally:
        if old is None:
            del os.environ["DVDA_WINDOW_SECTORS"]
        else:
            os.environ["DVDA_WINDOW_SECTORS"] = old


@pytest.mark.gpu
@pytest.mark.parametrize("bps_code,assignment", [(2, 12), (0, 1)])
def test_long_raw_pcm_tracks_are_read_in_windows_of_bounded_memory(pkg, oracle, bps_code, assignment):
    """A raw-PCM track of more sectors than a window is read, un-swizzled and handed out window by window (round 6;
    csrc/dvda_disc.c open_pcm_windowed; reference: src/dvd-audio.c:752-795, 1017-1083 streams a track packet by packet,
    src/pcm.c:99-193): sectors decode independently, so nothing crosses a cut but the count of frames delivered.
    dvda_read() gives the samples that went in, the payload pieces their write_signed packing, the track after it
    starts where this one's PTS length is covered, and what the reader holds does not grow with the track."""
    disc = pkg.disc
    ch = disc.CHANNELS[assignment]
    bits = disc.BPS[bps_code]
    peaks = {}
    old = os.environ.get("DVDA_WINDOW_SECTORS")
    os.environ["DVDA_WINDOW_SECTORS"] = "64"
    # (what the reader holds is measured from a clean start: the buffers a closed reader leaves for the next one -- sized
    #  by whatever track the tests before this one read -- count as held)
    pkg.discdec.lib().dvda_hip_release_cached_buffers()
    try:
        for n_sec in (400, 1200):
            rng = np.random.RandomState(100 + n_sec)
            per = (2048 - 14 - 6 - 7 - 9) // (2 * ch * (bits // 8)) * 2          # PCM frames a sector holds
            frames = per * n_sec
            pcm = rng.randint(-(1 << (bits - 1)), 1 << (bits - 1), size=(frames, ch))
            secs = disc.pcm_track_sectors(pcm, bps_code, 1, assignment)
            assert len(secs) == n_sec > 4 * 64
            with tempfile.TemporaryDirectory() as tmp:
                # a long track (windows) and a short one behind it (one batch): together the whole run of sectors
                # (the cut at a multiple of eight sectors: the track's length is then a whole number of PTS ticks -- a
                #  length that is not is rounded, and a track whose rounded length lies past its last sector takes the
                #  next one's first packet as well, here as in the reference)
                cut = n_sec - 24
                ats = disc.write_disc_titles(tmp, [disc.split_tracks(secs, [cut], [cut * per, frames - cut * per], 1)])
                a = pkg.discdec.read_track(ats, 1, 1, 1, chunk=3001)
                z = pkg.discdec.read_track(ats, 1, 1, 2)
                assert a["codec"] == "PCM" and a["bits"] == bits and a["channels"] == ch
                assert a["windowed"] and not a["failed"] and not z["windowed"]
                assert a["frames"] == len(a["pcm"]) == cut * per
                assert np.array_equal(np.concatenate([a["pcm"], z["pcm"]]), pcm)
                for fused in (False, True):
                    w = pkg.discdec.read_track(ats, 1, 1, 1, wav=True, fused=fused, pieces=True)
                    assert w["windowed"] and len(w["piece_sizes"]) >= 4 and not w["failed"]
                    assert w["payload"] == oracle.wav_pack(a["pcm"].T, bits)
                whole = pkg.discdec.read_track(ats, 1, 1, 1, wav=True, fused=True)
                assert whole["payload"] == oracle.wav_pack(a["pcm"].T, bits)
                peaks[n_sec] = (a["host_peak"], a["device_peak"], len(a["pcm"]) * ch * 4)
        (h1, d1, p1), (h2, d2, p2) = peaks[400], peaks[1200]
        assert p2 > 2.5 * p1
        assert h2 <= 1.35 * h1 + (1 << 20) and d2 <= 1.35 * d1 + (8 << 20)
        assert h2 < p2 / 2
    finally:
        if old is None:
            del os.environ["DVDA_WINDOW_SECTORS"]
        else:
            os.environ["DVDA_WINDOW_SECTORS"] = old


@pytest.mark.gpu
@pytest.mark.timeout(180)
@pytest.mark.parametrize("damage", ["flipped_byte", "file_cut_short"])
def test_windowed_reader_stops_at_a_damaged_window_and_says_so(pkg, oracle, damage):
    """A long track whose bytes are damaged half way (a flipped payload byte: the substream's parity / CRC-8 or its
    syntax fails; the AOB file shorter than the IFO says: the sectors end inside the track): the windows in front of the
    damage are handed out and equal the oracle's PCM, the reader then ends, reports the failure
    (dvda_hip_reader_failed) and never hangs -- producer thread and caller both come back.  The reference assert()s
    on the first (src/mlp.c:700-708) and returns short reads on the second (src/dvd-audio.c:1151-1166)."""
    syn, disc = pkg.synth, pkg.disc
    old = os.environ.get("DVDA_WINDOW_SECTORS")
    os.environ["DVDA_WINDOW_SECTORS"] = "128"
    try:
        b, f = syn.stream(syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=2400), 91)
        want, r, st = oracle.decode(b, 6, f)
        assert st == 0
        secs = disc.mlp_track_sectors(b)
        with tempfile.TemporaryDirectory() as tmp:
            ats = disc.write_disc_titles(tmp, [disc.split_tracks(secs, [], [f], 1)])
            aob = os.path.join(ats, "ATS_01_1.AOB")
            size = os.path.getsize(aob)
            mid = (size // 2048 // 2) * 2048
            if damage == "flipped_byte":
                with open(aob, "r+b") as fh:
                    fh.seek(mid + 1200)                  # well inside a sector's MLP payload
                    v = fh.read(1)
                    fh.seek(mid + 1200)
                    fh.write(bytes([v[0] ^ 0x5A]))
            else:
                with open(aob, "r+b") as fh:
                    fh.truncate(mid)
            a = pkg.discdec.read_track(ats, 1, 1, 1, chunk=4096)
            assert a["windowed"] and a["failed"]
            n = len(a["pcm"])
            assert 0 < n < f and a["frames"] == n
            assert np.array_equal(a["pcm"].T, want[:, :n])
            # ... and the payload interface ends the same way
            w = pkg.discdec.read_track(ats, 1, 1, 1, wav=True, fused=True, pieces=True)
            assert w["failed"] and w["payload"] == oracle.wav_pack(want[:, :n], 24)
    finally:
        if old is None:
            del os.environ["DVDA_WINDOW_SECTORS"]
        else:
            os.environ["DVDA_WINDOW_SECTORS"] = old


REF_INFO = os.path.join(ROOT, "oracle", "_ref", "debug_info_ref")
NATIVE_INFO = os.path.join(ROOT, "oracle", "_ref", "debug_info_native")


@pytest.mark.skipif(not (os.path.exists(REF_INFO) and os.path.exists(NATIVE_INFO)),
                    reason="reference tools not built (dev container only)")
def test_reference_debug_info_utility_prints_the_same_table(pkg):
    """The reference's utils/dvda-debug-info.c linked against libdvd_audio_hip.so prints, byte for
    byte, what the all-reference build prints (title / track / PTS / sector table).  No decoder is
    involved, so this runs on the CPU."""
    import subprocess
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "libdvd-audio_amd") + ":/opt/rocm/lib:" +
               os.environ.get("LD_LIBRARY_PATH", ""))
    with tempfile.TemporaryDirectory() as tmp:
        titles, _ = _titles(pkg)
        rng = np.random.RandomState(3)
        pcm = rng.randint(-32768, 32768, size=(3000, 2))
        secs = pkg.disc.pcm_track_sectors(pcm, 0, 0, 1)
        titles.append(pkg.disc.split_tracks(secs, [2, 4], [1004, 1004, 992], 0))
        ats = pkg.disc.write_disc_titles(tmp, titles)
        a = subprocess.run([REF_INFO, "-A", ats], capture_output=True, text=True, timeout=120)
        b = subprocess.run([NATIVE_INFO, "-A", ats], capture_output=True, text=True, timeout=120, env=env)
        assert a.returncode == 0 and b.returncode == 0, (a.stderr, b.stderr)
        assert a.stdout == b.stdout and "Title  Track" in a.stdout and a.stdout.count("\n") >= 10
