"""GPU parity: HIP decode path (through the C ABI) vs the CPU oracle, bit-exact."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def _both(hip, streams, **kw):
    """decode_streams in both PCM layouts of the C ABI (planar = the reference's `samples` order,
    interleaved = its dvda_read order): the same values and statuses, whatever the case is."""
    pcm, infos = hip.decode_streams(streams, layout=hip.PCM_PLANAR, **kw)
    pcm_i, infos_i = hip.decode_streams(streams, layout=hip.PCM_INTERLEAVED, **kw)
    assert len(pcm) == len(pcm_i)
    for i, (a, b, x, y) in enumerate(zip(pcm, pcm_i, infos, infos_i)):
        assert (x.status, x.pcm_frames, x.channels) == (y.status, y.pcm_frames, y.channels), \
            "stream %d: planar %#x/%d vs interleaved %#x/%d" % (i, x.status, x.pcm_frames, y.status, y.pcm_frames)
        if not (x.status & ~hip.ST_BENIGN):
            assert a.shape == b.shape and np.array_equal(a, b), "stream %d: layouts differ at %s" % (
                i, np.argwhere(a != b)[:4].tolist() if a.shape == b.shape else (a.shape, b.shape))
    return pcm, infos


def _check(pkg, oracle, cfgs_seeds, lanes=2):
    syn, hip = pkg.synth, pkg.hipdec
    streams, frames, cfgs = [], [], []
    for cfg, seed in cfgs_seeds:
        b, f = syn.stream(cfg, seed)
        streams.append(b)
        frames.append(f)
        cfgs.append(cfg)
    pcm, infos = _both(hip, streams, lanes_per_segment=lanes)
    for i, (b, f, cfg) in enumerate(zip(streams, frames, cfgs)):
        nch = syn.channels(cfg.assignment)
        want, r, st = oracle.decode(b, nch, f)
        assert st == 0 and r == f
        inf = infos[i]
        assert inf.status & ~hip.ST_BENIGN == 0, "stream %d status %#x" % (i, inf.status)
        assert inf.pcm_frames == f
        assert inf.channels == nch
        assert pcm[i].shape == want.shape
        assert np.array_equal(pcm[i], want), "stream %d differs at %s" % (
            i, np.argwhere(pcm[i] != want)[:4].tolist())


def test_recipe_6ch_96k(pkg, oracle):
    cfg = pkg.synth.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=64)
    _check(pkg, oracle, [(cfg, s) for s in range(1, 9)], lanes=1)


def test_recipe_2ch_96k(pkg, oracle):
    cfg = pkg.synth.make_cfg(assignment=1, rate_code=1, n_substreams=1, n_aus=96)
    _check(pkg, oracle, [(cfg, s) for s in range(1, 5)], lanes=1)


def test_recipe_two_substreams(pkg, oracle):
    cfg = pkg.synth.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=48)
    _check(pkg, oracle, [(cfg, s) for s in range(1, 5)], lanes=2)


@pytest.mark.parametrize("rate", [0, 1, 2, 8, 9, 10])          # 48 / 96 / 192 / 44.1 / 88.2 / 176.4 kHz
@pytest.mark.parametrize("assignment,S", [(12, 1), (12, 2), (1, 1), (0, 1), (0x14, 2), (0x12, 1), (6, 2)])
def test_fuzz_fast_features(pkg, oracle, assignment, S, rate):
    syn = pkg.synth
    cases = []
    for seed in range(4):
        # (the bit-depth codes 16 / 20 / 24 bit ride along: the major sync carries them, the decode does not clamp)
        cfg = syn.make_cfg(assignment=assignment, rate_code=rate, n_substreams=S, n_aus=24, profile=1,
                           features=syn.SF_FAST, restart_interval=[8, 3, 16, 5][seed], bps_code=(rate + seed) % 3)
        cases.append((cfg, 100 + seed))
    _check(pkg, oracle, cases, lanes=2)


@pytest.mark.parametrize("lanes", [0, 2, 64])
@pytest.mark.parametrize("S", [1, 2])
def test_disc_profile_streams(pkg, oracle, S, lanes):
    """What an encoder writes (generator feature DISC): chained titles -- no raw lead-in at restart points --, every
    block carries parameters, most channels re-send their FIR taps, mixed code books, fixed block positions; also with
    the two substreams' checkdata_present flags disagreeing (src/mlp.c:545: substream 0's goes for both).  Through the
    library's own kernel choice, the lane kernels and the cooperative kernel."""
    syn = pkg.synth
    SF = syn.SF
    cases = []
    for seed in range(6):
        feats = SF["DISC"] | SF["CHAINED"] | SF["FIRRAND"] | SF["MIXBOOKS"] | (SF["HUFFOFF"] if seed & 1 else 0) | \
            (SF["CHECKQUIRK"] | SF["NOCHECK"] if S == 2 and seed >= 3 else 0)
        cfg = syn.make_cfg(assignment=12 if seed % 3 else 0x14, rate_code=[1, 0, 2, 9, 1, 8][seed], n_substreams=S,
                           n_aus=40, profile=1, features=feats, restart_interval=[8, 4, 16, 8, 5, 8][seed],
                           blocks_per_au=[2, 1, 4, 2, 5, 2][seed])
        cases.append((cfg, 700 + seed))
    _check(pkg, oracle, cases, lanes=lanes)


@pytest.mark.parametrize("lanes", [0, 2, 64])
@pytest.mark.parametrize("S", [1, 2])
def test_major_syncs_without_restart_headers(pkg, oracle, S, lanes):
    """A major sync does not oblige the substreams to restart: the reference checks the sync's parameters against the
    first one's and decodes on with the state it has (src/mlp.c:449-460; decode_block src/mlp.c:748-753 reads a restart
    header only where the block's flag says so).  Generator feature SYNCONLY puts such syncs in front of three access
    units in ten; a segment that starts at one has no parameters of its own, the stream goes to the pass that carries
    them.  Same PCM as the oracle (which the compiled reference agrees with on these streams: tests/test_oracle.py)."""
    syn = pkg.synth
    SF = syn.SF
    cases = []
    for seed in range(6):
        feats = SF["SYNCONLY"] | [0, syn.SF_FAST, SF["CHAINED"] | SF["FIRRAND"], syn.SF_ALL, SF["DISC"] | SF["MIXBOOKS"],
                                  SF["IIR"] | SF["MIDRESTART"]][seed]
        cfg = syn.make_cfg(assignment=12 if seed % 3 else 0x14, rate_code=[1, 0, 2, 9, 1, 8][seed], n_substreams=S,
                           n_aus=40, profile=1, features=feats, restart_interval=[8, 4, 16, 8, 5, 3][seed])
        cases.append((cfg, 900 + seed))
    _check(pkg, oracle, cases, lanes=lanes)


def test_golden_vectors_on_gpu(pkg):
    """Every committed golden vector (PCM produced by the real reference) through the HIP path."""
    import glob
    import os
    hip = pkg.hipdec
    paths = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
    assert len(paths) >= 12
    zs = [np.load(p) for p in paths]
    pcm, infos = _both(hip, [z["mlp"] for z in zs], lanes_per_segment=2)
    for path, z, got, inf in zip(paths, zs, pcm, infos):
        assert inf.status & ~hip.ST_BENIGN == 0, "%s status %#x" % (os.path.basename(path), inf.status)
        assert got.shape == z["pcm"].shape, os.path.basename(path)
        assert np.array_equal(got, z["pcm"]), os.path.basename(path)


@pytest.mark.parametrize("feature", ["CHAINED", "MIDMATRIX", "MIDRESTART", "VARROWS"])
@pytest.mark.parametrize("S", [1, 2])
def test_deferred_features_general_pass(pkg, oracle, feature, S):
    """Chained FIR history, mid-frame matrix changes / restarts, non-standard timing: reported by
    the fast pass, decoded by the general pass, still bit-exact."""
    syn = pkg.synth
    SF = syn.SF
    extra = {"CHAINED": SF["FIRRAND"], "MIDMATRIX": SF["PARAMBLOCKS"] | SF["MATRIXRAND"] | SF["VARBLOCK"] |
             SF["QSS"] | SF["OUTSHIFT"], "MIDRESTART": SF["VARBLOCK"], "VARROWS": SF["VARBLOCK"]}[feature]
    cases = []
    for seed in range(6):
        cfg = syn.make_cfg(assignment=12 if seed % 2 == 0 else 1, rate_code=seed % 3, n_substreams=S,
                           n_aus=24, profile=1, features=SF[feature] | extra,
                           restart_interval=[4, 3, 8][seed % 3])
        cases.append((cfg, 300 + seed))
    _check(pkg, oracle, cases, lanes=2)


@pytest.mark.parametrize("S", [1, 2])
def test_fuzz_all_features(pkg, oracle, S):
    syn = pkg.synth
    cases = []
    for seed in range(12):
        cfg = syn.make_cfg(assignment=[12, 1, 0x14, 6][seed % 4] if S == 2 else [12, 1, 0, 0x12][seed % 4],
                           rate_code=seed % 3, n_substreams=S, n_aus=20, profile=1, features=syn.SF_ALL,
                           restart_interval=[5, 3, 8, 2][seed % 4])
        cases.append((cfg, 500 + seed))
    _check(pkg, oracle, cases, lanes=2)


@pytest.mark.parametrize("chunk", [2013, 777, 5000])
@pytest.mark.parametrize("feature", ["recipe", "chained", "all"])
def test_streaming_tier_mirrors_mlp_h(pkg, oracle, feature, chunk):
    """open / decode_packet / close with PES-payload sized chunks: the PCM and the per-call
    return values must equal what the oracle's (= the reference's) decode_packet returns."""
    import ctypes
    syn, hip = pkg.synth, pkg.hipdec
    if feature == "recipe":
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=40)
    elif feature == "chained":
        cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=40, profile=1,
                           features=syn.SF["CHAINED"] | syn.SF["FIRRAND"], restart_interval=4)
    else:
        cfg = syn.make_cfg(assignment=1, rate_code=0, n_substreams=2, n_aus=40, profile=1,
                           features=syn.SF_ALL, restart_interval=5)
    data, frames = syn.stream(cfg, 77)
    nch = syn.channels(cfg.assignment)
    # the oracle, packet by packet
    ol = oracle.lib
    ol.mlp_oracle_open.restype = ctypes.c_void_p
    ol.mlp_oracle_decode_packet.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    ol.mlp_oracle_decode_packet.restype = ctypes.c_uint
    ol.mlp_oracle_close.argtypes = [ctypes.c_void_p]
    od = ol.mlp_oracle_open(nch)
    dec = hip.MLPDecoder(cfg.bps_code, cfg.bps_code, cfg.rate_code, cfg.rate_code, cfg.assignment)
    samples = [[] for _ in range(nch)]
    try:
        for off in range(0, len(data), chunk):
            piece = np.ascontiguousarray(data[off:off + chunk])
            want_n = ol.mlp_oracle_decode_packet(od, piece.ctypes.data, len(piece))
            got_n = dec.decode_packet(piece, samples)
            assert dec.status & ~hip.ST_BENIGN == 0, hex(dec.status)
            assert got_n == want_n, "packet at %d: %d vs %d" % (off, got_n, want_n)
        path = dec.path
    finally:
        ol.mlp_oracle_close(od)
        dec.close()
    want, r, st = oracle.decode(data, nch, frames)
    assert st == 0
    got = np.asarray(samples, np.int32)
    assert got.shape == want.shape and np.array_equal(got, want)
    # standard streams stay on the path that keeps the decoder state on the device (a call decodes its own units only)
    if feature != "all":
        assert path == 0


def test_streaming_tier_takes_packets_of_any_size(pkg, oracle):
    """A caller may hand decode_packet far more than a PES payload: more access units than one step of the stepping
    kernel takes (48 units / 48 KB) are decoded in several steps and returned as one run per channel; a packet that
    ends inside an access unit leaves the rest queued."""
    syn, hip = pkg.synth, pkg.hipdec
    for S, nch_assign, rate in ((1, 12, 1), (2, 12, 2), (1, 1, 0)):
        cfg = syn.make_cfg(assignment=nch_assign, rate_code=rate, n_substreams=S, n_aus=230, profile=1,
                           features=syn.SF["CHAINED"] | syn.SF["FIRRAND"] | syn.SF["PARAMBLOCKS"], restart_interval=7)
        data, frames = syn.stream(cfg, 5150 + S)
        nch = syn.channels(cfg.assignment)
        want, r, st = oracle.decode(data, nch, frames)
        assert st == 0
        for cuts in ([len(data)], [3, 70001, 70003, len(data) - 1, len(data)], [100000, len(data)]):
            dec = hip.MLPDecoder(cfg.bps_code, cfg.bps_code, cfg.rate_code, cfg.rate_code, cfg.assignment)
            samples = [[] for _ in range(nch)]
            try:
                lo = 0
                for hi in cuts:
                    dec.decode_packet(np.ascontiguousarray(data[lo:hi]), samples)
                    assert dec.status & ~hip.ST_BENIGN == 0, hex(dec.status)
                    lo = hi
                assert dec.path == 0 and dec.queued_bytes < 4
            finally:
                dec.close()
            got = np.asarray(samples, np.int32)
            assert got.shape == want.shape and np.array_equal(got, want), (S, cuts)


@pytest.mark.parametrize("S", [1, 2])
def test_streaming_tier_state_on_the_device_and_its_fall_back(pkg, oracle, S):
    """Tier B keeps the decoder state on the device (k_coop<false, true>): IIR taps, six matrices, parameter and matrix
    changes inside a unit, restarts inside a unit, later major syncs with other parameters all stay on that path.  An
    access unit of non-standard length sends the stream to the batch-tier path for good -- same PCM either way."""
    syn, hip = pkg.synth, pkg.hipdec
    SF = syn.SF
    fast = SF["CHAINED"] | SF["FIRRAND"] | SF["IIR"] | SF["PARAMBLOCKS"] | SF["MATRIXRAND"] | SF["MIDMATRIX"] | \
        SF["QSS"] | SF["OUTSHIFT"] | SF["VARBLOCK"] | SF["MIXBOOKS"] | SF["MIDRESTART"]
    # (SYNCONLY: major syncs in front of units that restart nothing, src/mlp.c:449-460 -- the state on the device decodes
    #  through them; the queue is cut, and the fall-back starts, only at a sync whose substreams all restart)
    for feats, want_path in ((fast, 0), (fast | SF["VARROWS"], 1), (fast | SF["SYNCONLY"], 0),
                             (fast | SF["SYNCONLY"] | SF["VARROWS"], 1)):
        for seed in range(4):
            cfg = syn.make_cfg(assignment=12 if S == 2 or seed % 2 == 0 else 1, rate_code=seed % 3, n_substreams=S, n_aus=36,
                               profile=1, features=feats, restart_interval=[4, 3, 8, 5][seed])
            data, frames = syn.stream(cfg, 4100 + seed)
            nch = syn.channels(cfg.assignment)
            want, r, st = oracle.decode(data, nch, frames)
            assert st == 0
            dec = hip.MLPDecoder(cfg.bps_code, cfg.bps_code, cfg.rate_code, cfg.rate_code, cfg.assignment)
            samples = [[] for _ in range(nch)]
            try:
                for off in range(0, len(data), 2011):
                    dec.decode_packet(np.ascontiguousarray(data[off:off + 2011]), samples)
                    assert dec.status & ~hip.ST_BENIGN == 0, hex(dec.status)
                path = dec.path
            finally:
                dec.close()
            got = np.asarray(samples, np.int32)
            assert got.shape == want.shape and np.array_equal(got, want), (S, seed, want_path)
            if want_path == 0:
                assert path == 0, (S, seed)


def test_edge_cases_truncated_ragged_and_single_unit(pkg, oracle):
    """Ragged batch: titles of very different lengths, a truncated tail (left unconsumed, as the
    reference leaves it queued), a stream of a single access unit, a maximum-size 192 kHz unit."""
    syn, hip = pkg.synth, pkg.hipdec
    cfgs = [syn.make_cfg(assignment=12, rate_code=1, n_aus=1), syn.make_cfg(assignment=1, rate_code=0, n_aus=3),
            syn.make_cfg(assignment=12, rate_code=2, n_aus=17, restart_interval=16),
            syn.make_cfg(assignment=12, rate_code=1, n_aus=333)]
    streams, frames = [], []
    for i, c in enumerate(cfgs):
        b, f = syn.stream(c, 900 + i)
        streams.append(b)
        frames.append(f)
    cut = streams[3][:len(streams[3]) - 123]            # ends inside the last access unit
    pcm, infos = _both(hip, streams[:3] + [cut], lanes_per_segment=2)
    for i in range(3):
        want, r, st = oracle.decode(streams[i], syn.channels(cfgs[i].assignment), frames[i])
        assert infos[i].status & ~hip.ST_BENIGN == 0 and np.array_equal(pcm[i], want)
    want, r, st = oracle.decode(cut, 6, frames[3])
    assert st == 0 and r == frames[3] - 80
    assert infos[3].status & hip.ST["TRUNCATED"] and infos[3].status & ~hip.ST_BENIGN == 0
    assert infos[3].pcm_frames == r and np.array_equal(pcm[3], want)
    assert infos[3].bytes_consumed < len(cut)


def test_corruption_is_reported_not_decoded(pkg, oracle):
    """Where the reference assert()s (parity/CRC, bad restart sync, invalid code), both the oracle
    and the HIP path must flag the stream; untouched streams of the same batch stay exact."""
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=24)
    good, frames = syn.stream(cfg, 41)
    flip = good.copy()
    flip[len(flip) // 2] ^= 0x04                        # payload bit: parity and CRC both break
    crc_only = good.copy()
    # find the first frame's CRC byte: frame 0 = header 4 + sync 28 + info 2 + substream
    size0 = 2 * (((int(good[0]) & 0xF) << 8) | int(good[1]))
    crc_only[size0 - 1] ^= 0xFF
    nosync = good[size0:].copy()                        # starts with a non-sync frame
    pcm, infos = _both(hip, [good, flip, crc_only, nosync, good], lanes_per_segment=1)
    want, r, st = oracle.decode(good, 6, frames)
    for i in (0, 4):
        assert infos[i].status == 0 and np.array_equal(pcm[i], want)
    _, _, st1 = oracle.decode(flip, 6, frames)
    _, _, st2 = oracle.decode(crc_only, 6, frames)
    _, _, st3 = oracle.decode(nosync, 6, frames)
    assert st1 & (hip.ST["PARITY"] | hip.ST["CRC"]) and infos[1].status & (hip.ST["PARITY"] | hip.ST["CRC"])
    assert st2 & hip.ST["CRC"] and infos[2].status & hip.ST["CRC"]
    assert st3 & hip.ST["NO_SYNC"] and infos[3].status & hip.ST["NO_SYNC"]


def test_linearity_free_property_checksum_of_large_batch(pkg, oracle):
    """Full-size style check through a size-independent property: the GPU's per-title sums over a
    larger batch equal the oracle's, and decoding is idempotent (same bytes, same PCM, twice)."""
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=128)
    streams = [syn.stream(cfg, 7000 + i)[0] for i in range(48)]
    pcm1, infos1 = _both(hip, streams, lanes_per_segment=1)
    pcm2, infos2 = _both(hip, streams, lanes_per_segment=2)
    for i, b in enumerate(streams):
        want, r, st = oracle.decode(b, 6, 128 * 80)
        # (forced lane pairs: the two-wave kernel keeps four channels per substream in registers; a six-channel
        #  substream goes through the chain passes and says so -- an information bit, not an error)
        assert st == 0 and infos1[i].status == 0 and infos2[i].status & ~hip.ST_BENIGN == 0
        assert int(pcm1[i].astype(np.int64).sum()) == int(want.astype(np.int64).sum())
        assert np.array_equal(pcm1[i], pcm2[i]) and np.array_equal(pcm1[i], want)


def _inject_false_sync(stream, frame_index):
    """Grows one access unit by 64 ignored tail bytes that look exactly like a major-sync access
    unit header (the reference ignores everything behind the last substream, src/mlp.c:470-610)."""
    b = stream
    pos = 0
    for _ in range(frame_index):
        pos += 2 * (((int(b[pos]) & 0xF) << 8) | int(b[pos + 1]))
    size = 2 * (((int(b[pos]) & 0xF) << 8) | int(b[pos + 1]))
    fake = np.zeros(64, np.uint8)
    fake[0:2] = [0x00, 0x40]                      # size field: 0x40 words = 128 bytes
    fake[4:8] = [0xF8, 0x72, 0x6F, 0xBB]
    fake[8:10] = [0x22, 0x11]
    fake[11] = 12
    fake[20] = 0x10                               # substream count 1
    new_size = size + 64
    out = np.concatenate([b[:pos], b[pos:pos + size], fake, b[pos + size:]]).astype(np.uint8)
    out[pos] = (out[pos] & 0xF0) | ((new_size // 2) >> 8)
    out[pos + 1] = (new_size // 2) & 0xFF
    return out


@pytest.mark.parametrize("S", [1, 2])
def test_false_sync_pattern_in_ignored_bytes_is_resolved(pkg, oracle, S):
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=40)
    clean, frames = syn.stream(cfg, 1234)
    chained_cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=40, profile=1,
                               features=syn.SF["CHAINED"] | syn.SF["FIRRAND"], restart_interval=4)
    chained, cframes = syn.stream(chained_cfg, 4321)
    cases = [(_inject_false_sync(clean, 3), frames), (_inject_false_sync(clean, 39), frames),
             (_inject_false_sync(chained, 10), cframes), (clean, frames)]
    pcm, infos = _both(hip, [c[0] for c in cases], lanes_per_segment=2)
    for (b, f), got, inf in zip(cases, pcm, infos):
        want, r, st = oracle.decode(b, 6, f)
        assert st == 0 and r == f
        assert inf.status & ~hip.ST_BENIGN == 0, hex(inf.status)
        assert inf.pcm_frames == f and np.array_equal(got, want)
    # and through the streaming tier
    dec = hip.MLPDecoder(2, 2, 1, 1, 12)
    samples = [[] for _ in range(6)]
    b = cases[2][0]
    for off in range(0, len(b), 3001):
        dec.decode_packet(np.ascontiguousarray(b[off:off + 3001]), samples)
        assert dec.status & ~hip.ST_BENIGN == 0
    dec.close()
    want, r, st = oracle.decode(b, 6, cframes)
    assert np.array_equal(np.asarray(samples, np.int32), want)


@pytest.mark.gpu
def test_mixed_corpus_192k_mlp_and_raw_pcm_titles_on_concurrent_streams(pkg, oracle):
    """BASELINE configs[4] in small: 6-ch/192 kHz MLP titles and 6-ch/24-bit raw-PCM AOB titles are
    decoded at the same time, each kind from its own host thread on its own HIP stream (every
    C-ABI entry point takes the stream); both must match their oracles."""
    import threading
    import torch
    syn, disc = pkg.synth, pkg.disc
    cfg = syn.make_cfg(assignment=12, rate_code=2, n_aus=96)
    titles = [syn.stream(cfg, 700 + i) for i in range(4)]
    rng = np.random.RandomState(77)
    pcm_titles = [rng.randint(-(1 << 23), 1 << 23, size=(2000 + 200 * i, 6)) for i in range(3)]
    pcm_sectors = [np.frombuffer(b"".join(disc.pcm_track_sectors(p, 2, 2, 12)), np.uint8).copy() for p in pcm_titles]
    got, errors = {}, []

    def run_mlp():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                got["mlp"] = _both(pkg.hipdec, [b for b, _ in titles], lanes_per_segment=1)
        except Exception as e:          # surfaced in the main thread
            errors.append(e)

    def run_pcm():
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                got["pcm"] = [pkg.hipdec.pcm_decode_sectors(s, 24, 6) for s in pcm_sectors]
        except Exception as e:
            errors.append(e)

    ts = [threading.Thread(target=run_mlp), threading.Thread(target=run_pcm)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    pcm, infos = got["mlp"]
    for (b, f), p, inf in zip(titles, pcm, infos):
        want, r, st = oracle.decode(b, 6, f)
        assert st == 0 and r == f == p.shape[1] and f == 96 * 160
        assert inf.status & ~pkg.hipdec.ST_BENIGN == 0 and np.array_equal(p, want)
    for src, (out, bad) in zip(pcm_titles, got["pcm"]):
        assert bad == 0 and np.array_equal(out, src.T)


@pytest.mark.gpu
@pytest.mark.timeout(120)
@pytest.mark.parametrize("S", [1, 2])
def test_garbage_and_bit_flips_never_hang_and_never_pass_silently(pkg, oracle, S):
    """Robustness: random bytes, streams with random bit flips and streams cut at random places go
    through the whole batch path.  The call must return; a stream the oracle decodes cleanly must
    come out identical; a stream the oracle rejects must carry a non-benign status (or stop at the
    same PCM-frame count) -- never clean status with different PCM.  (S = 2: the two substreams of a
    segment run in different waves that meet at a barrier every row -- one of them dying must not
    stall or derail the other.)"""
    syn, hip = pkg.synth, pkg.hipdec
    rng = np.random.RandomState(2024)
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=16)
    base, frames = syn.stream(cfg, 99)
    streams = [rng.randint(0, 256, size=4000).astype(np.uint8)]          # pure noise
    noise_with_sync = rng.randint(0, 256, size=3000).astype(np.uint8)
    noise_with_sync[100:104] = [0xF0, 0x40, 0x00, 0x00]
    noise_with_sync[104:108] = [0xF8, 0x72, 0x6F, 0xBB]                  # a sync pattern in noise
    streams.append(noise_with_sync)
    for i in range(24):
        s = base.copy()
        for _ in range(1 + i % 3):
            pos = rng.randint(0, len(s))
            s[pos] ^= 1 << rng.randint(0, 8)
        streams.append(s)
    for i in range(6):
        streams.append(base[:rng.randint(40, len(base))].copy())          # cut anywhere
    pcm, infos = _both(hip, streams, lanes_per_segment=S)
    clean = 0
    for s, p, inf in zip(streams, pcm, infos):
        want, r, st = oracle.decode(s, 6, frames)
        benign = (inf.status & ~hip.ST_BENIGN) == 0
        if st == 0 and r == 0:
            assert p.shape[1] == 0               # nothing decodable (the batch tier also says NO_SYNC)
        elif st == 0:
            assert benign and p.shape[1] == r and np.array_equal(p, want)
            clean += 1
        elif benign:
            # the oracle stopped at an error the batch tier does not see as fatal: it must at
            # least agree on everything the oracle produced before stopping
            assert p.shape[1] >= r and np.array_equal(p[:, :r], want)
    assert clean >= 4


@pytest.mark.gpu
def test_long_last_access_unit_is_timing_not_overflow(pkg, oracle):
    """Found by tools/soak.py: when the last segment's access unit is longer than the standard 40/80/160
    frames, the fast pass runs past both its standard length and the output capacity at the same row.
    That is a timing matter (the general pass re-places the rows), so the stream must end without
    DVDA_ST_OVERFLOW although the total fits the capacity derived from the standard length."""
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=17, rate_code=2, n_substreams=2, n_aus=15, profile=1, features=syn.SF_ALL,
                       restart_interval=2)
    b, f = syn.stream(cfg, 10303)
    want, r, st = oracle.decode(b, 6, f)
    assert st == 0 and r == f
    pcm, infos = _both(hip, [b], lanes_per_segment=2)
    assert infos[0].status & hip.ST["TIMING"]
    assert infos[0].status & ~hip.ST_BENIGN == 0
    assert np.array_equal(pcm[0], want)


@pytest.mark.gpu
@pytest.mark.parametrize("codebook", [1, 2, 3])
def test_widest_symbols_step_the_bit_window_by_two_dwords(pkg, oracle, codebook):
    """A 9-bit code followed by 24 LSBs starting at bit 31 of the window consumes two whole dwords --
    the row loop's rare second window step (wave-uniform gate).  The generator reaches it with a
    code book and huffman_lsbs = 24 (a DVDA_EXP_COUNT build counted 22 such steps in 24 of these
    streams); every stream must stay bit-exact."""
    syn = pkg.synth
    cases = []
    for seed in range(6):
        cfg = syn.make_cfg(assignment=[12, 1, 9][seed % 3], rate_code=seed % 3, n_substreams=1 + seed % 2, n_aus=24,
                           profile=seed % 2, features=(syn.SF_FAST & ~syn.SF["MIXBOOKS"]) if seed % 2 else 0,
                           restart_interval=1 + seed, codebook=codebook, huffman_lsbs=24)
        cases.append((cfg, 31000 + 10 * codebook + seed))
    _check(pkg, oracle, cases, lanes=2)


@pytest.mark.gpu
def test_device_code_book_decode_equals_the_tables_exhaustively(pkg, oracle):
    """The row loop decodes the three code books arithmetically (no table).  Every (book, 9-bit peek)
    is run through that device function and compared with the oracle's tables (= reference
    src/mlp_codebook{1,2,3}.json); book 0 (no code) must read as value 0, length 0."""
    import ctypes
    L = pkg.hipdec.lib()
    out = (ctypes.c_uint32 * (4 * 512))()
    L.dvda_mlp_hip_selftest_huff.argtypes = [ctypes.c_int, ctypes.c_void_p]
    assert L.dvda_mlp_hip_selftest_huff(0, out) == 0
    got = np.array(list(out), np.uint32).reshape(4, 512)
    assert not got[0].any()
    ol = oracle.lib
    for book in (1, 2, 3):
        for peek in range(512):
            want = ol.mlp_oracle_test_huff(book, peek)
            if (want & 0xFF) == 0xFF:
                assert (int(got[book, peek]) & 0xFF) == 0xFF, (book, peek)
            else:
                assert int(got[book, peek]) == want, (book, peek, hex(int(got[book, peek])), hex(want))


@pytest.mark.parametrize("layout", ["planar", "interleaved"])
@pytest.mark.parametrize("assignment,lanes", [(12, 1), (1, 1), (6, 2), (12, 2)])
def test_output_buffer_not_16_byte_aligned(pkg, oracle, assignment, lanes, layout):
    """d_out_off / the buffer need not be 16-byte aligned: the kernels then leave the 16-byte stores
    for scalar ones (both PCM layouts; 6-, 2- and 5-channel flush paths)."""
    import torch
    syn, hip = pkg.synth, pkg.hipdec
    lay = hip.PCM_INTERLEAVED if layout == "interleaved" else hip.PCM_PLANAR
    dev = torch.device("cuda", 0)
    cfg = syn.make_cfg(assignment=assignment, rate_code=1, n_substreams=lanes if assignment != 12 or lanes == 2 else 1,
                       n_aus=24, profile=1, features=syn.SF_FAST, restart_interval=4)
    b, f = syn.stream(cfg, 4711)
    nch = syn.channels(assignment)
    want, r, st = oracle.decode(b, nch, f)
    assert st == 0
    flat, offs, lens = hip.pack_streams([b])
    d_bytes = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
    for mis in (1, 2, 3):
        ctx = hip.Context(0, 1, 1024, 2 if cfg.n_substreams == 2 else lanes, lay)
        try:
            ctx.index(d_bytes.data_ptr(), len(flat) - 64, d_off.data_ptr(), d_len.data_ptr(), 1, 0)
            stride = f + 8
            d_pcm = torch.full((stride * nch + 16,), 0x5A5A5A5A, dtype=torch.int32, device=dev)
            d_oo = torch.tensor([mis], dtype=torch.int64, device=dev)
            d_st = torch.tensor([stride], dtype=torch.int64, device=dev)
            ctx.decode(d_pcm.data_ptr(), d_oo.data_ptr(), d_st.data_ptr(), 0)
            inf = ctx.stream_info()[0]
            assert inf.status & ~hip.ST_BENIGN == 0 and inf.pcm_frames == f
            host = d_pcm.cpu().numpy()
            body = host[mis:mis + stride * nch]
            got = body.reshape(stride, nch).T[:, :f] if lay == hip.PCM_INTERLEAVED else body.reshape(nch, stride)[:, :f]
            assert np.array_equal(got, want), "misalignment %d" % mis
            assert (host[:mis] == 0x5A5A5A5A).all(), "wrote in front of the buffer"
        finally:
            ctx.close()


@pytest.mark.parametrize("mix", ["alike", "ragged", "mixed_shapes"])
def test_wave_flushes_its_pcm_together(pkg, oracle, monkeypatch, mix):
    """Frame-major int32 output, every lane of a wave flushing in the same turn: the wave stores each other's 96-byte
    runs (csrc/mlp_decode.h, DVDA_COOP_OUT) -- only in batches of more than a wave and a half per SIMD, which no other
    test reaches: forced here for every batch (DVDA_COOP_MIN_SEG=1, read when a context is made).  Titles alike (whole
    waves in lockstep, a part-filled last wave), of different lengths (lanes fall out of a wave one by one) and of
    different shapes (waves whose lanes do not flush together take the per-lane flush)."""
    monkeypatch.setenv("DVDA_COOP_MIN_SEG", "1")
    syn = pkg.synth
    cases = []
    for i in range(44):
        if mix == "alike":
            cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=24, restart_interval=8)
        elif mix == "ragged":
            cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=9 + (i * 7) % 23, restart_interval=4)
        else:
            asg, rc = [(12, 1), (1, 1), (12, 2), (6, 0)][i % 4]
            cfg = syn.make_cfg(assignment=asg, rate_code=rc, n_substreams=1, n_aus=16, profile=1, features=syn.SF_FAST & ~(
                syn.SF["IIR"] | syn.SF["MATRIXRAND"]), restart_interval=4)
        cases.append((cfg, 900 + i))
    _check(pkg, oracle, cases, lanes=1)


def test_wave_flush_into_a_pcm_buffer_of_more_than_16_gb(pkg, oracle, monkeypatch):
    """The cooperative flush hands a lane's output position round as ONE 32-bit element offset while every lane's fits,
    as two words otherwise (wave-uniform): titles placed either side of element 2^32 of one PCM buffer."""
    import torch
    monkeypatch.setenv("DVDA_COOP_MIN_SEG", "1")
    syn, hip = pkg.synth, pkg.hipdec
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 24 << 30:
        pytest.skip("needs a 17 GB PCM buffer")
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=16, restart_interval=8)
    n = 40                                          # 80 segments: one whole wave and a part-filled one
    streams, frames = [], []
    for i in range(n):
        b, f = syn.stream(cfg, 7000 + i)
        streams.append(b)
        frames.append(f)
    flat, offs, lens = hip.pack_streams(streams)
    d_bytes = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
    stride = max(frames)
    # titles 0..19 end just below element 2^32, titles 20..39 start just above it: both kinds of wave, and one that mixes them
    base = (1 << 32) - 20 * stride * 6
    out_off = np.array([base + i * stride * 6 + (64 if i >= 20 else 0) for i in range(n)], np.int64)
    d_pcm = torch.empty(int(out_off[-1]) + stride * 6 + 64, dtype=torch.int32, device=dev)
    ctx = hip.Context(0, n, 4096, 1, hip.PCM_INTERLEAVED)
    try:
        ctx.index(d_bytes.data_ptr(), len(flat) - 64, d_off.data_ptr(), d_len.data_ptr(), n, 0)
        d_oo = torch.from_numpy(out_off).to(dev)
        d_st = torch.full((n,), stride, dtype=torch.int64, device=dev)
        ctx.decode(d_pcm.data_ptr(), d_oo.data_ptr(), d_st.data_ptr(), 0)
        infos = ctx.stream_info()
        for i in range(n):
            want, r, st = oracle.decode(streams[i], 6, frames[i])
            assert st == 0 and infos[i].status & ~hip.ST_BENIGN == 0 and infos[i].pcm_frames == frames[i]
            got = d_pcm[int(out_off[i]):int(out_off[i]) + frames[i] * 6].cpu().numpy().reshape(frames[i], 6).T
            assert np.array_equal(got, want), "title %d" % i
    finally:
        ctx.close()
        del d_pcm
        torch.cuda.empty_cache()


def test_pcm_layout_argument_is_checked(pkg):
    """dvda_mlp_hip_set_pcm_layout: the two layouts are accepted, anything else is DVDA_HIP_EINVAL (-3)
    and leaves the context as it was."""
    hip = pkg.hipdec
    ctx = hip.Context(0, 1, 64, 1)
    try:
        L = hip.lib()
        assert L.dvda_mlp_hip_set_pcm_layout(ctx._h, hip.PCM_INTERLEAVED) == 0
        assert L.dvda_mlp_hip_set_pcm_layout(ctx._h, hip.PCM_WAV16) == 0 and L.dvda_mlp_hip_set_pcm_layout(ctx._h, 4) == -3
        assert L.dvda_mlp_hip_set_pcm_layout(ctx._h, hip.PCM_PLANAR) == 0
        assert L.dvda_mlp_hip_set_pcm_layout(None, hip.PCM_PLANAR) == -3
    finally:
        ctx.close()


@pytest.mark.parametrize("S", [1, 2])
def test_config3_1024_independent_access_units(pkg, oracle, S):
    """BASELINE configs[3]: a batch of 1 024 independent 6-ch/96 kHz streams of ONE access unit each
    (major sync + restart header + raw lead-in block), different seeds -- every one of them against the
    oracle, both PCM layouts; and the shard of that list over 8 ranks covers it exactly."""
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=1)
    flat, offs, sizes, frames = syn.batch(cfg, 1, 1024)
    streams = [flat[int(o):int(o + s)] for o, s in zip(offs, sizes)]
    assert (frames == 80).all()
    pcm, infos = _both(hip, streams, lanes_per_segment=S)
    for i, b in enumerate(streams):
        want, r, st = oracle.decode(b, 6, 80)
        assert st == 0 and r == 80
        assert infos[i].status == 0 and infos[i].pcm_frames == 80 and infos[i].segments == 1
        assert np.array_equal(pcm[i], want), "unit %d differs" % i
    parts = [pkg.shard.shard_titles(sizes, 8, r) for r in range(8)]
    assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(1024))
    loads = np.array([sizes[p].sum() for p in parts])
    assert loads.max() - loads.min() <= sizes.max()


def test_device_bit_reader_known_answers(pkg, oracle):
    """The kernels' bit reader on the reference's own known answers for the bytes B1 ED 3B C1
    (src/bitstream.c:4864-4868 unsigned, 4940-4944 signed), through the cold-path read()/read_signed() and
    through the row loop's branch-free read_resident(); then against the oracle's reader on random fields
    that cross every dword boundary of a longer buffer."""
    import ctypes
    L = pkg.hipdec.lib()
    L.dvda_mlp_hip_selftest_bits.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p,
                                             ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]

    def dev(data, widths, resident=0):
        b = np.ascontiguousarray(data, np.uint8)
        w = np.asarray(widths, np.int32)
        out = np.zeros(len(w), np.int64)
        assert L.dvda_mlp_hip_selftest_bits(0, b.ctypes.data, len(b), w.ctypes.data, len(w), out.ctypes.data,
                                            resident) == 0
        return out.tolist()

    kat = [0xB1, 0xED, 0x3B, 0xC1]
    assert dev(kat, [2, 3, 5, 3, 19]) == [2, 6, 7, 5, 0x53BC1]
    assert dev(kat, [2, 3, 5, 3, 19], resident=1) == [2, 6, 7, 5, 0x53BC1]
    assert dev(kat, [-2, -3, -5, -3, -19]) == [-2, -2, 7, -3, -181311]
    assert dev(kat, [0, 8, 0, 24]) == [0, 0xB1, 0, 0xED3BC1]
    assert dev(kat, [32]) == [0xB1ED3BC1]
    ol = oracle.lib
    ol.mlp_oracle_test_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int),
                                        ctypes.c_int, ctypes.POINTER(ctypes.c_long)]
    rng = np.random.RandomState(5)
    data = rng.randint(0, 256, size=600).astype(np.uint8)
    for resident in (0, 1):
        widths, total = [], 0
        while total < 4500:
            w = int(rng.randint(0, 32 if resident else 33))
            if not resident and rng.randint(0, 3) == 0 and w:
                w = -w
            widths.append(w)
            total += abs(w)
        cw = (ctypes.c_int * len(widths))(*widths)
        want = (ctypes.c_long * len(widths))()
        assert ol.mlp_oracle_test_read(data.ctypes.data, len(data), cw, len(widths), want) == 0
        assert dev(data, widths, resident) == list(want)


@pytest.mark.parametrize("S", [1, 2])
def test_changed_major_syncs_are_dropped_like_the_reference(pkg, oracle, S):
    """Later major syncs with other stream parameters (reference src/mlp.c:449-460): the index walks through
    them, the decode drops those access units and goes on with the state it has -- batch tier in both
    layouts, next to an untouched stream, and packet by packet through the streaming tier.  (The two
    sync_change_* golden vectors pin the same against PCM produced by the real reference.)"""
    from tests import stream_tools
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=48, restart_interval=4)
    clean, frames = syn.stream(cfg, 606 + S)
    one, _ = stream_tools.change_sync_params(clean, (3,), g1_bps=0)
    runs, _ = stream_tools.change_sync_params(clean, (1, 2, 5, 6, 7, 8, 11), assignment=3)
    pcm, infos = _both(hip, [one, clean, runs])
    for b, got, inf, drops in ((one, pcm[0], infos[0], 1), (clean, pcm[1], infos[1], 0), (runs, pcm[2], infos[2], 7)):
        want, r, st = oracle.decode(b, 6, frames)
        assert r == frames - 80 * drops and st == (2 if drops else 0)
        assert inf.status & ~hip.ST_BENIGN == 0, hex(inf.status)
        assert bool(inf.status & hip.ST["SYNC_CHANGE"]) == bool(drops)
        assert inf.pcm_frames == r and np.array_equal(got, want)
    # ten in a row (round 2 stopped at five): still what the reference does -- ten units dropped, the rest exact
    ten, _ = stream_tools.change_sync_params(clean, tuple(range(1, 11)), g1_bps=0)
    pcm10, inf10 = _both(hip, [ten, clean])
    want, r, st = oracle.decode(ten, 6, frames)
    assert r == frames - 80 * 10 and st == 2
    assert inf10[0].status & ~hip.ST_BENIGN == 0 and inf10[0].pcm_frames == r and np.array_equal(pcm10[0], want)
    # seventy in a row (rounds 2-4 stopped the walk at 64 and reported the stream): the reference drops any number
    # (src/mlp.c:449-460), and so does the index's walk of a candidate with the stream's own parameters (round 5)
    cfg_l = syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=80, restart_interval=1)
    long_clean, frames_l = syn.stream(cfg_l, 616 + S)
    many, _ = stream_tools.change_sync_params(long_clean, tuple(range(2, 72)), g1_bps=0)
    pcmm, infm = _both(hip, [many, long_clean])
    want, r, st = oracle.decode(many, 6, frames_l)
    assert r == frames_l - 80 * 70 and st == 2
    assert infm[0].status & ~hip.ST_BENIGN == 0, hex(infm[0].status)
    assert infm[0].pcm_frames == r and np.array_equal(pcmm[0], want)
    # streaming tier, PES-payload sized packets
    dec = hip.MLPDecoder(2, 2, 1, 1, 12)
    samples = [[] for _ in range(6)]
    for off in range(0, len(runs), 2013):
        dec.decode_packet(np.ascontiguousarray(runs[off:off + 2013]), samples)
        assert dec.status & ~hip.ST_BENIGN == 0, hex(dec.status)
    dec.close()
    want, r, st = oracle.decode(runs, 6, frames)
    assert np.array_equal(np.asarray(samples, np.int32), want)


def test_thousands_of_chains_and_midframe_segments_in_one_batch(pkg, oracle):
    """More deferred work than any fixed pool of frame buffers could hold (round 1 stopped at 2 048 run
    heads with DVDA_ST_CAPACITY): 4 500 chained titles + 700 with mid-frame parameter changes + 300 plain
    ones in one batch, every one of them against the oracle."""
    syn, hip = pkg.synth, pkg.hipdec
    SF = syn.SF
    specs = [(syn.make_cfg(assignment=12, rate_code=1, n_aus=6, profile=1, features=SF["CHAINED"] | SF["FIRRAND"],
                           restart_interval=2), 4500, 11000),
             (syn.make_cfg(assignment=1, rate_code=0, n_aus=6, profile=1,
                           features=SF["MIDMATRIX"] | SF["PARAMBLOCKS"] | SF["MATRIXRAND"] | SF["VARBLOCK"] | SF["QSS"],
                           restart_interval=3), 700, 22000),
             (syn.make_cfg(assignment=12, rate_code=1, n_aus=6), 300, 33000)]
    streams, meta = [], []
    for cfg, n, seed0 in specs:
        flat, offs, sizes, frames = syn.batch(cfg, seed0, n)
        for o, z, f in zip(offs, sizes, frames):
            streams.append(flat[int(o):int(o + z)])
            meta.append((syn.channels(cfg.assignment), int(f)))
    pcm, infos = hip.decode_streams(streams, layout=hip.PCM_INTERLEAVED)
    n_chained = sum(1 for inf in infos if inf.status & hip.ST["CHAINED"])
    n_mid = sum(1 for inf in infos if inf.status & hip.ST["MIDFRAME"])
    assert n_chained >= 4400 and n_mid >= 100, (n_chained, n_mid)
    for i, (b, (nch, f)) in enumerate(zip(streams, meta)):
        want, r, st = oracle.decode(b, nch, f)
        assert st == 0 and r == f
        assert infos[i].status & ~hip.ST_BENIGN == 0, "stream %d status %#x" % (i, infos[i].status)
        assert infos[i].pcm_frames == f and np.array_equal(pcm[i], want), "stream %d differs" % i


def test_mixed_batch_picks_the_kernels_itself(pkg, oracle):
    """One- and two-substream streams in ONE batch with nothing said about lanes (the default): the library
    runs the one-lane kernel for the former and the two-wave kernel for the latter; forcing one lane makes a
    two-substream stream an envelope error as documented."""
    syn, hip = pkg.synth, pkg.hipdec
    cfgs = [syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=24),
            syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=24),
            syn.make_cfg(assignment=1, rate_code=0, n_substreams=1, n_aus=40),
            syn.make_cfg(assignment=0x14, rate_code=2, n_substreams=2, n_aus=12, profile=1, features=syn.SF_ALL,
                         restart_interval=3),
            syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=24, profile=1,
                         features=syn.SF["CHAINED"] | syn.SF["FIRRAND"], restart_interval=4)]
    streams = [syn.stream(c, 5150 + i) for i, c in enumerate(cfgs)]
    pcm, infos = _both(hip, [b for b, _ in streams])
    for (b, f), c, got, inf in zip(streams, cfgs, pcm, infos):
        want, r, st = oracle.decode(b, syn.channels(c.assignment), f)
        assert st == 0 and r == f
        assert inf.status & ~hip.ST_BENIGN == 0 and inf.substreams == c.n_substreams
        assert np.array_equal(got, want)
    _, forced = hip.decode_streams([b for b, _ in streams], lanes_per_segment=1)
    assert forced[1].status & hip.ST["ENVELOPE"] and forced[0].status == 0


@pytest.mark.parametrize("ss0", [1, 2, 3, 4, 5])
def test_two_substreams_of_any_split(pkg, oracle, ss0):
    """Any split of six channels over two substreams -- one + five, two + four (what discs carry) ... five + one -- comes
    out exact from the fast pass itself: since round 6 one lane reads both substreams of its segment and a channel's
    register slot is its channel number, so no split is "too wide" (rounds 1-5: the two-wave kernel kept four channels
    per substream and handed wider ones to the chain passes, DVDA_ST_COLD).  Forced to the lane kernels: a batch this
    small would otherwise get the cooperative one."""
    syn, hip = pkg.synth, pkg.hipdec
    cfgs = [syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=24, ss0_channels=ss0),
            syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=24, ss0_channels=ss0, profile=1,
                         features=syn.SF_FAST, restart_interval=5),
            syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=24, ss0_channels=ss0, profile=1,
                         features=syn.SF["CHAINED"] | syn.SF["FIRRAND"], restart_interval=4)]
    streams = [syn.stream(c, 9300 + 10 * ss0 + i) for i, c in enumerate(cfgs)]
    pcm, infos = _both(hip, [b for b, _ in streams], lanes_per_segment=2)
    for i, ((b, f), got, inf) in enumerate(zip(streams, pcm, infos)):
        want, r, st = oracle.decode(b, 6, f)
        assert st == 0 and r == f
        assert inf.status & ~hip.ST_BENIGN == 0 and inf.substreams == 2
        if i == 0:
            assert inf.status == 0, hex(inf.status)         # (the plain recipe: nothing is deferred)
        assert np.array_equal(got, want)


def test_substreams_that_disagree_on_an_access_units_length_are_reported(pkg, oracle):
    """A two-substream stream in which one access unit is 38 PCM frames long in substream 0 and 19 in substream 1
    (captured by tools/soak_reuse.py, seed 99: the test generator splits access units per substream when it varies
    their lengths).  The reference rematrixes over the first channel's length past the other channels' arrays and
    appends channels of different lengths (src/mlp.c:1308-1320, 598-603) -- outside what it defines.  The oracle
    reports it; the HIP path used to decode on with a clean status and other PCM.  Both must report it, and agree on
    everything in front of that access unit."""
    import os
    hip = pkg.hipdec
    z = np.load(os.path.join(os.path.dirname(__file__), "data", "ragged_substreams_2ch_2ss.npz"))
    b = z["mlp"]
    want, r, st = oracle.decode(b, 2, 7000)
    assert st & hip.ST["ENVELOPE"]
    pcm, infos = _both(hip, [b])
    assert infos[0].status & hip.ST["ENVELOPE"]
    from tests import stream_tools
    offs = stream_tools.frame_offsets(b)
    head, rh, sth = oracle.decode(b[:offs[39]], 2, 7000)           # the 39 access units in front of it
    assert sth == 0 and rh == 2026
    assert pcm[0].shape[1] >= rh and np.array_equal(pcm[0][:, :rh], head)


def test_two_wave_kernel_with_an_unaligned_output_offset_in_the_group(pkg, oracle):
    """Forced lane pairs, frame-major output, one group of 64 segments holding single-substream titles next to a
    two-substream one, and one of the single-substream titles at an output offset that is not a multiple of four
    values.  The two waves of the group used to choose different staging layouts (the idle twin of the
    single-substream segment never loaded the offset) and the two-substream title's channels came out transposed,
    status clean.  Found by tools/soak_reuse.py."""
    import torch
    syn, hip = pkg.synth, pkg.hipdec
    c1 = syn.make_cfg(assignment=12, rate_code=0, n_substreams=1, n_aus=16, restart_interval=2)
    c2 = syn.make_cfg(assignment=12, rate_code=0, n_substreams=2, n_aus=24, restart_interval=1)
    (b1, f1), (b2, f2) = syn.stream(c1, 81), syn.stream(c2, 82)
    w1 = oracle.decode(b1, 6, f1)[0]
    w2 = oracle.decode(b2, 6, f2)[0]
    flat, offs, lens = hip.pack_streams([b1, b1, b2])
    dev = torch.device("cuda", 0)
    d_bytes = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
    out_off = np.array([0, f1 * 6 + 2, f1 * 12 + 8], np.int64)          # the second title starts 2 values off
    stride = np.array([f1, f1, f2], np.int64)
    pcm = torch.zeros(int(out_off[2] + f2 * 6 + 16), dtype=torch.int32, device=dev)
    d_oo = torch.from_numpy(out_off).to(dev)
    d_stride = torch.from_numpy(stride).to(dev)
    for lanes in (2, 0):
        ctx = hip.Context(0, 3, 256, lanes_per_segment=lanes, layout=hip.PCM_INTERLEAVED)
        ctx.index(d_bytes.data_ptr(), len(flat) - 64, d_off.data_ptr(), d_len.data_ptr(), 3, 0)
        pcm.zero_()
        ctx.decode(pcm.data_ptr(), d_oo.data_ptr(), d_stride.data_ptr(), 0)
        infos = ctx.stream_info(3)
        host = pcm.cpu().numpy()
        ctx.close()
        for i, (want, f) in enumerate(((w1, f1), (w1, f1), (w2, f2))):
            assert infos[i].status & ~hip.ST_BENIGN == 0 and infos[i].pcm_frames == f
            got = host[out_off[i]:out_off[i] + f * 6].reshape(f, 6).T
            assert np.array_equal(got, want), (lanes, i)


def test_stream_ranges_are_checked_not_trusted(pkg, oracle):
    """The index looks streams up by offset.  Ranges that are not ascending and disjoint (the same bytes given
    twice, a list in descending order, a range that leaves the buffer, a misaligned start) used to send its
    walks anywhere -- a descending list ended in a GPU memory fault.  They are checked on the device: the
    offending streams are reported and not decoded, the others come out exact."""
    import torch
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=24)
    b, f = syn.stream(cfg, 5)
    want, r, st = oracle.decode(b, 6, f)
    flat, offs, lens = hip.pack_streams([b, b, b])
    dev = torch.device("cuda", 0)
    d_bytes = torch.from_numpy(flat).to(dev)
    total = len(flat) - 64
    bad = hip.ST["IRREGULAR"]

    def run(o, l):
        d_off = torch.from_numpy(np.asarray(o, np.int64)).to(dev)
        d_len = torch.from_numpy(np.asarray(l, np.int64)).to(dev)
        oo = torch.from_numpy(np.arange(3, dtype=np.int64) * f * 6).to(dev)
        stride = torch.from_numpy(np.full(3, f, np.int64)).to(dev)
        pcm = torch.zeros(3 * f * 6, dtype=torch.int32, device=dev)
        ctx = hip.Context(0, 3, 64)
        ctx.index(d_bytes.data_ptr(), total, d_off.data_ptr(), d_len.data_ptr(), 3, 0)
        ctx.decode(pcm.data_ptr(), oo.data_ptr(), stride.data_ptr(), 0)
        infos = ctx.stream_info(3)
        host = pcm.cpu().numpy().reshape(3, 6, f)
        ctx.close()
        return infos, host

    infos, host = run(offs, lens)
    assert all(i.status == 0 and i.pcm_frames == f for i in infos) and all(np.array_equal(host[i], want) for i in range(3))
    # the same bytes twice: the first copy decodes, the second is reported, the third stream is untouched
    infos, host = run([offs[0], offs[0], offs[2]], lens)
    assert infos[0].status == 0 and np.array_equal(host[0], want)
    assert infos[1].status & bad and infos[1].pcm_frames == 0
    assert infos[2].status == 0 and np.array_equal(host[2], want)
    # descending order: nothing behind the first entry is in front of it
    infos, host = run(offs[::-1].copy(), lens)
    assert infos[1].status & bad and infos[2].status & bad and infos[1].pcm_frames == 0 and infos[2].pcm_frames == 0
    assert infos[0].status & ~hip.ST_BENIGN or np.array_equal(host[0], want)
    # a range that leaves the buffer, a start that is not 16-byte aligned
    infos, host = run(offs, [lens[0], lens[1], lens[2] + (1 << 20)])
    assert infos[0].status == 0 and infos[1].status == 0 and infos[2].status & bad
    infos, host = run([offs[0], offs[1] + 2, offs[2]], [lens[0], lens[1] - 2, lens[2]])
    assert infos[0].status == 0 and infos[1].status & bad and infos[2].status == 0 and np.array_equal(host[2], want)
    # ONE bad entry does not cost the streams behind it: a range in the middle that leaves the buffer
    infos, host = run(offs, [lens[0], lens[1] + (1 << 40), lens[2]])
    assert infos[0].status == 0 and infos[1].status & bad and infos[1].pcm_frames == 0
    assert infos[2].status == 0 and np.array_equal(host[2], want) and np.array_equal(host[0], want)
    # the same with streams of different shapes in the batch (the lanes are then dealt by stream shape, and the
    # refused stream's major syncs must take no lane): 6-ch title, 2-ch title, the 2-ch title's bytes once more
    cfg2 = syn.make_cfg(assignment=1, rate_code=0, n_aus=40)
    b2, f2 = syn.stream(cfg2, 6)
    want2, r2, st2 = oracle.decode(b2, 2, f2)
    flat2, offs2, lens2 = hip.pack_streams([b, b2, b2])
    d_bytes2 = torch.from_numpy(flat2).to(dev)
    d_off = torch.from_numpy(np.asarray([offs2[0], offs2[1], offs2[1]], np.int64)).to(dev)
    d_len = torch.from_numpy(np.asarray(lens2, np.int64)).to(dev)
    cap = max(f, f2)
    oo = torch.from_numpy(np.arange(3, dtype=np.int64) * cap * 6).to(dev)
    stride = torch.from_numpy(np.full(3, cap, np.int64)).to(dev)
    pcm = torch.zeros(3 * cap * 6, dtype=torch.int32, device=dev)
    ctx = hip.Context(0, 3, 64)
    for _ in range(2):                      # (twice on one context: the second index replays the first one's launches)
        ctx.index(d_bytes2.data_ptr(), len(flat2) - 64, d_off.data_ptr(), d_len.data_ptr(), 3, 0)
        ctx.decode(pcm.data_ptr(), oo.data_ptr(), stride.data_ptr(), 0)
        infos = ctx.stream_info(3)
        host = pcm.cpu().numpy()
        assert infos[0].status == 0 and infos[0].pcm_frames == f
        assert np.array_equal(host[:cap * 6].reshape(6, cap)[:, :f], want)
        assert infos[1].status == 0 and infos[1].pcm_frames == f2
        assert np.array_equal(host[cap * 6:cap * 6 + 2 * cap].reshape(2, cap)[:, :f2], want2)
        assert infos[2].status & bad and infos[2].pcm_frames == 0
    ctx.close()


def test_decoding_one_index_more_than_once(pkg, oracle):
    """A caller may decode an index again -- it came back with a larger PCM buffer after DVDA_ST_OVERFLOW, or
    wants the same titles in another buffer.  What the first decode left on the segments (status bits, row
    counts, which segments the chain passes took) must not be taken by the second for its own findings: the
    overflow would stick, and the chain passes -- started from a count of NEWLY flagged segments -- would not run,
    leaving chained titles undecoded under a clean status."""
    import torch
    syn, hip = pkg.synth, pkg.hipdec
    SF = syn.SF
    cfgs = [syn.make_cfg(assignment=12, rate_code=1, n_aus=24),
            syn.make_cfg(assignment=12, rate_code=1, n_aus=24, profile=1, features=SF["CHAINED"] | SF["FIRRAND"],
                         restart_interval=4),
            syn.make_cfg(assignment=3, rate_code=0, n_substreams=2, n_aus=24, profile=1,
                         features=SF["CHAINED"] | SF["FIRRAND"], restart_interval=3),
            syn.make_cfg(assignment=6, rate_code=0, n_aus=20, profile=1, features=SF["VARROWS"] | SF["VARBLOCK"],
                         restart_interval=5),
            syn.make_cfg(assignment=1, rate_code=0, n_aus=40)]
    made = [syn.stream(c, 9100 + i) for i, c in enumerate(cfgs)]
    streams = [b for b, _ in made]
    nch = [syn.channels(c.assignment) for c in cfgs]
    want = [oracle.decode(b, n, f + 64) for (b, f), n in zip(made, nch)]
    assert all(st == 0 for _, _, st in want)
    flat, offs, lens = hip.pack_streams(streams)
    dev = torch.device("cuda", 0)
    d_bytes = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
    n = len(streams)
    ctx = hip.Context(0, n, 256)
    try:
        ctx.index(d_bytes.data_ptr(), len(flat) - 64, d_off.data_ptr(), d_len.data_ptr(), n, 0)
        rows = [r for _, r, _ in want]

        def run(cap):
            oo, pos = [], 0
            for c, k in zip(cap, nch):
                oo.append(pos)
                pos += c * k
            d_oo = torch.tensor(oo, dtype=torch.int64, device=dev)
            d_st = torch.tensor(cap, dtype=torch.int64, device=dev)
            pcm = torch.full((max(pos, 1),), 0x5A5A5A5A, dtype=torch.int32, device=dev)
            ctx.decode(pcm.data_ptr(), d_oo.data_ptr(), d_st.data_ptr(), 0)
            infos = ctx.stream_info(n)
            host = pcm.cpu().numpy()
            return [host[o:o + c * k].reshape(k, c) for o, c, k in zip(oo, cap, nch)], infos

        # first with half the room: every title overflows, says how much it needs
        got, infos = run([r // 2 for r in rows])
        assert all(inf.status & hip.ST["OVERFLOW"] for inf in infos)
        # then, same index, with room -- three times over, every time into a buffer that holds nothing yet
        for _ in range(3):
            got, infos = run(rows)
            for g, inf, (w, r, _) in zip(got, infos, want):
                assert (inf.status & ~hip.ST_BENIGN) == 0 and int(inf.pcm_frames) == r, hex(inf.status)
                assert np.array_equal(g[:, :r], w[:, :r])
    finally:
        ctx.close()


def test_too_small_a_context_is_reported_not_truncated(pkg, oracle):
    """more major syncs than max_segments used to come back as short PCM with status 0."""
    import torch
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=64, restart_interval=2)       # 32 segments per title
    streams = [syn.stream(cfg, 8800 + i)[0] for i in range(3)]
    flat, offs, lens = hip.pack_streams(streams)
    dev = torch.device("cuda", 0)
    d_bytes = torch.from_numpy(flat).to(dev)
    d_off = torch.from_numpy(offs.astype(np.int64)).to(dev)
    d_len = torch.from_numpy(lens.astype(np.int64)).to(dev)
    ctx = hip.Context(0, 3, 40)                     # holds the first title and a quarter of the second
    try:
        ctx.index(d_bytes.data_ptr(), len(flat) - 64, d_off.data_ptr(), d_len.data_ptr(), 3, 0)
        with pytest.raises(hip.HipError):
            ctx.segment_count()
        infos = ctx.stream_info()
        assert infos[0].status == 0 and infos[0].mlp_frames == 64
        assert infos[1].status & hip.ST["CAPACITY"] and infos[2].status & hip.ST["CAPACITY"]
    finally:
        ctx.close()
    # decode_streams grows the context by itself
    pcm, infos = hip.decode_streams(streams, max_segments=40)
    for b, got, inf in zip(streams, pcm, infos):
        want, r, st = oracle.decode(b, 6, 64 * 80)
        assert inf.status == 0 and np.array_equal(got, want)


@pytest.mark.parametrize("bits", [24, 16])
def test_wav_payload_straight_out_of_the_decode(pkg, oracle, bits):
    """DVDA_PCM_WAV24 / DVDA_PCM_WAV16: the output stage (dvda_read's interleave + dvda2wav's
    write_signed, reference src/dvd-audio.c:781-792, utils/dvda2wav.c:326-334, src/bitstream.c:2846-2857) fused
    into the decode -- fast pass (6-ch identity order, 2-ch, the 5-ch RIFF permutation 0x12, two substreams),
    chain passes (chained titles), sequential pass (non-standard timing) -- must equal the oracle's packing of the
    oracle's PCM byte for byte.  16-bit packing of 24-bit content exercises write_signed outside its range (the
    sign bit is taken from v < 0, not bit 15)."""
    syn, hip = pkg.synth, pkg.hipdec
    SF = syn.SF
    cfgs = [syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=24),
            syn.make_cfg(assignment=1, rate_code=0, n_substreams=1, n_aus=40),
            syn.make_cfg(assignment=0x12, rate_code=2, n_substreams=1, n_aus=12),
            syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=24),
            syn.make_cfg(assignment=0x14, rate_code=1, n_substreams=2, n_aus=16, profile=1, features=syn.SF_FAST,
                         restart_interval=3),
            syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=24, profile=1,
                         features=SF["CHAINED"] | SF["FIRRAND"], restart_interval=4),
            syn.make_cfg(assignment=6, rate_code=0, n_substreams=1, n_aus=20, profile=1,
                         features=SF["VARROWS"] | SF["VARBLOCK"], restart_interval=5),
            syn.make_cfg(assignment=0, rate_code=1, n_substreams=1, n_aus=9)]
    streams = [syn.stream(c, 7300 + i) for i, c in enumerate(cfgs)]
    got, infos = hip.decode_streams_wav([b for b, _ in streams], bits)
    for (b, f), c, payload, inf in zip(streams, cfgs, got, infos):
        nch = syn.channels(c.assignment)
        want, r, st = oracle.decode(b, nch, f)
        assert st == 0 and r == f
        assert inf.status & ~hip.ST_BENIGN == 0 and inf.pcm_frames == f, hex(inf.status)
        ref = np.frombuffer(oracle.wav_pack(want, bits), np.uint8)
        assert len(payload) == len(ref) == f * nch * bits // 8
        assert np.array_equal(payload, ref), "assignment %#x: first difference at byte %d" % (
            c.assignment, int(np.argmax(payload != ref)))
    # and it is what the separate pack kernel makes of the int32 PCM
    pcm, _ = hip.decode_streams([streams[0][0]])
    assert np.array_equal(hip.pack_wav(pcm[0], bits), got[0])


@pytest.mark.parametrize("fill", ["0xA5", "0xFF"])
def test_workspaces_full_of_garbage_change_nothing(fill):
    """DVDA_POISON fills every workspace the library allocates: a kernel that reads what no kernel wrote before it
    (a plan entry, a record terminator, an end-of-segment note of no segment) then fails the same way every time
    instead of once in a long while.  The deferred-feature, chain, mixed-batch and repeated-decode tests once more,
    in a process of their own (the switch is read once per process)."""
    import subprocess
    env = dict(os.environ, DVDA_POISON=fill)
    pick = "deferred or chains_and_midframe or mixed_batch or more_than_once or wav_payload or any_split or recipe"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p",
                        "no:cacheprovider", "-k", pick], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("layout", ["planar", "interleaved", "wav24"])
def test_title_list_dealt_to_several_devices_from_c(pkg, oracle, layout):
    """dvda_mlp_hip_decode_multi (SURVEY 8(e) from a C host): a list of host streams dealt to the entries of a device
    list -- here {0, 0, 0}: three contexts and three host threads on the one GPU a test box has -- every stream's PCM
    against the oracle, the summary against the sums, the partition against shard.py."""
    syn, hip = pkg.synth, pkg.hipdec
    streams, frames, nchs = [], [], []
    kinds = [dict(assignment=12, n_substreams=1, n_aus=40), dict(assignment=1, n_substreams=1, n_aus=72),
             dict(assignment=12, n_substreams=2, n_aus=24),
             dict(assignment=12, n_substreams=1, n_aus=32, profile=1, features=syn.SF["CHAINED"])]
    for i in range(14):
        cfg = syn.make_cfg(rate_code=1, **kinds[i % len(kinds)])
        b, f = syn.stream(cfg, 300 + i)
        streams.append(b)
        frames.append(f)
        nchs.append(syn.channels(cfg.assignment))
    lay = {"planar": hip.PCM_PLANAR, "interleaved": hip.PCM_INTERLEAVED, "wav24": hip.PCM_WAV24}[layout]
    pcm, infos, summ = hip.decode_streams_multi(streams, [0, 0, 0], layout=lay)
    assert summ.devices == 3 and summ.streams_with_errors == 0
    assert summ.reduction == 0          # (a device named more than once: the summary is added up on the host)
    assert summ.pcm_frames == sum(frames) and summ.samples == sum(f * c for f, c in zip(frames, nchs))
    assert summ.compressed_bytes == sum(len(b) for b in streams)
    owner = hip.shard_c([len(b) for b in streams], 3)
    assert summ.compressed_bytes_max_device == max(sum(len(b) for b, o in zip(streams, owner) if o == r) for r in range(3))
    # (round 5) what a run on N devices says about itself: the deal's imbalance and the device entries' wall times
    assert abs(summ.imbalance - summ.compressed_bytes_max_device * 3 / summ.compressed_bytes) < 1e-9 and summ.imbalance >= 1.0
    assert 0 < summ.device_ms_min <= summ.device_ms_max
    for i, (b, f, nch) in enumerate(zip(streams, frames, nchs)):
        want, r, st = oracle.decode(b, nch, f)
        assert st == 0 and infos[i].pcm_frames == f and infos[i].channels == nch
        assert not (infos[i].status & ~hip.ST_BENIGN)
        if layout == "wav24":
            assert pcm[i].tobytes() == oracle.wav_pack(want, 24), "stream %d" % i
        else:
            assert np.array_equal(pcm[i], want), "stream %d" % i


def test_multi_summary_goes_over_rccl_when_every_device_is_named_once(pkg, oracle):
    """The C host's one reduction (north_star: "partitioned with RCCL over xGMI only as an
    embarrassingly-parallel shard of the title list"): with a device list that names every device once the summary of
    dvda_mlp_hip_decode_multi is two RCCL all-reduces (csrc/mlp_multi.cpp; librccl opened at run time) -- on a test
    box that is a communicator of ONE rank, which still goes through ncclCommInitAll / ncclAllReduce --; with
    DVDA_MULTI_RCCL=0 the host adds up.  Both give the same summary, and the PCM is the oracle's either way."""
    import os
    syn, hip = pkg.synth, pkg.hipdec
    streams, frames = [], []
    for i in range(6):
        b, f = syn.stream(syn.make_cfg(assignment=12, rate_code=1, n_substreams=1 + (i & 1), n_aus=24), 900 + i)
        streams.append(b)
        frames.append(f)
    got = {}
    old = os.environ.get("DVDA_MULTI_RCCL")
    try:
        for mode in ("1", "0"):
            os.environ["DVDA_MULTI_RCCL"] = mode
            pcm, infos, summ = hip.decode_streams_multi(streams, [0])
            got[mode] = summ
            for b, f, p in zip(streams, frames, pcm):
                want, r, st = oracle.decode(b, 6, f)
                assert st == 0 and np.array_equal(p, want)
    finally:
        if old is None:
            del os.environ["DVDA_MULTI_RCCL"]
        else:
            os.environ["DVDA_MULTI_RCCL"] = old
    assert got["0"].reduction == 0
    assert got["1"].reduction == 1, "librccl.so is part of the ROCm image: the summary should have gone over RCCL"
    for k in ("pcm_frames", "samples", "compressed_bytes", "compressed_bytes_max_device", "streams_with_errors", "devices"):
        assert getattr(got["0"], k) == getattr(got["1"], k), k
    assert got["1"].pcm_frames == sum(frames) and got["1"].samples == 6 * sum(frames)
    assert got["1"].device_ms_max > 0 and abs(got["1"].imbalance - 1.0) < 1e-9
