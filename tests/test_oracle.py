"""CPU tests of the oracle: pinned against the committed golden vectors (produced by the
real reference, tests/golden/make_golden.py), the reference's own bit-reader known answers,
and -- where oracle/_ref exists -- the compiled reference on fresh fuzz streams."""
import ctypes
import glob
import os

import numpy as np
import pytest

from tests import oracle_lib

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def test_golden_present():
    assert len(GOLDEN) >= 12


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
@pytest.mark.parametrize("chunk", [0, 2013, 97])
def test_oracle_matches_golden(oracle, path, chunk):
    z = np.load(path)
    want = z["pcm"]
    got, frames, status = oracle.decode(z["mlp"], want.shape[0], want.shape[1], chunk=chunk)
    # bit 1 is information: later major syncs with other stream parameters were dropped, as the reference
    # drops them (src/mlp.c:449-460) -- the sync_change_* vectors are exactly that, produced by the reference
    assert status & ~2 == 0
    assert bool(status & 2) == os.path.basename(path).startswith("sync_change")
    assert frames == want.shape[1]
    assert np.array_equal(got, want)


def test_bitreader_known_answers(oracle):
    # reference src/bitstream.c:4864-4868 and 4940-4944 on the bytes B1 ED 3B C1
    data = (ctypes.c_uint8 * 4)(0xB1, 0xED, 0x3B, 0xC1)
    lib = oracle.lib
    lib.mlp_oracle_test_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int),
                                         ctypes.c_int, ctypes.POINTER(ctypes.c_long)]
    for widths, want in (([2, 3, 5, 3, 19], [2, 6, 7, 5, 0x53BC1]),
                         ([-2, -3, -5, -3, -19], [-2, -2, 7, -3, -181311]),
                         ([0, 8, 0, 24], [0, 0xB1, 0, 0xED3BC1])):
        w = (ctypes.c_int * len(widths))(*widths)
        out = (ctypes.c_long * len(widths))()
        eof = lib.mlp_oracle_test_read(data, 4, w, len(widths), out)
        assert eof == 0
        assert list(out) == want
    # reading past the end raises the EOF condition (reference: br_abort -> longjmp)
    w = (ctypes.c_int * 2)(32, 1)
    out = (ctypes.c_long * 2)()
    assert lib.mlp_oracle_test_read(data, 4, w, 2, out) == 1


def test_crc8_table(oracle):
    # rows of the reference table, src/mlp.c:1364-1366 and :1395
    first = [0x00, 0x63, 0xC6, 0xA5, 0xEF, 0x8C, 0x29, 0x4A, 0xBD, 0xDE, 0x7B, 0x18, 0x52, 0x31, 0x94, 0xF7,
             0x19, 0x7A, 0xDF, 0xBC, 0xF6, 0x95, 0x30, 0x53]
    last = [0x3A, 0x59, 0xFC, 0x9F, 0xD5, 0xB6, 0x13, 0x70]
    got = [oracle.lib.mlp_oracle_test_crc8(i) for i in range(256)]
    assert got[:24] == first
    assert got[248:] == last


# value -> code string, from reference src/mlp_codebook{1,2,3}.json
BOOKS = {
    1: {0: "000000001", 1: "00000001", 2: "0000001", 3: "000001", 4: "00001", 5: "0001", 6: "001",
        7: "100", 8: "101", 9: "110", 10: "111", 11: "011", 12: "0101", 13: "01001", 14: "010001",
        15: "0100001", 16: "01000001", 17: "010000001"},
    2: {0: "000000001", 1: "00000001", 2: "0000001", 3: "000001", 4: "00001", 5: "0001", 6: "001",
        7: "10", 8: "11", 9: "011", 10: "0101", 11: "01001", 12: "010001", 13: "0100001",
        14: "01000001", 15: "010000001"},
    3: {0: "000000001", 1: "00000001", 2: "0000001", 3: "000001", 4: "00001", 5: "0001", 6: "001",
        7: "1", 8: "011", 9: "0101", 10: "01001", 11: "010001", 12: "0100001", 13: "01000001",
        14: "010000001"},
}


@pytest.mark.parametrize("book", [1, 2, 3])
def test_codebook_lut(oracle, book):
    lut = {}
    for value, code in BOOKS[book].items():
        lo = int(code, 2) << (9 - len(code))
        for i in range(1 << (9 - len(code))):
            lut[lo + i] = (value, len(code))
    for peek in range(512):
        e = oracle.lib.mlp_oracle_test_huff(book, peek)
        if peek in lut:
            assert (e & 0xFF, e >> 8) == lut[peek]
        else:
            assert e & 0xFF == 0xFF          # the two invalid all-zero-tail codes
            assert peek in (0, 0x80)


@pytest.mark.skipif(not oracle_lib.Reference.available(), reason="compiled reference not present")
def test_oracle_vs_compiled_reference(oracle, pkg):
    syn = pkg.synth
    ref = oracle_lib.Reference()
    n = 0
    for asg, S in ((12, 1), (12, 2), (1, 1), (0x14, 2), (0, 1)):
        for rate in (0, 1, 2):
            for feat in (0, syn.SF_ALL, syn.SF_FAST):
                for seed in range(2):
                    cfg = syn.make_cfg(assignment=asg, rate_code=rate, n_substreams=S, n_aus=16,
                                       profile=1 if feat else 0, features=feat,
                                       restart_interval=[8, 3][seed])
                    data, frames = syn.stream(cfg, 1000 + n)
                    want, r = ref.decode(data, asg, rate, cfg.bps_code, frames)
                    got, r2, st = oracle.decode(data, syn.channels(asg), frames, chunk=[0, 777][seed])
                    assert st == 0 and r == r2 == frames
                    assert np.array_equal(got, want)
                    n += 1
    assert n == 90


@pytest.mark.skipif(not oracle_lib.Reference.available(), reason="compiled reference not present")
def test_oracle_vs_reference_on_major_syncs_without_restart_headers(oracle, pkg):
    """reference src/mlp.c:449-460, 748-753: a major sync in front of an access unit whose blocks carry no restart
    header (generator feature SYNCONLY) -- the parameters are compared, the decode goes on with the state it has."""
    syn = pkg.synth
    ref = oracle_lib.Reference()
    n = 0
    for asg, S in ((12, 1), (12, 2), (1, 1), (0x14, 2)):
        for feat in (0, syn.SF_ALL, syn.SF_FAST, syn.SF["CHAINED"] | syn.SF["FIRRAND"]):
            for seed in range(2):
                cfg = syn.make_cfg(assignment=asg, rate_code=1, n_substreams=S, n_aus=32, profile=1,
                                   features=feat | syn.SF["SYNCONLY"], restart_interval=[8, 3][seed])
                data, frames = syn.stream(cfg, 2000 + n)
                v = data[:len(data) - len(data) % 2].reshape(-1, 2)
                syncs = int(np.count_nonzero((v[:-1, 0] == 0xF8) & (v[:-1, 1] == 0x72) & (v[1:, 0] == 0x6F) & (v[1:, 1] == 0xBB)))
                assert syncs > (32 + [8, 3][seed] - 1) // [8, 3][seed]          # more syncs than restart points
                want, r = ref.decode(data, asg, 1, cfg.bps_code, frames)
                got, r2, st = oracle.decode(data, syn.channels(asg), frames, chunk=[0, 777][seed])
                assert st == 0 and r == r2 == frames
                assert np.array_equal(got, want)
                n += 1
    assert n == 32


def test_oracle_flags_corruption(oracle, pkg):
    """Where the reference would assert()/abort, the oracle records an error bit."""
    syn = pkg.synth
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=8)
    data, frames = syn.stream(cfg, 5)
    bad = data.copy()
    bad[len(bad) // 2] ^= 0x10          # payload bit flip -> parity / CRC mismatch
    _, _, st = oracle.decode(bad, 6, frames)
    assert st & (4 | 8)
    # a truncated stream simply leaves the tail queued: fewer frames, no error
    _, r, st = oracle.decode(data[:len(data) - 11], 6, frames)
    assert st == 0 and r == frames - 80
    # empty input
    _, r, st = oracle.decode(data[:0], 6, 16)
    assert r == 0 and st == 0


@pytest.mark.skipif(not oracle_lib.Reference.available(), reason="compiled reference not present")
@pytest.mark.parametrize("S", [1, 2])
def test_oracle_vs_reference_on_changed_major_syncs(oracle, pkg, S):
    """reference src/mlp.c:449-460: an access unit whose major sync announces other stream parameters
    than the first one is dropped (restart header and all) and decoding goes on with the state the decoder
    has.  The restatement must do exactly that: same PCM, 80 frames fewer per dropped unit.  (The reference
    runs in a child process: should it assert() on such a stream the test fails instead of the session.)"""
    import subprocess
    import sys
    import tempfile
    from tests import stream_tools
    syn = pkg.synth
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=40, restart_interval=4)
    data, frames = syn.stream(cfg, 4040 + S)
    changed, _ = stream_tools.change_sync_params(data, (1, 4, 5, 6, 9), g1_bps=0, assignment=None)
    want_frames = frames - 5 * 80
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "in.npy"), changed)
        code = ("import sys, numpy as np; sys.path.insert(0, %r); from tests import oracle_lib; "
                "d = np.load(%r); pcm, r = oracle_lib.Reference().decode(d, 12, 1, 2, %d, chunk=1999); "
                "np.save(%r, pcm)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                      os.path.join(tmp, "in.npy"), frames, os.path.join(tmp, "out.npy")))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        want = np.load(os.path.join(tmp, "out.npy"))
    assert want.shape == (6, want_frames)
    got, r2, st = oracle.decode(changed, 6, frames)
    assert st == 2 and r2 == want_frames and np.array_equal(got, want)


@pytest.mark.skipif(not oracle_lib.Reference.available(), reason="compiled reference not present")
def test_oracle_vs_reference_on_a_long_run_of_changed_major_syncs(oracle, pkg):
    """src/mlp.c:449-460 drops ANY number of consecutive access units whose major sync announces other parameters.
    The HIP index walks through at most 64 in a row (csrc/mlp_index.h MAX_DROP) and reports a longer run
    (tests/test_gpu_parity.py); the restatement has no such bound: 66 in a row against the compiled reference."""
    import subprocess
    import sys
    import tempfile
    from tests import stream_tools
    syn = pkg.synth
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=72, restart_interval=1)
    data, frames = syn.stream(cfg, 4100)
    which = tuple(range(2, 68))
    changed, _ = stream_tools.change_sync_params(data, which, g1_bps=0, assignment=None)
    want_frames = frames - len(which) * 80
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "in.npy"), changed)
        code = ("import sys, numpy as np; sys.path.insert(0, %r); from tests import oracle_lib; "
                "d = np.load(%r); pcm, r = oracle_lib.Reference().decode(d, 12, 1, 2, %d, chunk=1999); "
                "np.save(%r, pcm)" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                      os.path.join(tmp, "in.npy"), frames, os.path.join(tmp, "out.npy")))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        want = np.load(os.path.join(tmp, "out.npy"))
    assert want.shape == (6, want_frames)
    got, r2, st = oracle.decode(changed, 6, frames)
    assert st == 2 and r2 == want_frames and np.array_equal(got, want)


def test_oracle_reports_substreams_of_different_access_unit_length():
    """src/mlp.c:1308-1320, 598-603: with substreams that disagree on an access unit's length the reference reads one
    substream's channels past their arrays and appends channels of different lengths.  The restatement reports the
    access unit (MLP_ORA_ERR_ENVELOPE) instead of following it there."""
    import os
    from tests import oracle_lib, stream_tools
    z = np.load(os.path.join(os.path.dirname(__file__), "data", "ragged_substreams_2ch_2ss.npz"))
    b = z["mlp"]
    o = oracle_lib.Oracle()
    want, r, st = o.decode(b, 2, 7000)
    assert st & 0x200
    offs = stream_tools.frame_offsets(b)
    head, rh, sth = o.decode(b[:offs[39]], 2, 7000)
    assert sth == 0 and rh == 2026 and np.array_equal(want[:, :rh], head)
