"""CPU tests of the synthetic stream generator and host-side packing helpers."""
import numpy as np


def test_generator_is_deterministic(pkg):
    syn = pkg.synth
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=16)
    a, fa = syn.stream(cfg, 7)
    b, fb = syn.stream(cfg, 7)
    c, _ = syn.stream(cfg, 8)
    assert fa == fb == 16 * 80
    assert np.array_equal(a, b)
    assert not np.array_equal(a[:len(c)], c[:len(a)])


def test_recipe_matches_baseline_size(pkg):
    """~984 B per 6-ch access unit, ~346 B per 2-ch one."""
    syn = pkg.synth
    b6, _ = syn.stream(syn.make_cfg(assignment=12, rate_code=1, n_aus=256), 1)
    b2, _ = syn.stream(syn.make_cfg(assignment=1, rate_code=1, n_aus=256), 1)
    assert 950 < len(b6) / 256 < 1010
    assert 330 < len(b2) / 256 < 360


def test_frames_chain_and_syncs(pkg):
    syn = pkg.synth
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_aus=40, restart_interval=8)
    b, _ = syn.stream(cfg, 3)
    pos, n, syncs = 0, 0, 0
    while pos < len(b):
        size = 2 * (((int(b[pos]) & 0xF) << 8) | int(b[pos + 1]))
        if bytes(b[pos + 4:pos + 8]) == b"\xf8\x72\x6f\xbb":
            syncs += 1
        pos += size
        n += 1
    assert pos == len(b) and n == 40 and syncs == 5


def test_batch_layout(pkg):
    syn = pkg.synth
    cfg = syn.make_cfg(assignment=1, rate_code=0, n_aus=8)
    flat, off, siz, frm = syn.batch(cfg, 10, 5, threads=2)
    assert len(flat) % 16 == 0 and (off % 16 == 0).all()
    for i in range(5):
        one, f = syn.stream(cfg, 10 + i)
        assert f == frm[i] and siz[i] == len(one)
        assert np.array_equal(flat[int(off[i]):int(off[i] + siz[i])], one)


def test_pack_streams(pkg):
    a = np.arange(5, dtype=np.uint8)
    b = np.arange(33, dtype=np.uint8)
    flat, off, lens = pkg.hipdec.pack_streams([a, b])
    assert list(off) == [0, 16] and list(lens) == [5, 33]
    assert len(flat) == 16 + 48 + 64
    assert np.array_equal(flat[16:49], b)
