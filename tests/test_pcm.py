"""PCM un-swizzle: CPU tests pin the oracle to the compiled reference and to
the sample values that went into the muxer; GPU tests compare the HIP kernels with the oracle."""
import ctypes
import os

import numpy as np
import pytest

from tests import oracle_lib

LAYOUTS = [(16, 1, 0), (16, 2, 1), (16, 3, 2), (16, 4, 3), (16, 5, 6), (16, 6, 12),
           (24, 1, 0), (24, 2, 1), (24, 3, 2), (24, 4, 3), (24, 5, 6), (24, 6, 12)]


def _oracle():
    lib = ctypes.CDLL(oracle_lib.build_oracle())
    lib.pcm_oracle_unswizzle.restype = ctypes.c_long
    lib.pcm_oracle_unswizzle.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint,
                                         ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
    lib.pcm_oracle_decode_sectors.restype = ctypes.c_long
    lib.pcm_oracle_decode_sectors.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint,
                                              ctypes.c_void_p, ctypes.c_size_t]
    return lib


def _samples(bps, ch, frames, seed):
    rng = np.random.RandomState(seed)
    lim = 1 << (bps - 1)
    s = rng.randint(-lim, lim, size=(frames, ch))
    s[0, :] = -lim
    s[1, :] = lim - 1
    return s


def _sectors(pkg, bps, ch, asg, frames, seed):
    s = _samples(bps, ch, frames, seed)
    sec = pkg.disc.pcm_track_sectors(s, {16: 0, 24: 2}[bps], 1, asg)
    return np.frombuffer(b"".join(sec), np.uint8).copy(), s


@pytest.mark.parametrize("bps,ch,asg", LAYOUTS)
def test_oracle_sector_walk_recovers_the_muxed_samples(pkg, bps, ch, asg):
    lib = _oracle()
    data, s = _sectors(pkg, bps, ch, asg, 1500, 3)
    out = np.zeros((ch, len(s) + 8), np.int32)
    r = lib.pcm_oracle_decode_sectors(data.ctypes.data, len(data) // 2048, bps, ch, out.ctypes.data, out.shape[1])
    assert r == len(s)
    assert np.array_equal(out[:, :r], s.T)


@pytest.mark.skipif(not oracle_lib.Reference.available(), reason="compiled reference not present")
@pytest.mark.parametrize("bps,ch,asg", LAYOUTS)
def test_oracle_unswizzle_equals_reference(bps, ch, asg):
    lib = _oracle()
    ref = ctypes.CDLL(oracle_lib.REF_SO)
    ref.ref_pcm_decode.restype = ctypes.c_long
    ref.ref_pcm_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_uint,
                                   ctypes.c_void_p, ctypes.c_size_t]
    rng = np.random.RandomState(bps * 10 + ch)
    for n in (0, 5, 36 * 7 + 11, 1993):
        payload = rng.randint(0, 256, size=n).astype(np.uint8)
        a = np.zeros((ch, 2100), np.int32)
        b = np.zeros((ch, 2100), np.int32)
        ra = lib.pcm_oracle_unswizzle(payload.ctypes.data, n, bps, ch, a.ctypes.data, 2100, 0)
        rb = ref.ref_pcm_decode(payload.ctypes.data, n, bps, ch, b.ctypes.data, 2100)
        assert ra == rb == 2 * (n // (2 * ch * (bps // 8)))
        assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("bps,ch,asg", LAYOUTS)
def test_gpu_unswizzle_matches_oracle(pkg, bps, ch, asg):
    lib = _oracle()
    data, s = _sectors(pkg, bps, ch, asg, 5000 + 2 * ch, 11)
    want = np.zeros((ch, len(s) + 8), np.int32)
    r = lib.pcm_oracle_decode_sectors(data.ctypes.data, len(data) // 2048, bps, ch, want.ctypes.data, want.shape[1])
    got, bad = pkg.hipdec.pcm_decode_sectors(data, bps, ch)
    assert bad == 0 and got.shape[1] == r == len(s)
    assert np.array_equal(got, want[:, :r])
    assert np.array_equal(got, s.T)


@pytest.mark.gpu
def test_gpu_pcm_large_track_and_bad_sector(pkg):
    lib = _oracle()
    data, s = _sectors(pkg, 24, 6, 12, 6 * 110 * 4100, 5)       # > 4096 sectors: multi-block scan
    got, bad = pkg.hipdec.pcm_decode_sectors(data, 24, 6)
    assert bad == 0 and np.array_equal(got, s.T)
    broken = data.copy()
    broken[2048 * 3 + 3] = 0                                     # sector 3 loses its pack start code
    got, bad = pkg.hipdec.pcm_decode_sectors(broken, 24, 6)
    assert bad == 1
    n3 = 110                                                     # frames sector 3 carried
    assert got.shape[1] == len(s) - n3
    assert np.array_equal(got[:, :3 * n3], s.T[:, :3 * n3]) and np.array_equal(got[:, 3 * n3:], s.T[:, 4 * n3:])


@pytest.mark.gpu
@pytest.mark.parametrize("n_aus,S", [(9, 1), (200, 2), (2300, 1)])
def test_gpu_mlp_track_demux_then_decode(pkg, oracle, n_aus, S):
    """AOB sectors -> MLP bytes on the GPU (must equal what went into the muxer),
    then straight into the decode path."""
    syn = pkg.synth
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=n_aus)
    data, frames = syn.stream(cfg, 31)
    sectors = np.frombuffer(b"".join(pkg.disc.mlp_track_sectors(data)), np.uint8).copy()
    got, bad = pkg.hipdec.mlp_demux_sectors(sectors)
    assert bad == 0
    assert len(got) == len(data) and np.array_equal(got, data)
    pcm, infos = pkg.hipdec.decode_streams([got], lanes_per_segment=2)
    want, r, st = oracle.decode(data, 6, frames)
    assert st == 0 and infos[0].status & ~pkg.hipdec.ST_BENIGN == 0
    assert np.array_equal(pcm[0], want)


def _wav_oracle():
    lib = _oracle()
    lib.wav_oracle_pack.restype = ctypes.c_long
    lib.wav_oracle_pack.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_size_t,
                                    ctypes.c_uint, ctypes.c_void_p]
    return lib


def _wide_samples(ch, frames, seed):
    """values inside and OUTSIDE the nominal width: write_signed is not plain truncation there"""
    rng = np.random.RandomState(seed)
    s = rng.randint(-(1 << 25), 1 << 25, size=(ch, frames)).astype(np.int32)
    s[:, ::3] = rng.randint(-(1 << 15), 1 << 15, size=s[:, ::3].shape)
    s[0, :4] = [(1 << 23), -(1 << 23) - 1, (1 << 31) - 1, -(1 << 31)]
    return s


@pytest.mark.skipif(not oracle_lib.Reference.available(), reason="compiled reference not present")
@pytest.mark.parametrize("bits", [16, 24])
@pytest.mark.parametrize("ch", [1, 2, 5, 6])
def test_wav_pack_oracle_equals_reference_writer(bits, ch):
    lib = _wav_oracle()
    ref = ctypes.CDLL(oracle_lib.REF_SO)
    ref.ref_wav_pack.restype = ctypes.c_long
    ref.ref_wav_pack.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint, ctypes.c_void_p]
    s = _wide_samples(ch, 333, ch + bits)
    inter = np.ascontiguousarray(s.T.reshape(-1))
    a = np.zeros(len(inter) * 3 + 8, np.uint8)
    b = np.zeros(len(inter) * 3 + 8, np.uint8)
    na = lib.wav_oracle_pack(s.ctypes.data, s.shape[1], ch, s.shape[1], bits, a.ctypes.data)
    nb = ref.ref_wav_pack(inter.ctypes.data, len(inter), bits, b.ctypes.data)
    assert na == nb == len(inter) * bits // 8
    assert np.array_equal(a[:na], b[:nb])


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [16, 24])
@pytest.mark.parametrize("ch,frames", [(1, 1), (2, 255), (3, 256), (5, 1000), (6, 4099), (1, 5000), (2, 2048),
                                       (3, 3333), (4, 10240), (5, 7777), (6, 1024)])
def test_gpu_wav_pack_matches_oracle(pkg, bits, ch, frames):
    lib = _wav_oracle()
    s = _wide_samples(ch, max(frames, 4), frames)[:, :frames].copy()
    want = np.zeros(frames * ch * 3 + 8, np.uint8)
    n = lib.wav_oracle_pack(s.ctypes.data, frames, ch, frames, bits, want.ctypes.data)
    got = pkg.hipdec.pack_wav(s, bits)
    assert len(got) == n and np.array_equal(got, want[:n])


@pytest.mark.gpu
def test_gpu_sector_with_more_audio_packets_than_the_kernels_keep(pkg):
    """a crafted sector with more than 8 small audio packets used to be counted in full by the scan
    kernels while the gather kernels kept 8 -- output offsets for bytes nobody writes.  Such a sector is now
    malformed for every kernel alike: counted as bad, contributing nothing, the sectors around it unaffected
    (PCM tier and MLP demux)."""
    import struct
    disc = pkg.disc

    def many_packets(codec, n, body, params=b""):
        pes = b""
        for _ in range(n):
            payload = b"\x81\x00" + b"\x00" + bytes([codec, 0, 0, len(params)]) + params + body
            pes += b"\x00\x00\x01\xBD" + struct.pack(">H", len(payload)) + payload
        room = 2048 - 14 - len(pes)
        assert room >= 6
        tail = b"\x00\x00\x01\xBE" + struct.pack(">H", room - 6) + b"\xFF" * (room - 6)
        return disc._pack_header() + pes + tail

    # ---- PCM: 2-ch / 16-bit, 9 packets of two chunks (4 frames) each between two regular sectors
    s = _samples(16, 2, 1000, 9)
    regular = disc.pcm_track_sectors(s, 0, 1, 1)
    params = bytes([0, 0, 0, 0x00, 0x11, 0, 1, 0, 0])             # 9-byte block: bps codes 0, rate 1, assignment 1
    crafted = many_packets(0xA0, 9, b"\x01\x02\x03\x04\x05\x06\x07\x08" * 2, params)
    ok8 = many_packets(0xA0, 8, b"\x01\x02\x03\x04\x05\x06\x07\x08" * 2, params)
    data = np.frombuffer(regular[0] + crafted + regular[1], np.uint8).copy()
    got, bad = pkg.hipdec.pcm_decode_sectors(data, 16, 2)
    per = len(s) if len(regular) == 1 else None
    assert bad == 1
    want_a, _ = pkg.hipdec.pcm_decode_sectors(np.frombuffer(regular[0] + regular[1], np.uint8).copy(), 16, 2)
    assert np.array_equal(got, want_a)                             # the crafted sector contributes nothing
    got8, bad8 = pkg.hipdec.pcm_decode_sectors(np.frombuffer(regular[0] + ok8 + regular[1], np.uint8).copy(), 16, 2)
    assert bad8 == 0 and got8.shape[1] == want_a.shape[1] + 8 * 4  # eight packets are still fine
    # ---- MLP demux
    mlp = disc.mlp_track_sectors(np.arange(5000, dtype=np.uint32).astype(np.uint8))
    crafted_m = many_packets(0xA1, 9, b"\xAA" * 20)
    sec = np.frombuffer(mlp[0] + crafted_m + mlp[1], np.uint8).copy()
    out, badm = pkg.hipdec.mlp_demux_sectors(sec)
    ref, _ = pkg.hipdec.mlp_demux_sectors(np.frombuffer(mlp[0] + mlp[1], np.uint8).copy())
    assert badm == 1 and np.array_equal(out, ref)
