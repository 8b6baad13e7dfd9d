"""The wave-cooperative kernel (csrc/mlp_coop.h: one wave per (segment, substream), what small batches get) against
the same oracle, by the same cases as the lane kernels: the parity tests of tests/test_gpu_parity.py run again with
the kernel choice forced to it (dvda_mlp_hip_set_lanes_per_segment(ctx, 64)), plus the cases where the library picks
it by itself (lanes 0, a batch below the device-side threshold)."""
import numpy as np
import pytest

from tests import test_gpu_parity as T

pytestmark = pytest.mark.gpu


@pytest.fixture
def coop(pkg, monkeypatch):
    """Every decode of the test goes through the cooperative kernel, whatever lane count the case asks for."""
    hip = pkg.hipdec
    real, real_wav, real_ctx = hip.decode_streams, hip.decode_streams_wav, hip.Context

    def ds(streams, *a, **kw):
        kw["lanes_per_segment"] = 64
        return real(streams, *a, **kw)

    def dsw(streams, bits, *a, **kw):
        kw["lanes_per_segment"] = 64
        return real_wav(streams, bits, *a, **kw)

    monkeypatch.setattr(hip, "decode_streams", ds)
    monkeypatch.setattr(hip, "decode_streams_wav", dsw)
    return hip


def test_recipes(pkg, oracle, coop):
    T.test_recipe_6ch_96k(pkg, oracle)
    T.test_recipe_2ch_96k(pkg, oracle)
    T.test_recipe_two_substreams(pkg, oracle)


@pytest.mark.parametrize("rate", [0, 1, 2])
@pytest.mark.parametrize("assignment,S", [(12, 1), (12, 2), (1, 1), (0, 1), (0x14, 2), (0x12, 1), (6, 2)])
def test_fuzz_fast_features(pkg, oracle, coop, assignment, S, rate):
    T.test_fuzz_fast_features(pkg, oracle, assignment, S, rate)


def test_golden_vectors(pkg, coop):
    T.test_golden_vectors_on_gpu(pkg)


@pytest.mark.parametrize("feature", ["CHAINED", "MIDMATRIX", "MIDRESTART", "VARROWS"])
@pytest.mark.parametrize("S", [1, 2])
def test_features_the_lane_kernel_defers(pkg, oracle, coop, feature, S):
    """Mid-frame matrix changes and restarts are decoded by the cooperative kernel itself (it rematrixes per access
    unit, like the reference); chained history and non-standard timing are deferred as in the lane kernel."""
    T.test_deferred_features_general_pass(pkg, oracle, feature, S)


@pytest.mark.parametrize("S", [1, 2])
def test_fuzz_all_features(pkg, oracle, coop, S):
    T.test_fuzz_all_features(pkg, oracle, S)


def test_edge_cases_and_corruption(pkg, oracle, coop):
    T.test_edge_cases_truncated_ragged_and_single_unit(pkg, oracle)
    T.test_corruption_is_reported_not_decoded(pkg, oracle)
    T.test_long_last_access_unit_is_timing_not_overflow(pkg, oracle)


@pytest.mark.timeout(180)
@pytest.mark.parametrize("S", [1, 2])
def test_garbage_and_bit_flips(pkg, oracle, coop, S):
    T.test_garbage_and_bit_flips_never_hang_and_never_pass_silently(pkg, oracle, S)


@pytest.mark.parametrize("codebook", [1, 2, 3])
def test_widest_symbols(pkg, oracle, coop, codebook):
    T.test_widest_symbols_step_the_bit_window_by_two_dwords(pkg, oracle, codebook)


@pytest.mark.parametrize("S", [1, 2])
def test_config3_1024_independent_access_units(pkg, oracle, coop, S):
    T.test_config3_1024_independent_access_units(pkg, oracle, S)


@pytest.mark.parametrize("S", [1, 2])
def test_changed_major_syncs(pkg, oracle, coop, S):
    T.test_changed_major_syncs_are_dropped_like_the_reference(pkg, oracle, S)


@pytest.mark.parametrize("ss0", [1, 2, 3, 4, 5])
def test_two_substreams_of_any_split(pkg, oracle, coop, ss0):
    """Any split of six channels over two substreams (the lane kernel hands a five-channel substream to the chain
    passes; here a substream's channels are lanes of its wave, up to six)."""
    syn, hip = pkg.synth, pkg.hipdec
    cfgs = [syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=24, ss0_channels=ss0),
            syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=24, ss0_channels=ss0, profile=1,
                         features=syn.SF_FAST, restart_interval=5),
            syn.make_cfg(assignment=12, rate_code=1, n_substreams=2, n_aus=24, ss0_channels=ss0, profile=1,
                         features=syn.SF["CHAINED"] | syn.SF["FIRRAND"], restart_interval=4)]
    streams = [syn.stream(c, 9300 + 10 * ss0 + i) for i, c in enumerate(cfgs)]
    pcm, infos = T._both(hip, [b for b, _ in streams])
    for (b, f), got, inf in zip(streams, pcm, infos):
        want, r, st = oracle.decode(b, 6, f)
        assert st == 0 and r == f
        assert inf.status & ~hip.ST_BENIGN == 0 and inf.substreams == 2 and inf.pcm_frames == f
        assert np.array_equal(got, want)


@pytest.mark.parametrize("bits", [24, 16])
def test_wav_payload(pkg, oracle, coop, bits):
    T.test_wav_payload_straight_out_of_the_decode(pkg, oracle, bits)


def test_the_library_picks_it_for_small_batches_and_not_for_large_ones(pkg, oracle):
    """lanes 0: a batch of a few segments is decoded by the cooperative kernel, one past the device-side threshold
    (4 096 segments) by the lane kernels -- the same PCM either way, nothing said by the caller."""
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=1, n_aus=16)
    small = [syn.stream(cfg, 9000 + i) for i in range(6)]
    pcm, infos = hip.decode_streams([b for b, _ in small])
    for (b, f), got, inf in zip(small, pcm, infos):
        want, r, st = oracle.decode(b, 6, f)
        assert st == 0 and inf.status == 0 and np.array_equal(got, want)
    # 4 200 single-unit streams = 4 200 segments: past the threshold
    cfg1 = syn.make_cfg(assignment=1, rate_code=0, n_substreams=1, n_aus=1)
    flat, offs, sizes, frames = syn.batch(cfg1, 77, 4200)
    streams = [flat[int(o):int(o + s)] for o, s in zip(offs, sizes)]
    pcm, infos = hip.decode_streams(streams)
    for i in range(0, 4200, 97):
        want, r, st = oracle.decode(streams[i], 2, int(frames[i]))
        assert st == 0 and infos[i].status == 0 and np.array_equal(pcm[i], want)


def padded_unit(b, which, new_size):
    """stream `b` with access unit `which` grown to `new_size` bytes by zero bytes behind its last substream (the
    reference ignores what follows the last substream of a frame, src/mlp.c:463-468: the size field says where the
    next frame starts)"""
    pos = 0
    for _ in range(which):
        pos += 2 * (((int(b[pos]) & 0x0F) << 8) | int(b[pos + 1]))
    size = 2 * (((int(b[pos]) & 0x0F) << 8) | int(b[pos + 1]))
    assert new_size % 2 == 0 and size < new_size <= 8190
    out = np.concatenate([b[:pos + size], np.zeros(new_size - size, np.uint8), b[pos + size:]])
    w = new_size // 2
    out[pos] = (int(b[pos]) & 0xF0) | (w >> 8)
    out[pos + 1] = w & 0xFF
    return out


@pytest.mark.parametrize("S", [1, 2])
def test_an_access_unit_larger_than_the_stage(pkg, oracle, coop, S):
    """The kernel stages an access unit of up to 4 KB in LDS; the 12-bit size field allows 8 190 bytes.  Such a unit
    (here: padded behind its last substream) sends the stream to the sequential pass: same PCM, DVDA_ST_SEQ set."""
    syn, hip = pkg.synth, pkg.hipdec
    cfg = syn.make_cfg(assignment=12, rate_code=1, n_substreams=S, n_aus=24)
    streams = []
    for i, (which, size) in enumerate([(5, 4200), (0, 6000), (23, 8190), (9, 4090)]):
        b, f = syn.stream(cfg, 4400 + i)
        streams.append((padded_unit(b, which, size), f, size))
    pcm, infos = hip.decode_streams([b for b, _, _ in streams])
    for (b, f, size), got, inf in zip(streams, pcm, infos):
        want, r, st = oracle.decode(b, 6, f)
        assert st == 0 and r == f
        assert inf.status & ~hip.ST_BENIGN == 0 and inf.pcm_frames == f
        assert bool(inf.status & hip.ST["SEQ"]) == (size > 4096)
        assert np.array_equal(got, want)
