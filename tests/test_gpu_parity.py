"""GPU parity: HIP decode path (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _check(pkg, oracle, cfgs_seeds, lanes=2):
    syn, hip = pkg.synth, pkg.hipdec
    streams, frames, cfgs = [], [], []
    for cfg, seed in cfgs_seeds:
        b, f = syn.stream(cfg, seed)
        streams.append(b)
        frames.append(f)
        cfgs.append(cfg)
    pcm, infos = hip.decode_streams(streams, lanes_per_segment=lanes)
    for i, (b, f, cfg) in enumerate(zip(streams, frames, cfgs)):
        nch = syn.channels(cfg.assignment)
        want, r, st = oracle.decode(b, nch, f)
        assert st == 0 and r == f
        inf = infos[i]
        assert inf.status == 0, "stream %d status %#x" % (i, inf.status)
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


@pytest.mark.parametrize("rate", [0, 1, 2])
@pytest.mark.parametrize("assignment,S", [(12, 1), (12, 2), (1, 1), (0, 1), (0x14, 2), (0x12, 1), (6, 2)])
def test_fuzz_fast_features(pkg, oracle, assignment, S, rate):
    syn = pkg.synth
    cases = []
    for seed in range(4):
        cfg = syn.make_cfg(assignment=assignment, rate_code=rate, n_substreams=S, n_aus=24, profile=1,
                           features=syn.SF_FAST, restart_interval=[8, 3, 16, 5][seed])
        cases.append((cfg, 100 + seed))
    _check(pkg, oracle, cases, lanes=2)
