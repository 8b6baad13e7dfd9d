"""The chain passes have two forms (csrc/mlp_chain.h: k_chain_fused, one walk over the planes by two-wave workgroups, what
a batch with thousands of chains gets; csrc/mlp_chain_small.h: the recursion in place + a parallel rematrix pass, what a
small batch gets).  The library picks by batch size, so the small parity cases would only ever see the second: here the
cases that defer segments to the chain passes run again with EACH form forced (dvda_mlp_hip_set_chain_form), against the
same oracle."""
import pytest

from tests import test_gpu_parity as T

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 2], ids=["fused", "two_pass"])
def form(pkg, request):
    hip = pkg.hipdec
    hip.CHAIN_FORM = request.param
    yield request.param
    hip.CHAIN_FORM = 0


@pytest.mark.parametrize("feature", ["CHAINED", "MIDMATRIX", "MIDRESTART", "VARROWS"])
@pytest.mark.parametrize("S", [1, 2])
def test_deferred_features(pkg, oracle, form, feature, S):
    T.test_deferred_features_general_pass(pkg, oracle, feature, S)


def test_golden_vectors(pkg, form):
    T.test_golden_vectors_on_gpu(pkg)


@pytest.mark.parametrize("S", [1, 2])
def test_fuzz_all_features(pkg, oracle, form, S):
    T.test_fuzz_all_features(pkg, oracle, S)


@pytest.mark.parametrize("lanes", [0, 2, 64])
@pytest.mark.parametrize("S", [1, 2])
def test_disc_profile_streams(pkg, oracle, form, S, lanes):
    T.test_disc_profile_streams(pkg, oracle, S, lanes)


def test_thousands_of_chains(pkg, oracle, form):
    T.test_thousands_of_chains_and_midframe_segments_in_one_batch(pkg, oracle)


@pytest.mark.parametrize("bits", [16, 24])
def test_wav_payload(pkg, oracle, form, bits):
    T.test_wav_payload_straight_out_of_the_decode(pkg, oracle, bits)


@pytest.mark.parametrize("ss0", [1, 2, 3, 4, 5])
def test_two_substreams_of_any_split(pkg, oracle, form, ss0):
    T.test_two_substreams_of_any_split(pkg, oracle, ss0)
