#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz in the DEV CONTAINER (needs /root/reference compiled into
oracle/_ref by `make -C oracle ref`).

Each fixture = the bytes of one small synthetic MLP stream + the planar int32 PCM that the
REAL reference decoder (tuffy/libdvd-audio src/mlp.c through its mlp.h entry points, fed in
2013-byte chunks like PES payloads) produced for it.  Fixtures are data only: inputs and
expected outputs.  The generator parameters are recorded so a fixture can be re-derived.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import libdvd_audio_amd as pkg  # noqa: E402
from tests import oracle_lib  # noqa: E402

syn = pkg.synth
SF = syn.SF

# name, assignment, rate_code, substreams, n_aus, profile, features, seed, restart_interval
FIXTURES = [
    ("recipe_6ch_96k", 12, 1, 1, 24, 0, 0, 1, 8),
    ("recipe_6ch_96k_2ss", 12, 1, 2, 24, 0, 0, 2, 8),
    ("recipe_2ch_96k", 1, 1, 1, 40, 0, 0, 3, 8),
    ("recipe_6ch_192k", 12, 2, 1, 12, 0, 0, 4, 4),
    ("recipe_mono_48k", 0, 0, 1, 40, 0, 0, 5, 8),
    ("recipe_5ch_0x12_44k", 0x12, 8, 1, 24, 0, 0, 6, 8),
    ("recipe_6ch_0x14_2ss", 0x14, 1, 2, 16, 0, 0, 7, 8),
    ("fuzz_fast_6ch", 12, 1, 1, 24, 1, syn.SF_FAST, 11, 5),
    ("fuzz_fast_6ch_2ss", 12, 1, 2, 24, 1, syn.SF_FAST, 12, 3),
    ("fuzz_iir_state", 12, 1, 1, 16, 1, SF["IIR"] | SF["FIRRAND"] | SF["PARAMBLOCKS"], 13, 4),
    ("fuzz_all_6ch", 12, 1, 1, 24, 1, syn.SF_ALL, 14, 6),
    ("fuzz_all_2ch_2ss", 1, 2, 2, 16, 1, syn.SF_ALL, 15, 4),
    ("fuzz_chained", 12, 1, 1, 24, 1, SF["CHAINED"] | SF["FIRRAND"], 16, 4),
    ("fuzz_midframe", 12, 1, 1, 16, 1, SF["MIDMATRIX"] | SF["PARAMBLOCKS"] | SF["MATRIXRAND"] |
     SF["QSS"] | SF["OUTSHIFT"] | SF["VARBLOCK"], 17, 4),
    ("fuzz_varrows", 6, 0, 1, 24, 1, SF["VARROWS"] | SF["VARBLOCK"], 18, 8),
    # round 4: the other rate / bit-depth codes (88.2 kHz 20-bit, 176.4 kHz 16-bit); two substreams whose
    # checkdata_present flags disagree (src/mlp.c:545: substream 0's counts for both), with and without check bytes;
    # a chained two-substream title with parameters on every block (what an encoder writes)
    ("recipe_6ch_88k_20bit", 12, 9, 1, 16, 0, 0, 23, 8, dict(bps_code=1)),
    ("fuzz_2ch_176k_16bit", 1, 10, 1, 12, 1, syn.SF_FAST, 24, 4, dict(bps_code=0)),
    ("check_flags_disagree_2ss", 12, 1, 2, 24, 1, SF["CHECKQUIRK"] | SF["NOCHECK"] | SF["FIRRAND"] | SF["PARAMBLOCKS"], 25, 4),
    ("disc_profile_2ss", 12, 1, 2, 24, 1, SF["DISC"] | SF["CHAINED"] | SF["FIRRAND"] | SF["MIXBOOKS"], 26, 8),
    # round 6: major syncs in front of access units that carry NO restart header (src/mlp.c:449-460: the sync's
    # parameters are compared and the decode goes on with the state it has; decode_block src/mlp.c:748-753)
    ("sync_only_6ch", 12, 1, 1, 32, 1, SF["SYNCONLY"] | SF["FIRRAND"] | SF["PARAMBLOCKS"] | SF["IIR"], 27, 8),
    ("sync_only_chained_2ss", 12, 1, 2, 32, 1, SF["SYNCONLY"] | SF["CHAINED"] | SF["FIRRAND"] | SF["MIXBOOKS"], 28, 5),
]


# Streams in which later major syncs announce OTHER stream parameters (reference src/mlp.c:449-460: that
# access unit is dropped, restart header and all, and decoding goes on with the state the decoder has):
# name -> (indices of the major syncs to change, new group-1 bps code, new channel assignment or None).
# The expected PCM is whatever the real reference makes of it -- 80 PCM frames fewer per dropped unit.
SYNC_CHANGES = {
    "sync_change_6ch": ((2, 4, 5), 1, None),
    "sync_change_2ch_2ss": ((1, 2, 3), 0, 3),
    # (round 5) eighty changed major syncs in a row -- more than the 64 the index's walk used to take
    "sync_change_80_in_a_row": (tuple(range(5, 85)), 0, None),
}
FIXTURES += [
    ("sync_change_6ch", 12, 1, 1, 28, 0, 0, 21, 4),
    ("sync_change_2ch_2ss", 1, 1, 2, 30, 0, 0, 22, 5),
    ("sync_change_80_in_a_row", 1, 0, 1, 100, 0, 0, 23, 1),
]


def main():
    assert oracle_lib.Reference.available(), "build oracle/_ref first: make -C oracle ref"
    from tests import stream_tools
    ref = oracle_lib.Reference()
    here = os.path.dirname(os.path.abspath(__file__))
    only_missing = "--all" not in sys.argv
    for name, asg, rate, S, naus, prof, feat, seed, ri, *more in FIXTURES:
        if only_missing and os.path.exists(os.path.join(here, name + ".npz")):
            continue
        cfg = syn.make_cfg(assignment=asg, rate_code=rate, n_substreams=S, n_aus=naus, profile=prof,
                           features=feat, restart_interval=ri, **(more[0] if more else {}))
        data, frames = syn.stream(cfg, seed)
        if name in SYNC_CHANGES:
            which, bps, new_asg = SYNC_CHANGES[name]
            data, _ = stream_tools.change_sync_params(data, which, g1_bps=bps, assignment=new_asg)
            frames -= len(which) * syn.rows_per_au(rate)
        pcm, r = ref.decode(data, asg, rate, cfg.bps_code, frames + 400, chunk=2013)
        assert r == frames, (name, r, frames)
        np.savez_compressed(os.path.join(here, name + ".npz"), mlp=data, pcm=pcm,
                            meta=np.array([asg, rate, S, naus, prof, feat, seed, ri, cfg.bps_code],
                                          np.int64))
        print("%-24s %6d bytes -> %s PCM" % (name, len(data), pcm.shape))


if __name__ == "__main__":
    main()
