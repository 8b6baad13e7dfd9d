"""Helpers for tests that edit synthetic MLP byte streams (framing per reference src/mlp.c:384-405, 614-654)."""
import numpy as np


def frame_offsets(stream):
    """byte offset of every complete access unit (12-bit size chain)"""
    out, pos, n = [], 0, len(stream)
    while pos + 4 <= n:
        size = 2 * (((int(stream[pos]) & 0xF) << 8) | int(stream[pos + 1]))
        if size < 4 or pos + size > n:
            break
        out.append(pos)
        pos += size
    return out


def is_major_sync(stream, pos):
    return (len(stream) >= pos + 32 and bytes(stream[pos + 4:pos + 8]) == b"\xF8\x72\x6F\xBB" and
            (int(stream[pos + 20]) >> 4) in (1, 2))


def change_sync_params(stream, which, g1_bps=None, assignment=None):
    """Returns a copy of `stream` whose `which`-th major syncs (indices into the list of major-sync access
    units, 0 = the stream's first) announce other stream parameters: group-1 bits-per-sample code and / or
    channel assignment -- two of the five fields reference dvda_params_equal compares (src/mlp.c:449-460).
    Nothing else changes; the major sync is covered by no parity / CRC byte the reference checks."""
    out = np.array(stream, np.uint8, copy=True)
    syncs = [p for p in frame_offsets(out) if is_major_sync(out, p)]
    for w in which:
        p = syncs[w]
        if g1_bps is not None:
            out[p + 8] = (int(out[p + 8]) & 0xF0) | (g1_bps & 0xF)
        if assignment is not None:
            out[p + 11] = (int(out[p + 11]) & 0xE0) | (assignment & 0x1F)
    return out, [syncs[w] for w in which]
