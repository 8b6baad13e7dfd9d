"""Synthetic DVD-Audio disc image writer (tooling for the end-to-end tests).

Writes the three files the reference's readers need for one titleset:

    AUDIO_TS/AUDIO_TS.IFO   "DVDAUDIO-AMG", byte 63 = titleset count      (reference src/dvd-audio.c:824-857)
    AUDIO_TS/ATS_01_0.IFO   "DVDAUDIO-ATS", title / track / index tables  (src/dvd-audio.c:860-950)
    AUDIO_TS/ATS_01_1.AOB   2048-byte sectors: pack header + PES packets  (src/packet.c:61-188,
                            audio packet header src/dvd-audio.c:1238-1248)

Each track starts on a sector boundary.  MLP tracks carry raw MLP bytes behind the 0xA1 codec
header; PCM tracks carry 0xA0 packets with the 9-byte parameter block (src/pcm.c:80-97) and
samples swizzled with the inverse of AOB_BYTE_SWAP (src/pcm.c:103-139).
"""
import os
import struct

import numpy as np

SECTOR = 2048
MLP_CODEC, PCM_CODEC = 0xA1, 0xA0
RATES = {0: 48000, 1: 96000, 2: 192000, 8: 44100, 9: 88200, 10: 176400}
BPS = {0: 16, 1: 20, 2: 24}
CHANNELS = [1, 2, 3, 4, 3, 4, 5, 3, 4, 5, 4, 5, 6, 4, 5, 4, 5, 6, 5, 5, 6]

# AOB byte order of one chunk (2 PCM frames): aob byte i is byte AOB_BYTE_SWAP[i] of the
# little-endian, frame-major sample block (reference src/pcm.c:103-139)
AOB_BYTE_SWAP = {
    16: [
        [1, 0, 3, 2],
        [1, 0, 3, 2, 5, 4, 7, 6],
        [1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10],
        [1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 13, 12, 15, 14],
        [1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 13, 12, 15, 14, 17, 16, 19, 18],
        [5, 4, 7, 6, 17, 16, 19, 18, 1, 0, 3, 2, 9, 8, 11, 10, 13, 12, 15, 14, 21, 20, 23, 22],
    ],
    24: [
        [2, 1, 5, 4, 0, 3],
        [2, 1, 5, 4, 8, 7, 11, 10, 0, 3, 6, 9],
        [8, 7, 17, 16, 6, 15, 2, 1, 5, 4, 11, 10, 14, 13, 0, 3, 9, 12],
        [8, 7, 11, 10, 20, 19, 23, 22, 6, 9, 18, 21, 2, 1, 5, 4, 14, 13, 17, 16, 0, 3, 12, 15],
        [8, 7, 11, 10, 14, 13, 23, 22, 26, 25, 29, 28, 6, 9, 12, 21, 24, 27,
         2, 1, 5, 4, 17, 16, 20, 19, 0, 3, 15, 18],
        [8, 7, 11, 10, 26, 25, 29, 28, 6, 9, 24, 27, 2, 1, 5, 4, 14, 13, 17, 16, 20, 19, 23, 22,
         32, 31, 35, 34, 0, 3, 12, 15, 18, 21, 30, 33],
    ],
}


def _pack_header(scr=0):
    # "32u 2u 3u 1u 15u 1u 15u 1u 9u 1u 22u 2u 5p 3u" with the marker bits the reader checks
    bits = 0
    n = 0

    def put(width, value):
        nonlocal bits, n
        bits = (bits << width) | (value & ((1 << width) - 1))
        n += width

    put(32, 0x000001BA)
    put(2, 1)
    put(3, (scr >> 30) & 7)
    put(1, 1)
    put(15, (scr >> 15) & 0x7FFF)
    put(1, 1)
    put(15, scr & 0x7FFF)
    put(1, 1)
    put(9, 0)
    put(1, 1)
    put(22, 0x0189C3)
    put(2, 3)
    put(5, 0)
    put(3, 0)
    assert n == 112
    return bits.to_bytes(14, "big")


def _sector(codec, body, params=b""):
    """One sector holding one audio PES packet: codec header, optional PCM parameter block
    (counted in pad_2), `body` bytes, then filler so that the packets tile the sector."""
    room = SECTOR - 14 - 6 - 7          # payload room behind the 7-byte audio packet header
    extra = room - len(params) - len(body)
    assert extra >= 0
    pad2_fill = 0
    tail = b""
    if extra and extra + len(params) <= 255:
        pad2_fill = extra               # absorbed by pad_2 (skipped by the reader)
        extra = 0
    elif extra:
        assert extra >= 6, "cannot tile the sector"
        tail = b"\x00\x00\x01\xBE" + struct.pack(">H", extra - 6) + b"\xFF" * (extra - 6)
    pad2 = len(params) + pad2_fill
    payload = b"\x81\x00" + b"\x00" + bytes([codec, 0, 0, pad2]) + params + b"\x00" * pad2_fill + body
    pes = b"\x00\x00\x01\xBD" + struct.pack(">H", len(payload)) + payload
    out = _pack_header() + pes + tail
    assert len(out) == SECTOR, len(out)
    return out


def mlp_track_sectors(mlp_bytes):
    data = bytes(np.asarray(mlp_bytes, np.uint8))
    room = SECTOR - 14 - 6 - 7
    out = []
    for off in range(0, len(data), room):
        out.append(_sector(MLP_CODEC, data[off:off + room]))
    return out


def pcm_track_sectors(samples, bps_code, rate_code, assignment):
    """samples: int array [frames, channels] (frames even)."""
    samples = np.asarray(samples, np.int64)
    frames, ch = samples.shape
    assert ch == CHANNELS[assignment] and frames % 2 == 0
    bps = BPS[bps_code]
    assert bps in (16, 24)
    nbytes = bps // 8
    swap = AOB_BYTE_SWAP[bps][ch - 1]
    chunk_size = 2 * ch * nbytes
    params = struct.pack(">HBBBBBBB", 0, 0, (bps_code << 4) | bps_code, (rate_code << 4) | rate_code, 0,
                         assignment, 0, 0)
    assert len(params) == 9
    room = SECTOR - 14 - 6 - 7 - 9
    chunks_per_sector = room // chunk_size
    # little-endian frame-major block -> AOB order
    le = np.zeros((frames * ch, nbytes), np.uint8)
    flat = samples.reshape(-1) & ((1 << bps) - 1)
    for b in range(nbytes):
        le[:, b] = (flat >> (8 * b)) & 0xFF
    le = le.reshape(frames // 2, chunk_size)
    aob = np.zeros_like(le)
    for i, s in enumerate(swap):
        aob[:, i] = le[:, s]
    out = []
    for off in range(0, len(aob), chunks_per_sector):
        body = aob[off:off + chunks_per_sector].tobytes()
        out.append(_sector(PCM_CODEC, body, params))
    return out


def write_disc(root, tracks):
    """tracks: list of dicts {"sectors": [bytes...], "pcm_frames": n, "rate_code": r}: one title.
    Returns the AUDIO_TS path."""
    return write_disc_titles(root, [tracks])


def split_tracks(sectors, cuts, pcm_frames, rate_code):
    """One continuous run of sectors cut into consecutive tracks at the sector indices `cuts`
    (tracks of a real title share one MLP stream: a track's last frames usually sit in the first
    sector of the next one).  pcm_frames: per-track PCM frame counts (PTS lengths; only raw-PCM
    tracks use them)."""
    edges = [0] + list(cuts) + [len(sectors)]
    return [{"sectors": sectors[a:b], "pcm_frames": f, "rate_code": rate_code}
            for a, b, f in zip(edges[:-1], edges[1:], pcm_frames)]


def write_disc_titles(root, titles, titlesets=1, titleset=1):
    """titles: list of titles, each a list of track dicts (see write_disc); all of them go into
    title set `titleset` (ATS_XX_0.IFO + ATS_XX_1.AOB) in order.  Returns the AUDIO_TS path."""
    ats = os.path.join(root, "AUDIO_TS")
    os.makedirs(ats, exist_ok=True)
    amg = bytearray(SECTOR)
    amg[0:12] = b"DVDAUDIO-AMG"
    amg[63] = titlesets
    open(os.path.join(ats, "AUDIO_TS.IFO"), "wb").write(amg)

    ifo = bytearray(4 * SECTOR)
    ifo[0:12] = b"DVDAUDIO-ATS"
    base = SECTOR
    struct.pack_into(">HHI", ifo, base, len(titles), 0, 0)       # title_count
    table_off = 0x100
    pos = 0
    for ti, tracks in enumerate(titles):
        n = len(tracks)
        struct.pack_into(">BBHI", ifo, base + 8 + 8 * ti, ti + 1, 0, 0, table_off)
        t0 = base + table_off
        pts = [int(round(t["pcm_frames"] * 90000.0 / RATES[t["rate_code"]])) for t in tracks]
        sector_ptr_off = 16 + 20 * n
        struct.pack_into(">HBBIIHH", ifo, t0, 0, n, n, sum(pts), 0, sector_ptr_off, 0)
        acc = 0
        for i in range(n):
            struct.pack_into(">IBBII", ifo, t0 + 16 + 20 * i, 0, i + 1, 0, acc, pts[i])
            acc += pts[i]
        for i in range(n):
            first = pos
            pos += len(tracks[i]["sectors"])
            struct.pack_into(">III", ifo, t0 + sector_ptr_off + 12 * i, i + 1, first, pos - 1)
        table_off += (sector_ptr_off + 12 * n + 15) & ~15
    assert base + table_off <= len(ifo)
    open(os.path.join(ats, "ATS_%02d_0.IFO" % titleset), "wb").write(ifo)
    with open(os.path.join(ats, "ATS_%02d_1.AOB" % titleset), "wb") as f:
        for tracks in titles:
            for t in tracks:
                for sec in t["sectors"]:
                    f.write(sec)
    return ats
