/* oracle/pcm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's PCM path for SURVEY.md 8(f-2): the AOB byte un-swizzle of
 * src/pcm.c:99-170 (tables :103-139, converters :172-193) and the sector walk that feeds it
 * (pack header src/packet.c:138-188, PES packets :61-117, audio packet header
 * src/dvd-audio.c:1238-1248, parameter block src/pcm.c:80-97, packet loop
 * src/dvd-audio.c:1016-1084).  Pinned against the compiled reference (oracle/ref_driver.c:
 * ref_pcm_decode) in tests/test_pcm.py.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* aob byte i of a 2-frame chunk is byte SWAP[i] of the little-endian frame-major sample block */
static const uint8_t SWAP16[6][24] = {
    {1, 0, 3, 2},
    {1, 0, 3, 2, 5, 4, 7, 6},
    {1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10},
    {1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 13, 12, 15, 14},
    {1, 0, 3, 2, 5, 4, 7, 6, 9, 8, 11, 10, 13, 12, 15, 14, 17, 16, 19, 18},
    {5, 4, 7, 6, 17, 16, 19, 18, 1, 0, 3, 2, 9, 8, 11, 10, 13, 12, 15, 14, 21, 20, 23, 22}};
static const uint8_t SWAP24[6][36] = {
    {2, 1, 5, 4, 0, 3},
    {2, 1, 5, 4, 8, 7, 11, 10, 0, 3, 6, 9},
    {8, 7, 17, 16, 6, 15, 2, 1, 5, 4, 11, 10, 14, 13, 0, 3, 9, 12},
    {8, 7, 11, 10, 20, 19, 23, 22, 6, 9, 18, 21, 2, 1, 5, 4, 14, 13, 17, 16, 0, 3, 12, 15},
    {8, 7, 11, 10, 14, 13, 23, 22, 26, 25, 29, 28, 6, 9, 12, 21, 24, 27, 2, 1, 5, 4, 17, 16, 20, 19,
     0, 3, 15, 18},
    {8, 7, 11, 10, 26, 25, 29, 28, 6, 9, 24, 27, 2, 1, 5, 4, 14, 13, 17, 16, 20, 19, 23, 22,
     32, 31, 35, 34, 0, 3, 12, 15, 18, 21, 30, 33}};

/* src/pcm.c:99-170.  payload = the packet bytes behind the parameter block; whole chunks only.
 * out is planar: channel c at out[c * stride + first_frame ...].  Returns PCM frames. */
long
pcm_oracle_unswizzle(const uint8_t *payload, size_t len, unsigned bps, unsigned channels,
                     int32_t *out, size_t stride, size_t first_frame)
{
    const unsigned nb = bps / 8;
    const unsigned chunk = nb * channels * 2;
    const uint8_t *swap = (bps == 16) ? SWAP16[channels - 1] : SWAP24[channels - 1];
    long frames = 0;
    if ((bps != 16 && bps != 24) || channels < 1 || channels > 6)
        return -1;
    while (len >= chunk) {
        uint8_t un[36];
        unsigned i;
        for (i = 0; i < chunk; i++)
            un[swap[i]] = payload[i];
        for (i = 0; i < channels * 2; i++) {
            const uint8_t *s = un + i * nb;
            int32_t v;
            if (nb == 2)
                v = (int16_t)(uint16_t)(s[0] | (s[1] << 8));
            else
                v = (int32_t)((uint32_t)(s[0] | (s[1] << 8) | (s[2] << 16)) << 8) >> 8;
            out[(size_t)(i % channels) * stride + first_frame + frames + i / channels] = v;
        }
        payload += chunk;
        len -= chunk;
        frames += 2;
    }
    return frames;
}

/* Walks 2048-byte sectors the way packet_reader_next_audio_packet + decode_pcm_audio do and
 * un-swizzles every PCM packet.  Returns PCM frames, or -1 on a malformed sector. */
long
pcm_oracle_decode_sectors(const uint8_t *sectors, size_t n_sectors, unsigned bps, unsigned channels,
                          int32_t *out, size_t stride)
{
    long frames = 0;
    size_t s;
    for (s = 0; s < n_sectors; s++) {
        const uint8_t *p = sectors + s * 2048;
        size_t pos;
        if (p[0] != 0 || p[1] != 0 || p[2] != 1 || p[3] != 0xBA)
            return -1;
        if ((p[4] >> 6) != 1 || !(p[4] & 4) || !(p[6] & 4) || !(p[8] & 4) || !(p[9] & 1) ||
            (p[12] & 3) != 3)
            return -1;                                  /* marker bits, src/packet.c:172-176 */
        pos = 14 + (p[13] & 7);
        while (pos + 6 <= 2048) {
            const unsigned id = p[pos + 3], plen = (p[pos + 4] << 8) | p[pos + 5];
            if (p[pos] != 0 || p[pos + 1] != 0 || p[pos + 2] != 1)
                return -1;
            if (pos + 6 + plen > 2048)
                return -1;
            if (id == 0xBD) {
                const uint8_t *q = p + pos + 6;
                unsigned pad1, codec, pad2;
                size_t hdr;
                long f;
                if (plen < 7)
                    return -1;
                pad1 = q[2];
                if (plen < 7u + pad1)
                    return -1;
                codec = q[3 + pad1];
                pad2 = q[6 + pad1];
                hdr = 7 + pad1 + pad2;               /* parameters (9 bytes) are inside pad_2 */
                if (codec != 0xA0 || pad2 < 9 || hdr > plen)
                    return -1;
                f = pcm_oracle_unswizzle(q + hdr, plen - hdr, bps, channels, out, stride, (size_t)frames);
                if (f < 0)
                    return -1;
                frames += f;
            }
            pos += 6 + plen;
        }
    }
    return frames;
}

/* ---- planar int32 -> interleaved little-endian WAV payload.
 * dvda_read interleaves frame-major in RIFF channel order (src/dvd-audio.c:781-792); dvda2wav
 * writes every value with write_signed(bits) (utils/dvda2wav.c:326-334), which emits the low
 * bits-1 bits and then a sign bit taken from v < 0 (src/bitstream.c:2846-2857) -- not plain
 * truncation for out-of-range positives. */
long
wav_oracle_pack(const int32_t *planar, size_t stride, unsigned channels, size_t frames, unsigned bits,
                uint8_t *out)
{
    const unsigned nb = bits / 8;
    const uint32_t low = (1u << (bits - 1)) - 1u;
    size_t f;
    unsigned c, b;
    if ((bits != 16 && bits != 24) || channels < 1 || channels > 6)
        return -1;
    for (f = 0; f < frames; f++)
        for (c = 0; c < channels; c++) {
            const int32_t v = planar[(size_t)c * stride + f];
            const uint32_t u = ((uint32_t)v & low) | (v < 0 ? (1u << (bits - 1)) : 0u);
            for (b = 0; b < nb; b++)
                *out++ = (uint8_t)(u >> (8 * b));
        }
    return (long)(frames * channels * nb);
}
