/* oracle/ref_driver.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A thin driver (our code) around the REAL reference decoder, compiled by
 * oracle/Makefile together with the reference's own sources into
 * oracle/_ref/libdvda_ref.so.  It feeds a raw MLP byte stream to the
 * reference's mlp.h entry points exactly as the reference's track reader does
 * (reference src/dvd-audio.c:1127-1140, 1181-1184, 1212-1215): one
 * br_open_buffer() reader per "packet", dvda_mlpdecoder_decode_packet(), close.
 *
 * Nothing here is part of the product; the product never links this file.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "mlp.h"   /* reference src/mlp.h:29-42 (pulled in via -I$(REF)/src) */

/* Decodes `len` bytes of MLP data in chunks of `chunk` bytes (0 = one chunk),
 * mimicking PES-payload sized feeding.  Output: planar int32, channel c at
 * out[c * cap .. c * cap + frames), RIFF-WAVE channel order (that is what the
 * reference appends into `samples`, src/mlp.c:527-533).
 * Returns the number of PCM frames decoded, or -1 if `cap` was too small. */
long
ref_mlp_decode(const uint8_t *data, size_t len, size_t chunk,
               unsigned g0_bps, unsigned g1_bps,
               unsigned g0_rate, unsigned g1_rate,
               unsigned channel_assignment,
               unsigned nch, int32_t *out, size_t cap)
{
    struct stream_parameters p;
    MLPDecoder *dec;
    aa_int *samples = aa_int_new();
    unsigned c;
    size_t off = 0;
    long total = 0;

    p.group_0_bps = g0_bps;
    p.group_1_bps = g1_bps;
    p.group_0_rate = g0_rate;
    p.group_1_rate = g1_rate;
    p.channel_assignment = channel_assignment;

    for (c = 0; c < nch; c++)
        (void)samples->append(samples);

    dec = dvda_open_mlpdecoder(&p);
    if (chunk == 0)
        chunk = len ? len : 1;

    while (off < len) {
        size_t n = (len - off < chunk) ? (len - off) : chunk;
        BitstreamReader *r = br_open_buffer(data + off, (unsigned)n,
                                            BS_BIG_ENDIAN);
        total += dvda_mlpdecoder_decode_packet(dec, r, samples);
        r->close(r);
        off += n;
    }

    for (c = 0; c < nch; c++) {
        const a_int *ch = samples->_[c];
        if (ch->len > cap) {
            total = -1;
            break;
        }
        if (ch->len)
            memcpy(out + (size_t)c * cap, ch->_, sizeof(int32_t) * ch->len);
    }

    dvda_close_mlpdecoder(dec);
    samples->del(samples);
    return total;
}

/* per-channel decoded length is not necessarily `frames` for malformed
 * channel layouts; expose it for tests that care. */
long
ref_mlp_decode_lens(const uint8_t *data, size_t len, size_t chunk,
                    unsigned g0_bps, unsigned g1_bps,
                    unsigned g0_rate, unsigned g1_rate,
                    unsigned channel_assignment,
                    unsigned nch, unsigned *lens)
{
    struct stream_parameters p;
    MLPDecoder *dec;
    aa_int *samples = aa_int_new();
    unsigned c;
    size_t off = 0;
    long total = 0;

    p.group_0_bps = g0_bps;
    p.group_1_bps = g1_bps;
    p.group_0_rate = g0_rate;
    p.group_1_rate = g1_rate;
    p.channel_assignment = channel_assignment;
    for (c = 0; c < nch; c++)
        (void)samples->append(samples);
    dec = dvda_open_mlpdecoder(&p);
    if (chunk == 0)
        chunk = len ? len : 1;
    while (off < len) {
        size_t n = (len - off < chunk) ? (len - off) : chunk;
        BitstreamReader *r = br_open_buffer(data + off, (unsigned)n,
                                            BS_BIG_ENDIAN);
        total += dvda_mlpdecoder_decode_packet(dec, r, samples);
        r->close(r);
        off += n;
    }
    for (c = 0; c < nch; c++)
        lens[c] = samples->_[c]->len;
    dvda_close_mlpdecoder(dec);
    samples->del(samples);
    return total;
}

/* ---- PCM: the reference's un-swizzle, src/pcm.c:99-170 */
#include "pcm.h"

/* Decodes one packet payload (the bytes behind the 9-byte parameter block) with the reference's
 * PCM decoder.  out is planar [channels][cap].  Returns PCM frames. */
long
ref_pcm_decode(const uint8_t *payload, size_t len, unsigned bits_per_sample, unsigned channels,
               int32_t *out, size_t cap)
{
    PCMDecoder *dec = dvda_open_pcmdecoder(bits_per_sample, channels);
    aa_int *samples = aa_int_new();
    BitstreamReader *r = br_open_buffer(payload, (unsigned)len, BS_BIG_ENDIAN);
    unsigned c;
    long frames;
    for (c = 0; c < channels; c++)
        (void)samples->append(samples);
    frames = dvda_pcmdecoder_decode_packet(dec, r, samples);
    r->close(r);
    for (c = 0; c < channels; c++) {
        if (samples->_[c]->len > cap) {
            frames = -1;
            break;
        }
        memcpy(out + (size_t)c * cap, samples->_[c]->_, sizeof(int32_t) * samples->_[c]->len);
    }
    dvda_close_pcmdecoder(dec);
    samples->del(samples);
    return frames;
}

/* ---- WAV payload: dvda_read's interleave + dvda2wav's write_signed
 *      (src/dvd-audio.c:781-792, utils/dvda2wav.c:326-334, src/bitstream.c:2846-2857) */
long
ref_wav_pack(const int *interleaved, size_t n, unsigned bits, uint8_t *out)
{
    BitstreamRecorder *r = bw_open_recorder(BS_LITTLE_ENDIAN);
    BitstreamWriter *w = (BitstreamWriter *)r;
    size_t i;
    long bytes;
    for (i = 0; i < n; i++)
        w->write_signed(w, bits, interleaved[i]);
    bytes = (long)r->bytes_written(r);
    memcpy(out, r->data(r), (size_t)bytes);
    r->close(r);
    return bytes;
}
