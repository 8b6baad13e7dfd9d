/* integration/mlp_hip_shim.c -- the reference-side binding: a drop-in for the reference's
 * src/mlp.c behind its own src/mlp.h (open / decode_packet / close, src/mlp.h:29-42), forwarding
 * to tier B of include/dvda_mlp_hip.h.  This is OUR file; it is compiled against the reference's
 * headers only where the reference tree exists (oracle/Makefile target `ref_tools`), to prove
 * that the reference's own dvd-audio.c / dvda2wav.c link and run unchanged on the HIP decoder.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mlp.h"            /* the reference's header (-I$(REF)/src) */
#include "dvda_mlp_hip.h"   /* -I include */

struct MLPDecoder_s {
    dvda_hip_mlpdecoder *h;
};

MLPDecoder *
dvda_open_mlpdecoder(const struct stream_parameters *p)
{
    MLPDecoder *d = malloc(sizeof(*d));
    d->h = dvda_hip_open_mlpdecoder(p->group_0_bps, p->group_1_bps, p->group_0_rate, p->group_1_rate,
                                    p->channel_assignment, 0);
    if (!d->h) {
        fprintf(stderr, "dvda_open_mlpdecoder: no HIP device (the MLP decoder has no CPU path)\n");
        abort();
    }
    return d;
}

void
dvda_close_mlpdecoder(MLPDecoder *d)
{
    dvda_hip_close_mlpdecoder(d->h);
    free(d);
}

unsigned
dvda_mlpdecoder_decode_packet(MLPDecoder *d, BitstreamReader *packet_reader, aa_int *samples)
{
    /* the reference enqueues every remaining byte of the reader (src/mlp.c:349-351) */
    const unsigned n = packet_reader->size(packet_reader);
    uint8_t *buf = malloc(n ? n : 1);
    const int32_t *planar[6];
    unsigned channels = 0, frames, c;

    packet_reader->read_bytes(packet_reader, buf, n);
    frames = dvda_hip_mlpdecoder_decode_packet(d->h, buf, n, planar, &channels);
    free(buf);
    if (dvda_hip_mlpdecoder_status(d->h) & ~DVDA_ST_BENIGN) {
        fprintf(stderr, "MLP decode error, status %#x\n", dvda_hip_mlpdecoder_status(d->h));
        abort();            /* the reference assert()s on the same streams */
    }
    for (c = 0; c < channels && c < samples->len; c++) {
        a_int *ch = samples->_[c];   /* appended to, never reset (src/mlp.c:527-533) */
        ch->resize_for(ch, frames);
        memcpy(ch->_ + ch->len, planar[c], frames * sizeof(int));
        ch->len += frames;
    }
    return frames;
}
