/* mlp_step.h -- between mlp_stream.c (tier B, the mlp.h mirror: plain C) and mlp_hip.hip: a decoder whose state stays
 * on the device from one packet to the next.  Internal to the library (tier B is the public face:
 * include/dvda_mlp_hip.h, dvda_hip_mlpdecoder_decode_packet).
 *
 * One step = the whole access units of one packet: bytes up (one copy from pinned memory), parity / CRC-8
 * (k_sync_mask's partial sums + k_au_check, mlp_check.h), ONE workgroup of k_coop<false, true> (mlp_coop.h) that
 * takes the decoder state from the device record (reference struct substream + filter histories, src/mlp.c:103-115,
 * 297-304), decodes the units and puts the state back, PCM and a result record down (one copy).  Nothing in front of
 * the packet is decoded again. */
#ifndef DVDA_MLP_STEP_H
#define DVDA_MLP_STEP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define DVDA_STEP_MAX_BYTES 49152u      /* one step's access units: at most this many bytes ... */
#define DVDA_STEP_MAX_UNITS 48u         /* ... and this many units (a caller with more steps more than once) */

typedef struct dvda_mlp_hip_stepper dvda_mlp_hip_stepper;

typedef struct {
    uint32_t status;        /* DVDA_ST_* of this step's access units */
    uint32_t frames_out;    /* access units that yielded PCM (a unit with a foreign major sync yields none) */
    uint32_t rows_written;  /* PCM frames per channel */
    uint32_t sync_seen;     /* 1 + index of the step's last unit that carries the stream's own major sync, 0: none */
    int32_t fir[2][48];     /* FIR histories in front of that unit, [substream][slot * 8 + tap] */
} dvda_mlp_step_result;

int dvda_mlp_hip_stepper_create(dvda_mlp_hip_stepper **out, int device);
void dvda_mlp_hip_stepper_destroy(dvda_mlp_hip_stepper *s);
/* bytes[0, len): n_units whole access units (len even, <= DVDA_STEP_MAX_BYTES; n_units <= DVDA_STEP_MAX_UNITS);
 * packed_sync: the stream's latched major sync (g0 bps | g1 bps << 4 | g0 rate << 8 | g1 rate << 12 | assignment << 16 |
 * substreams << 24); fresh != 0: the first step of a decoder (no state yet).
 * -> *res (host, valid until the next step), *pcm planar int32 [channel][*stride] in RIFF order (host, pinned),
 * *channels.  Returns DVDA_HIP_OK or a DVDA_HIP_E* code; decode conditions are in res->status. */
int dvda_mlp_hip_stepper_step(dvda_mlp_hip_stepper *s, const uint8_t *bytes, size_t len, uint32_t n_units,
                              uint32_t packed_sync, int fresh, const dvda_mlp_step_result **res, const int32_t **pcm,
                              uint64_t *stride, unsigned *channels);

#ifdef __cplusplus
}
#endif
#endif
