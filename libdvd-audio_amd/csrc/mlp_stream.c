/* mlp_stream.c -- tier B of include/dvda_mlp_hip.h: the mlp.h mirror, host side, plain C.
 *
 * Mirrors reference src/mlp.c:265-382 (open / close / decode_packet + the byte queue and the
 * "decode while a whole frame is queued" loop) on top of the batch tier.  The reference carries
 * decoder state from packet to packet inside MLPDecoder; here the state that can cross a call
 * is reduced to what the format really needs by cutting only at major syncs:
 *
 *   - the queue keeps every byte from the last major-sync access unit onward (round 6: the last one whose substreams
 *     all open with a restart header -- unit_restarts(); a sync need not restart anything, src/mlp.c:449-460), so each call
 *     re-decodes at most one restart segment it has seen before (parameters, noise seed, IIR
 *     history are all re-established by that segment's restart header, src/mlp.c:867-990);
 *   - the FIR history -- the one thing the reference never resets (src/mlp.c:297-304, 1302) --
 *     is fetched from the device at the end of the previous segment and handed back as the
 *     stream's initial history on the next call;
 *   - PCM frames of the retained segment that were already returned are skipped.
 *
 * Round 4: the decoder state itself stays on the device between calls (mlp_step.h; k_coop<false, true>, mlp_coop.h):
 * a call sends its packet's whole access units up, ONE workgroup decodes exactly those from the state the call
 * before left, PCM and a small result record come down -- nothing in front of the packet is decoded again, no
 * index is built.  The queue discipline above stays (bytes from the last major sync on, the FIR history in front
 * of it, frames already returned): a stream the stepping kernel does not take -- an access unit of non-standard
 * length, or one larger than its stage -- falls back, for good, to the batch-tier path that was all of this file
 * until round 3 and decodes from those three things.
 *
 * Every call is one small GPU job: this is the compatibility tier, not the fast path.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dvda_mlp_hip.h"
#include "mlp_step.h"

struct dvda_hip_mlpdecoder {
    int device;
    unsigned params[5];          /* as given at open (the reference stores and ignores them too) */
    dvda_mlp_hip_stepper *step;  /* the decoder state on the device + the one-workgroup decode of a packet's units */
    int stepped;                 /* a step has run: the device holds state */
    int slow;                    /* the stream left what the stepping kernel takes: batch-tier path from here on */
    dvda_mlp_hip_ctx *ctx;       /* batch-tier path (made when first needed) */
    uint32_t ctx_segments;
    /* byte queue: starts at a major-sync access unit once one has been seen */
    uint8_t *q;
    size_t qlen, qcap;
    size_t decoded_end;          /* bytes of q covered by access units already decoded */
    uint64_t rows_before;        /* PCM frames of q's first segment already handed out */
    int have_fir;
    int32_t fir[2 * 48];         /* FIR history at the start of q's first segment */
    int have_sync;
    uint8_t sync_params[4];      /* bytes 8,9,11(low 5 bits) and 20(high nibble) of the first sync */
    unsigned status;
    /* device / host buffers, grown on demand */
    uint8_t *d_bytes;
    size_t d_bytes_cap;
    uint64_t *d_meta;            /* off, len, out_off, out_stride */
    int32_t *d_pcm;
    size_t d_pcm_cap;
    int32_t *d_fir;
    int32_t *h_pcm;
    size_t h_pcm_cap;
};

static unsigned rows_per_au(unsigned rate_code)
{
    switch (rate_code) {
    case 0: case 8: return 40;
    case 1: case 9: return 80;
    case 2: case 10: return 160;
    default: return 0;
    }
}

/* major-sync access unit at q[pos]?  (reference src/mlp.c:621-639) */
static int sync_at(const uint8_t *q, size_t pos, size_t size)
{
    return size >= 32 && q[pos + 4] == 0xF8 && q[pos + 5] == 0x72 && q[pos + 6] == 0x6F &&
           q[pos + 7] == 0xBB && ((q[pos + 20] >> 4) == 1 || (q[pos + 20] >> 4) == 2);
}

/* Does every substream of the sync unit q[pos, pos + size) open with a restart header?  A major sync does not oblige the
 * substreams to restart: the reference compares its parameters and decodes on with the state it has (src/mlp.c:449-460;
 * decode_block src/mlp.c:748-753 reads a restart header where the block's two flags say so).  Only a unit that restarts
 * everything is one a decode can START from -- the queue is cut at those, and k_coop<false, true> (mlp_coop.h) reports
 * the FIR histories in front of exactly those, judged by the same bits.  S: the stream's latched substream count, which
 * is what the directory is read with (src/mlp.c:463-468, 661-667). */
static int unit_restarts(const uint8_t *q, size_t pos, size_t size, unsigned S)
{
    size_t p = 4 + 28, end0 = 0;
    unsigned s;
    if (S != 1 && S != 2)
        return 0;
    for (s = 0; s < S; s++) {
        unsigned e;
        if (p + 2 > size)
            return 0;
        e = ((unsigned)q[pos + p] << 8) | q[pos + p + 1];
        if (s == 0)
            end0 = (size_t)(e & 0xFFFu) * 2;
        p += (e & 0x8000u) ? 4 : 2;
    }
    if (p >= size || (q[pos + p] & 0xC0) != 0xC0)
        return 0;
    if (S == 2 && (p + end0 >= size || (q[pos + p + end0] & 0xC0) != 0xC0))
        return 0;
    return 1;
}

dvda_hip_mlpdecoder *dvda_hip_open_mlpdecoder(unsigned g0_bps, unsigned g1_bps, unsigned g0_rate,
                                              unsigned g1_rate, unsigned channel_assignment, int device)
{
    dvda_hip_mlpdecoder *d = (dvda_hip_mlpdecoder *)calloc(1, sizeof(*d));
    if (!d)
        return NULL;
    d->device = device;
    d->params[0] = g0_bps;
    d->params[1] = g1_bps;
    d->params[2] = g0_rate;
    d->params[3] = g1_rate;
    d->params[4] = channel_assignment;
    d->ctx_segments = 4096;
    if (dvda_mlp_hip_stepper_create(&d->step, device) != DVDA_HIP_OK) {
        free(d);
        return NULL;                    /* no GPU: fail loudly, there is no CPU decoder here */
    }
    if (hipMalloc((void **)&d->d_meta, 4 * sizeof(uint64_t)) != hipSuccess ||
        hipMalloc((void **)&d->d_fir, sizeof(d->fir)) != hipSuccess) {
        dvda_hip_close_mlpdecoder(d);
        return NULL;
    }
    return d;
}

void dvda_hip_close_mlpdecoder(dvda_hip_mlpdecoder *d)
{
    if (!d)
        return;
    (void)hipSetDevice(d->device);
    dvda_mlp_hip_stepper_destroy(d->step);
    if (d->ctx)
        dvda_mlp_hip_destroy(d->ctx);
    (void)hipFree(d->d_bytes);
    (void)hipFree(d->d_meta);
    (void)hipFree(d->d_pcm);
    (void)hipFree(d->d_fir);
    free(d->h_pcm);
    free(d->q);
    free(d);
}

unsigned dvda_hip_mlpdecoder_status(const dvda_hip_mlpdecoder *d) { return d ? d->status : ~0u; }
size_t dvda_hip_mlpdecoder_queued_bytes(const dvda_hip_mlpdecoder *d) { return d ? d->qlen - d->decoded_end : 0; }
int dvda_hip_mlpdecoder_path(const dvda_hip_mlpdecoder *d) { return d ? d->slow : -1; }

static int grow_dev(void **p, size_t *cap, size_t need)
{
    if (need <= *cap)
        return 1;
    (void)hipFree(*p);
    *p = NULL;
    *cap = 0;
    need += need / 2 + 4096;
    if (hipMalloc(p, need) != hipSuccess)
        return 0;
    *cap = need;
    return 1;
}

unsigned dvda_hip_mlpdecoder_decode_packet(dvda_hip_mlpdecoder *d, const uint8_t *data, size_t len,
                                           const int32_t **planar, unsigned *channels)
{
    size_t pos, complete_end = 0, last_sync = 0;
    uint32_t n_sync = 0, n_sync_all = 0;     /* major syncs of the stream's own parameters: that restart every substream / all */
    dvda_mlp_stream_info info;
    uint64_t meta[4];
    uint64_t rows_cap, R, fresh;
    unsigned c, attempt;
    size_t padded;

    if (channels)
        *channels = 0;
    if (!d || (len && !data))
        return 0;
    /* a decoder that has hit what makes the reference assert() stays stopped: nothing more is queued
       (the queue cannot grow without bound behind an error) */
    if (d->status & ~(unsigned)DVDA_ST_BENIGN)
        return 0;
    /* ---- enqueue everything (src/mlp.c:349-351) */
    if (d->qlen + len > d->qcap) {
        size_t nc = d->qcap ? d->qcap : 8192;
        uint8_t *nq;
        while (nc < d->qlen + len)
            nc *= 2;
        nq = (uint8_t *)realloc(d->q, nc);
        if (!nq)
            return 0;
        d->q = nq;
        d->qcap = nc;
    }
    if (len)
        memcpy(d->q + d->qlen, data, len);
    d->qlen += len;

    /* ---- which access units are complete? (src/mlp.c:384-405) */
    for (pos = 0; pos + 4 <= d->qlen;) {
        const size_t size = 2 * ((((size_t)d->q[pos] & 0x0F) << 8) | d->q[pos + 1]);
        if (size < 4) {
            d->status |= DVDA_ST_EOF;   /* the reference stalls forever on such a header */
            break;
        }
        if (pos + size > d->qlen)
            break;
        if (sync_at(d->q, pos, size)) {
            const uint8_t p[4] = {d->q[pos + 8], d->q[pos + 9], (uint8_t)(d->q[pos + 11] & 0x1F),
                                  (uint8_t)(d->q[pos + 20] >> 4)};
            if (!d->have_sync) {
                memcpy(d->sync_params, p, 4);
                d->have_sync = 1;
            } else if (memcmp(d->sync_params, p, 3) != 0) {
                /* the reference drops such a frame and decodes on with the state it has
                   (src/mlp.c:450-455): it starts no segment; the batch tier walks through it */
                d->status |= DVDA_ST_SYNC_CHANGE;
                pos += size;
                complete_end = pos;
                continue;
            }
            n_sync_all++;
            /* (the queue is cut, and a fall-back decode starts, only at a unit that restarts every substream) */
            if (unit_restarts(d->q, pos, size, d->sync_params[3])) {
                n_sync++;
                last_sync = pos;
            }
        } else if (pos == 0) {
            d->status |= DVDA_ST_NO_SYNC;           /* substream count unknown: undefined in the reference */
            return 0;
        }
        pos += size;
        complete_end = pos;
    }
    if (complete_end <= d->decoded_end)
        return 0;                                   /* nothing newly decodable: bytes stay queued */

    /* ---- the new access units, from the state the call before left on the device */
    if (!d->slow) {
        const uint8_t *sp = d->sync_params;
        const uint32_t packed = (uint32_t)(sp[0] >> 4) | ((uint32_t)(sp[0] & 0x0F) << 4) | ((uint32_t)(sp[1] >> 4) << 8) |
                                ((uint32_t)(sp[1] & 0x0F) << 12) | ((uint32_t)sp[2] << 16) | ((uint32_t)sp[3] << 24);
        const unsigned rpa = rows_per_au(sp[1] >> 4);
        size_t at = d->decoded_end, units_all = 0, rows_cap_all, got = 0;
        int leave = 0, have_new_fir = 0, stepped = d->stepped, n_steps = 0, failed = 0, step_rc;
        int32_t new_fir[2 * 48];
        const int32_t *one_pcm = NULL;
        uint64_t one_stride = 0;
        unsigned nch = 0;
        for (pos = at; pos < complete_end; pos += 2 * ((((size_t)d->q[pos] & 0x0F) << 8) | d->q[pos + 1]))
            units_all++;
        rows_cap_all = units_all * (rpa ? rpa : 1);
        /* (a packet of more units than one step takes -- not what a disc reader sends -- is put together here) */
        if ((units_all > DVDA_STEP_MAX_UNITS || complete_end - at > DVDA_STEP_MAX_BYTES) && rows_cap_all * 6 * 4 > d->h_pcm_cap) {
            free(d->h_pcm);
            d->h_pcm_cap = rows_cap_all * 6 * 4 * 2 + 4096;
            d->h_pcm = (int32_t *)malloc(d->h_pcm_cap);
            if (!d->h_pcm) {
                d->h_pcm_cap = 0;
                return 0;
            }
        }
        while (at < complete_end && !leave) {
            size_t end = at, units = 0;
            const dvda_mlp_step_result *res = NULL;
            const int32_t *pcm = NULL;
            uint64_t stride = 0;
            while (end < complete_end && units < DVDA_STEP_MAX_UNITS) {
                const size_t size = 2 * ((((size_t)d->q[end] & 0x0F) << 8) | d->q[end + 1]);
                if (end + size - at > DVDA_STEP_MAX_BYTES)
                    break;
                end += size;
                units++;
            }
            if (units == 0 || rpa == 0) {
                leave = 1;              /* (a rate code outside the table, ...: the batch tier reports it) */
                break;
            }
            step_rc = dvda_mlp_hip_stepper_step(d->step, d->q + at, end - at, (uint32_t)units, packed, !stepped, &res, &pcm,
                                                &stride, &nch);
            if (step_rc == DVDA_HIP_ECAPACITY || step_rc == DVDA_HIP_EINVAL) {
                leave = 1;              /* not what a step takes: the batch tier decodes (and reports) it */
                break;
            }
            if (step_rc != DVDA_HIP_OK) {
                /* a HIP runtime failure is not a property of the stream: it is reported, not papered over by a silent
                   switch to the batch tier (which would meet the same device) */
                d->status |= DVDA_ST_DEVICE;
                failed = 1;
                break;
            }
            stepped = 1;
            n_steps++;
            if (res->status & (DVDA_ST_TIMING | DVDA_ST_SEQ)) {
                leave = 1;              /* not what the stepping kernel takes: the batch tier, from the last major sync */
                break;
            }
            if (res->status & ~(unsigned)DVDA_ST_BENIGN) {
                d->status |= res->status;           /* the reference would have assert()ed */
                failed = 1;
                break;
            }
            if (res->sync_seen) {
                memcpy(new_fir, res->fir, sizeof(new_fir));
                have_new_fir = 1;
            }
            if (n_steps == 1 && end == complete_end) {
                one_pcm = pcm;                      /* the usual case: one step, its pinned buffer is the answer */
                one_stride = stride;
            } else {
                if (rows_cap_all * nch * 4 > d->h_pcm_cap)
                    return 0;                       /* (sized above for six channels: cannot happen) */
                for (c = 0; c < nch; c++)
                    memcpy(d->h_pcm + (size_t)c * rows_cap_all + got, pcm + (size_t)c * stride, (size_t)res->rows_written * 4);
            }
            got += res->rows_written;
            at = end;
        }
        d->stepped = stepped;
        if (failed) {
            /* what the steps in front of the failing one decoded is handed out (they are in h_pcm: a call that fails
               on its first step has nothing), and counted, before the decoder stops for good */
            if (got == 0 || n_steps < 1 || one_pcm)
                return 0;
            for (c = 0; c < nch; c++)
                if (planar)
                    planar[c] = d->h_pcm + (size_t)c * rows_cap_all;
            if (channels)
                *channels = nch;
            d->rows_before += got;
            d->decoded_end = at;
            return (unsigned)got;
        }
        if (!leave) {
            for (c = 0; c < nch; c++)
                if (planar)
                    planar[c] = one_pcm ? one_pcm + (size_t)c * one_stride : d->h_pcm + (size_t)c * rows_cap_all;
            if (channels)
                *channels = nch;
            /* ---- keep what a fall-back needs: bytes from the last major sync on, the FIR history in front of it, how many
             *      of its frames have been handed out (the batch-tier path's own invariants, below) */
            if (n_sync >= 2) {
                uint64_t rows_after = 0;
                for (pos = last_sync; pos < complete_end;) {
                    const size_t size = 2 * ((((size_t)d->q[pos] & 0x0F) << 8) | d->q[pos + 1]);
                    const int foreign = sync_at(d->q, pos, size) &&
                                        (d->q[pos + 8] != sp[0] || d->q[pos + 9] != sp[1] || (d->q[pos + 11] & 0x1F) != sp[2]);
                    if (!foreign)
                        rows_after += rpa;
                    pos += size;
                }
                if (have_new_fir) {
                    memcpy(d->fir, new_fir, sizeof(d->fir));
                    d->have_fir = 1;
                }
                d->rows_before = rows_after;
                memmove(d->q, d->q + last_sync, d->qlen - last_sync);
                d->qlen -= last_sync;
                d->decoded_end = complete_end - last_sync;
            } else {
                d->rows_before += got;
                d->decoded_end = complete_end;
            }
            return (unsigned)got;
        }
        d->slow = 1;
    }

    /* ---- one small batch on the GPU over q[0, complete_end) */
    if (hipSetDevice(d->device) != hipSuccess)
        return 0;
    if (!d->ctx && dvda_mlp_hip_create(&d->ctx, d->device, 1, d->ctx_segments) != DVDA_HIP_OK)
        return 0;
    if (n_sync_all + 1 > d->ctx_segments) {
        dvda_mlp_hip_destroy(d->ctx);
        d->ctx = NULL;
        d->ctx_segments = 2 * (n_sync_all + 1);
        if (dvda_mlp_hip_create(&d->ctx, d->device, 1, d->ctx_segments) != DVDA_HIP_OK)
            return 0;
    }
    padded = (complete_end + 15) & ~(size_t)15;
    if (!grow_dev((void **)&d->d_bytes, &d->d_bytes_cap, padded + 64))
        return 0;
    if (hipMemset(d->d_bytes + complete_end, 0, padded + 64 - complete_end) != hipSuccess ||
        hipMemcpy(d->d_bytes, d->q, complete_end, hipMemcpyHostToDevice) != hipSuccess)
        return 0;
    meta[0] = 0;
    meta[1] = complete_end;
    meta[2] = 0;
    meta[3] = 0;
    if (hipMemcpy(d->d_meta, meta, sizeof(meta), hipMemcpyHostToDevice) != hipSuccess)
        return 0;
    for (;;) {
        uint32_t n_dev = 0;
        int rc;
        if (dvda_mlp_hip_index(d->ctx, d->d_bytes, padded, d->d_meta, d->d_meta + 1, 1, NULL))
            return 0;
        rc = dvda_mlp_hip_segment_count(d->ctx, &n_dev, NULL);
        if (rc == DVDA_HIP_OK)
            break;
        if (rc != DVDA_HIP_ECAPACITY)
            return 0;
        /* sync patterns inside payload bytes count as candidates too: a larger context */
        dvda_mlp_hip_destroy(d->ctx);
        d->ctx = NULL;
        d->ctx_segments = 2 * n_dev + 64;
        if (dvda_mlp_hip_create(&d->ctx, d->device, 1, d->ctx_segments) != DVDA_HIP_OK)
            return 0;
    }
    if (dvda_mlp_hip_stream_info(d->ctx, &info, 1, NULL))
        return 0;
    if (info.status & ~DVDA_ST_BENIGN) {
        d->status |= info.status;
        return 0;
    }
    rows_cap = info.mlp_frames * rows_per_au(info.group0_rate);
    if (d->have_fir) {
        if (hipMemcpy(d->d_fir, d->fir, sizeof(d->fir), hipMemcpyHostToDevice) != hipSuccess)
            return 0;
        dvda_mlp_hip_set_initial_fir(d->ctx, d->d_fir);
    } else {
        dvda_mlp_hip_set_initial_fir(d->ctx, NULL);
    }
    for (attempt = 0; attempt < 2; attempt++) {
        if (!grow_dev((void **)&d->d_pcm, &d->d_pcm_cap, (size_t)(rows_cap ? rows_cap : 1) * info.channels * 4))
            return 0;
        meta[3] = rows_cap;
        if (hipMemcpy(d->d_meta, meta, sizeof(meta), hipMemcpyHostToDevice) != hipSuccess)
            return 0;
        if (attempt && dvda_mlp_hip_index(d->ctx, d->d_bytes, padded, d->d_meta, d->d_meta + 1, 1, NULL))
            return 0;
        if (dvda_mlp_hip_decode(d->ctx, d->d_pcm, d->d_meta + 2, d->d_meta + 3, NULL) ||
            dvda_mlp_hip_stream_info(d->ctx, &info, 1, NULL))
            return 0;
        if (!(info.status & DVDA_ST_OVERFLOW))
            break;
        rows_cap = info.pcm_frames;                 /* non-standard timing: exact size, once more */
    }
    if (info.status & ~DVDA_ST_BENIGN) {
        d->status |= info.status;                   /* the reference would have assert()ed */
        return 0;
    }
    R = info.pcm_frames;
    if (R < d->rows_before)
        return 0;
    fresh = R - d->rows_before;

    /* ---- PCM frames decoded by THIS call, planar, RIFF order (src/mlp.c:527-533) */
    if ((size_t)R * info.channels * 4 > d->h_pcm_cap) {
        free(d->h_pcm);
        d->h_pcm_cap = (size_t)R * info.channels * 4 * 2 + 4096;
        d->h_pcm = (int32_t *)malloc(d->h_pcm_cap);
        if (!d->h_pcm) {
            d->h_pcm_cap = 0;
            return 0;
        }
    }
    for (c = 0; c < info.channels; c++) {
        if (fresh && hipMemcpy(d->h_pcm + (size_t)c * fresh, d->d_pcm + (size_t)c * rows_cap + d->rows_before,
                               (size_t)fresh * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return 0;
        if (planar)
            planar[c] = d->h_pcm + (size_t)c * fresh;
    }
    if (channels)
        *channels = info.channels;

    /* ---- keep what the next call needs: bytes from the last major sync on, the FIR history in
     *      front of that segment, and how many of its frames have been handed out */
    if (n_sync >= 2) {
        /* device segment indices also count sync patterns found inside payload bytes (DVDA_ST_FALSE_SYNC) and the
           syncs that restart nothing: wanted are the LIVE segments from the queue's new first unit on (their PCM frames
           have been handed out) and the live one in front of them (its FIR history at its end) */
        dvda_mlp_segment_info si;
        uint32_t n_dev = 0, prev_i;
        uint64_t rows_after = 0;
        int found = 0;
        if (dvda_mlp_hip_segment_count(d->ctx, &n_dev, NULL) || n_dev < 2)
            return 0;
        for (prev_i = n_dev; prev_i-- > 0;) {
            if (dvda_mlp_hip_segment_info(d->ctx, prev_i, &si, NULL))
                return 0;
            if (si.status & DVDA_ST_FALSE_SYNC)
                continue;
            if (si.offset < last_sync) {
                found = 1;
                break;
            }
            rows_after += si.pcm_frames;
        }
        if (!found)
            return 0;
        if (dvda_mlp_hip_segment_fir(d->ctx, prev_i, d->fir, NULL))
            return 0;
        d->have_fir = 1;
        d->rows_before = rows_after;
        memmove(d->q, d->q + last_sync, d->qlen - last_sync);
        d->qlen -= last_sync;
        d->decoded_end = complete_end - last_sync;
    } else {
        d->rows_before = R;
        d->decoded_end = complete_end;
    }
    return (unsigned)fresh;
}
