/* dvda_mlp_hip.h -- C ABI of the MI355X-native MLP decode path.
 *
 * This is the drop-in boundary for the ONE hot path of tuffy/libdvd-audio:
 * MLP (Meridian Lossless Packing) access-unit decode.  Everything a binding
 * needs is plain C: pointers, sizes and opaque handles -- no C++/HIP/torch
 * types.  Pointers named d_* are DEVICE pointers (HBM); `stream` is a
 * hipStream_t passed as void* (NULL = default stream).
 *
 * Two tiers:
 *
 *  (A) Batch tier -- the fast path.  Many independent MLP byte streams (titles,
 *      tracks, or single access units) resident in HBM are framed, parsed,
 *      filtered, rematrixed and written as planar int32 PCM in RIFF-WAVE
 *      channel order by hand-written gfx950 kernels.  It replaces, for a whole
 *      batch at once, what the reference does one PES payload at a time in
 *        dvda_mlpdecoder_decode_packet()   reference src/mlp.h:39-42, src/mlp.c:344-354
 *        mlpdecoder_decode/read_mlp_frame  reference src/mlp.c:360-405   (framing)
 *        decode_mlp_frame                  reference src/mlp.c:407-612   (sync, substreams,
 *                                                                          rematrix, shift, order)
 *        decode_substream .. filter_channel reference src/mlp.c:714-1306 (parse + FIR/IIR)
 *        rematrix_channels                 reference src/mlp.c:1308-1358
 *        checkdata_callback                reference src/mlp.c:1360-1399 (parity / CRC-8)
 *
 *  (B) Streaming tier -- the mlp.h mirror (declared further below):
 *      dvda_hip_open_mlpdecoder / dvda_hip_mlpdecoder_decode_packet /
 *      dvda_hip_close_mlpdecoder keep the exact call shape and return-value
 *      meaning of reference src/mlp.h:29-42 with the reference's BitstreamReader
 *      and aa_int containers replaced by (pointer, length) pairs; INTEGRATION.md
 *      shows the 40-line src/mlp.c replacement that binds them.
 *
 * All functions return 0 on success or a negative DVDA_HIP_E* code; they never
 * fall back to a CPU implementation.
 */
#ifndef DVDA_MLP_HIP_H
#define DVDA_MLP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVDA_HIP_OK          0
#define DVDA_HIP_ENODEV     -1   /* no usable HIP device / HIP runtime error */
#define DVDA_HIP_ENOMEM     -2
#define DVDA_HIP_EINVAL     -3
#define DVDA_HIP_ECAPACITY  -4   /* more streams / segments / bytes than the context was created for */
#define DVDA_HIP_ESTATE     -5   /* call order violated (decode before index, ...) */

/* per-stream / per-segment status bits (0 = decoded exactly as the reference
 * decodes a well-formed stream).  The low 10 bits mirror oracle/mlp_oracle.h. */
#define DVDA_ST_NO_SYNC      (1u << 0)   /* stream does not begin with a major sync             */
#define DVDA_ST_SYNC_CHANGE  (1u << 1)   /* a later major sync carries other stream parameters: that access
                                            unit was dropped and decoding went on, as the reference does
                                            (src/mlp.c:449-460) -- informational; together with
                                            DVDA_ST_IRREGULAR: could not be resolved, not decoded        */
#define DVDA_ST_PARITY       (1u << 2)
#define DVDA_ST_CRC          (1u << 3)
#define DVDA_ST_EOF          (1u << 4)   /* parse ran past a substream / frame end               */
#define DVDA_ST_RESTART      (1u << 5)   /* bad restart header                                   */
#define DVDA_ST_PARAMS       (1u << 6)   /* bad decoding parameters                              */
#define DVDA_ST_HUFFMAN      (1u << 7)   /* invalid residual code                                */
#define DVDA_ST_FILTER       (1u << 8)   /* FIR+IIR order > 8 or shift mismatch                  */
#define DVDA_ST_ENVELOPE     (1u << 9)   /* stream outside the reference's defined behaviour     */
/* conditions the batch tier reports instead of guessing: */
#define DVDA_ST_IRREGULAR    (1u << 16)  /* frame chain does not land on the next major sync     */
#define DVDA_ST_TIMING       (1u << 17)  /* an access unit's PCM-frame count differs from the
                                            stream's standard 40/80/160: stream decoded in order
                                            by the sequential pass                               */
#define DVDA_ST_MIDFRAME     (1u << 18)  /* matrix-class parameters changed after a frame's
                                            first block: segment decoded by the chain passes
                                            (rematrix per access unit, src/mlp.c:504-525)        */
#define DVDA_ST_CHAINED      (1u << 19)  /* segment's first block uses the FIR history of the
                                            previous segment (the reference never clears it):
                                            decoded by the chain passes                          */
#define DVDA_ST_OVERFLOW     (1u << 20)  /* output capacity (out_stride) too small               */
#define DVDA_ST_TRUNCATED    (1u << 21)  /* stream ends inside a frame (tail not consumed)       */
#define DVDA_ST_CAPACITY     (1u << 22)  /* more major syncs than max_segments: the stream (and those behind
                                            it) were indexed only in part -- create a larger context     */
#define DVDA_ST_GENERAL      (1u << 23)  /* segment was decoded by a pass behind the fast pass (informational) */
#define DVDA_ST_FALSE_SYNC   (1u << 24)  /* segment-level only: a sync pattern inside another segment's
                                            frame chain (payload / padding bytes), resolved and skipped */
#define DVDA_ST_SEQ          (1u << 25)  /* a restart header inside an access unit, or very dense parameter
                                            changes in a segment of the chain passes: stream decoded in
                                            order by the sequential pass                                  */
#define DVDA_ST_COLD         (1u << 26)  /* IIR taps, or more than two matrices: the fused row loop keeps
                                            neither in registers -- segment decoded by the chain passes   */
#define DVDA_ST_YIELD        (1u << 27)  /* the segment in front of a chained one (or one of the few segments
                                            without a chain among many with one), decoded by the chain passes
                                            with them instead of alone by the fast pass (informational)  */
#define DVDA_ST_DEVICE       (1u << 29)  /* streaming tier only: a HIP runtime call failed inside a step -- not a
                                            property of the stream; the decoder stops (it does not fall back to the
                                            batch tier on the same device without a word)                     */
/* DVDA_ST_CHAINED, _MIDFRAME, _COLD, _YIELD, _TIMING and _SEQ are raised by the fast pass and then decoded exactly by
 * the passes behind it (chain passes: parse in parallel, the filter recursion alone per channel, rematrix
 * in parallel; or the sequential pass); they stay set as information.  Bits that do not invalidate the PCM: */
#define DVDA_ST_BENIGN (DVDA_ST_TRUNCATED | DVDA_ST_CHAINED | DVDA_ST_MIDFRAME | DVDA_ST_TIMING | DVDA_ST_GENERAL | \
                        DVDA_ST_SEQ | DVDA_ST_SYNC_CHANGE | DVDA_ST_COLD | DVDA_ST_YIELD)

typedef struct dvda_mlp_hip_ctx dvda_mlp_hip_ctx;

typedef struct dvda_mlp_stream_info {
    uint64_t mlp_frames;      /* complete access units found                                  */
    uint64_t pcm_frames;      /* PCM frames decoded per channel                               */
    uint64_t bytes_consumed;  /* bytes covered by complete access units                       */
    uint32_t status;          /* DVDA_ST_* bits                                               */
    uint32_t channels;        /* channel count of the stream's channel assignment            */
    uint32_t substreams;      /* 1 or 2 (from the first major sync)                           */
    uint32_t assignment;      /* 5-bit channel assignment                                     */
    uint32_t group0_bps, group1_bps, group0_rate, group1_rate;  /* major sync codes          */
    uint32_t segments;        /* restart-delimited segments (units of parallel decode)        */
    uint32_t reserved;
} dvda_mlp_stream_info;

/* ------------------------------------------------------------------ tier A */

/* Creates a decode context on HIP device `device` able to hold an index for up
 * to max_streams streams / max_segments restart segments.  The workspace of the index and of the
 * fast pass is allocated here.  A batch that needs the passes behind the fast pass (chained FIR
 * history, mid-frame parameter changes, non-standard timing) grows their workspace on first use
 * inside dvda_mlp_hip_decode. */
int dvda_mlp_hip_create(dvda_mlp_hip_ctx **ctx, int device, uint32_t max_streams,
                        uint32_t max_segments);
void dvda_mlp_hip_destroy(dvda_mlp_hip_ctx *ctx);

/* Frame index (reference src/mlp.c:384-405 + the major-sync test of :614-654):
 * finds every major-sync access unit, walks the 12-bit size chain between
 * them, and derives each segment's first output row.  d_bytes must be readable
 * for total_bytes + 64 bytes; stream i occupies
 * [d_stream_off[i], d_stream_off[i] + d_stream_len[i]) and starts 16-byte aligned; the streams' ranges are
 * ascending and do not overlap (a major sync belongs to the one stream whose range holds it).  The ranges are
 * checked on the device: a stream that starts before the end of a stream in front of it (of two streams given
 * the same bytes, the second), is misaligned or leaves the buffer is not decoded and carries
 * DVDA_ST_IRREGULAR (| DVDA_ST_ENVELOPE) -- never a walk outside the buffer. */
/* (A caller that indexes the same buffers again and again -- a pipeline that reuses its staging buffers -- gets
 * the index's launch sequence replayed as one hipGraph from the third such call on; `stream` must then be a
 * real stream, not NULL.  DVDA_INDEX_GRAPH=0 in the environment switches that off.) */
int dvda_mlp_hip_index(dvda_mlp_hip_ctx *ctx, const uint8_t *d_bytes, uint64_t total_bytes,
                       const uint64_t *d_stream_off, const uint64_t *d_stream_len,
                       uint32_t n_streams, void *stream);

/* Decodes every indexed segment.  Stream i's PCM is written planar at
 * d_pcm[d_out_off[i] + channel * d_out_stride[i] + pcm_frame], channel in
 * RIFF-WAVE order (reference src/mlp.c:416-438, 527-533).
 * The call waits once on `stream` for the fast pass: a 32-byte summary tells the host whether the
 * chain passes / the sequential pass have anything to do (they are not launched otherwise); what it
 * enqueues after that is asynchronous again.
 * An index may be decoded more than once (a caller that comes back with a larger buffer after
 * DVDA_ST_OVERFLOW, or wants the titles in a second buffer): every call starts from what the index left,
 * nothing of the call before it is carried over. */
int dvda_mlp_hip_decode(dvda_mlp_hip_ctx *ctx, int32_t *d_pcm, const uint64_t *d_out_off,
                        const uint64_t *d_out_stride, void *stream);

/* The same without the host ever waiting and without any allocation: every pass is enqueued on `stream` and the call
 * returns -- so a caller can have the index of the next batch running (another context, another stream) while this
 * batch decodes, or capture the call.  What the fast pass defers is decoded on the workspaces dvda_mlp_hip_reserve
 * left: the chain passes are launched for as many deferred segments / PCM frames as were reserved and find their
 * work on the device, the sequential pass for as many streams as were reserved; a batch that needs more than was
 * reserved is NOT decoded short silently -- the streams left over carry DVDA_ST_CAPACITY (dvda_mlp_hip_stream_info),
 * and the blocking call or a larger reservation decodes them.  With nothing reserved (a batch of regular titles with
 * raw lead-in blocks at their restart points) only the fast pass and two small bookkeeping kernels are launched.
 * Host-blocking entry points are dvda_mlp_hip_decode (once per call, twice when chains exist; it also grows the
 * workspaces on first use) and the *_info / *_count calls. */
int dvda_mlp_hip_decode_async(dvda_mlp_hip_ctx *ctx, int32_t *d_pcm, const uint64_t *d_out_off,
                              const uint64_t *d_out_stride, void *stream);

/* Sizes the workspaces of the passes behind the fast pass ahead of time (they only ever grow): chain passes for
 * `chain_segments` deferred segments holding `chain_pcm_frames` PCM frames in all (for titles whose restart points
 * carry FIR taps -- what encoders write -- that is every segment and every frame of the batch), the sequential pass
 * for `seq_streams` streams at a time.  After it neither decode call allocates for a batch within these sizes. */
int dvda_mlp_hip_reserve(dvda_mlp_hip_ctx *ctx, uint64_t chain_pcm_frames, uint32_t chain_segments,
                         uint32_t seq_streams);

/* PCM layout of dvda_mlp_hip_decode (default DVDA_PCM_PLANAR, above).  DVDA_PCM_INTERLEAVED writes
 * frame-major instead: d_pcm[d_out_off[i] + pcm_frame * channels + channel] -- the order the
 * reference's dvda_read() hands out (src/dvd-audio.c:781-792); d_out_stride[i] stays the capacity
 * in PCM frames, so stream i occupies channels * d_out_stride[i] values either way.  Same sample
 * values; each lane's stores become one contiguous run, which the memory side takes much better
 * (6-channel titles: 5-8 % faster, 40 % less write traffic; 2-channel: the same speed; other channel
 * counts: a few per cent slower than planar -- DESIGN.md section 4). */
#define DVDA_PCM_PLANAR      0u
#define DVDA_PCM_INTERLEAVED 1u
/* The output stage fused into the decode: d_pcm receives the interleaved little-endian
 * WAV payload dvda2wav writes -- frame-major like DVDA_PCM_INTERLEAVED, every value as write_signed(24) /
 * write_signed(16) (reference utils/dvda2wav.c:326-334, src/bitstream.c:2846-2857: low bits - 1 bits plus a
 * sign bit taken from v < 0).  Stream i's payload starts at byte 4 * d_out_off[i] of d_pcm (d_out_off stays in
 * int32 units, so every stream starts dword-aligned) and holds pcm_frames * channels * (3 | 2) bytes;
 * d_out_stride[i] stays the capacity in PCM frames.  Same bytes as dvda_mlp_hip_pack_wav() of the int32 PCM. */
#define DVDA_PCM_WAV24       2u
#define DVDA_PCM_WAV16       3u
int dvda_mlp_hip_set_pcm_layout(dvda_mlp_hip_ctx *ctx, uint32_t layout);

/* Blocks until the work enqueued on `stream` is done and copies the per-stream
 * results to host memory. */
int dvda_mlp_hip_stream_info(dvda_mlp_hip_ctx *ctx, dvda_mlp_stream_info *infos, uint32_t n,
                             void *stream);

/* Number of segments found by the last index call (blocks on `stream`). */
int dvda_mlp_hip_segment_count(dvda_mlp_hip_ctx *ctx, uint32_t *n_segments, void *stream);

/* Average duration in milliseconds of the fast-pass kernel launches recorded since
 * the last call (HIP events on the launch stream, a ring of the newest 256 decode calls made at
 * create time); *launches receives the count.  Blocks until those launches have finished. */
int dvda_mlp_hip_kernel_time(dvda_mlp_hip_ctx *ctx, double *avg_ms, uint32_t *launches);

/* The same ring read the other way: average device time of a whole decode call -- from the first kernel of the fast
 * pass to the last kernel the call enqueued (chain passes, sequential pass, bookkeeping; for the blocking call the
 * gaps while the host read the summary count too).  Does not reset the ring: call it before
 * dvda_mlp_hip_kernel_time. */
int dvda_mlp_hip_decode_time(dvda_mlp_hip_ctx *ctx, double *avg_ms, uint32_t *calls);

/* ------------------------------------------------------------------ several GPUs from one C host
 * titles are independent -- the reference decodes one track at a time through one decoder,
 * src/dvd-audio.c:597-657, and two decoders share nothing -- so a host with a list of streams gives every GPU its
 * own sub-list: greedy longest-processing-time on the compressed size (dvda_mlp_hip_shard, the same deterministic
 * partition as libdvd-audio_amd/shard.py), one host thread + decode context + HIP stream per device entry, no
 * exchange between devices, a small summary added up over RCCL -- or on the host, see `reduction` -- (csrc/mlp_multi.cpp).  The streams and the PCM are
 * HOST buffers here (each device gets its own copies); a device may be named more than once. */
typedef struct dvda_mlp_hip_multi dvda_mlp_hip_multi;
typedef struct dvda_mlp_multi_summary {
    uint64_t pcm_frames;                  /* sum over the streams                                         */
    uint64_t samples;                     /* sum of pcm_frames x channels                                 */
    uint64_t compressed_bytes;            /* sum of the stream lengths                                    */
    uint64_t compressed_bytes_max_device; /* the largest share one device entry got (load balance)        */
    uint32_t streams_with_errors;         /* streams whose status has a bit outside DVDA_ST_BENIGN        */
    uint32_t devices;                     /* device entries the list was dealt to                         */
    /* (round 5) what a caller needs to judge a run on N devices without a profiler: wall time of the slowest and the
     * fastest device entry (copy in + index + decode + copy out, its host thread's clock), and the load imbalance
     * of the deal = largest share of the compressed bytes / mean share (1.0 = even)                              */
    double device_ms_max;
    double device_ms_min;
    double imbalance;
    /* (round 6) who added the summary up: 1 = RCCL (one communicator per device, two all-reduces -- a device list that
     * names every device once, librccl present), 0 = the host (a device named twice, no librccl, DVDA_MULTI_RCCL=0) */
    uint32_t reduction;
    uint32_t reserved;
} dvda_mlp_multi_summary;

/* part_of[i] = which of `parts` parts stream i (sizes[i] bytes) goes to */
int dvda_mlp_hip_shard(const uint64_t *sizes, uint32_t n, uint32_t parts, uint32_t *part_of);
/* one decode context per entry of devices[]; every one sized for max_streams / max_segments */
int dvda_mlp_hip_create_multi(dvda_mlp_hip_multi **multi, const int *devices, uint32_t n_devices,
                              uint32_t max_streams, uint32_t max_segments);
void dvda_mlp_hip_destroy_multi(dvda_mlp_hip_multi *multi);
uint32_t dvda_mlp_hip_multi_devices(const dvda_mlp_hip_multi *multi);
/* wall time (ms) the last dvda_mlp_hip_decode_multi spent on device entry `entry`, and the compressed bytes it got */
int dvda_mlp_hip_multi_device_time(const dvda_mlp_hip_multi *multi, uint32_t entry, double *ms, uint64_t *bytes);
/* Decodes n_streams host streams (streams[i], lengths[i] bytes) and returns when all of them are done.  pcm[i]
 * receives stream i in `layout` (DVDA_PCM_*): planar = [channel][capacity_frames[i]] int32, interleaved =
 * [frame][channel] int32, WAV24 / WAV16 = the packed payload; capacity_frames[i] = PCM frames per channel pcm[i]
 * has room for (a stream that needs more is reported DVDA_ST_OVERFLOW, infos[i].pcm_frames = the size needed).
 * infos[i] as dvda_mlp_hip_stream_info would give it; summary may be NULL.  Blocks; no CPU fallback. */
int dvda_mlp_hip_decode_multi(dvda_mlp_hip_multi *multi, const uint8_t *const *streams, const uint64_t *lengths,
                              uint32_t n_streams, uint32_t layout, void *const *pcm, const uint64_t *capacity_frames,
                              dvda_mlp_stream_info *infos, dvda_mlp_multi_summary *summary);

/* Range-checked diagnostic build (-DDVDA_BOUNDS, tests/test_gpu_soak.py): every index a kernel forms into a
 * workspace of the library is compared with the workspace's size; out4 = {violations so far, and of the first
 * one: array tag, index, capacity}.  Returns 1 from a checked build, 0 from the shipped library (which does not
 * check and reports zeros), < 0 on error. */
int dvda_mlp_hip_bounds_violations(unsigned long long *out4);

/* Which fast-pass kernels run.  0 (default): chosen per batch from the substream counts the index
 * found -- streams with one substream take the one-lane-per-segment kernel, streams with two the kernel
 * whose lane reads both substreams of its segment (round 6; any split of the channels over the two), a
 * mixed batch both (a kernel whose class is absent exits at once).  1 forces the one-substream kernel for
 * the whole batch (a two-substream stream is then reported as DVDA_ST_ENVELOPE); 2 = 3.
 * Under 0 a SMALL batch -- at most 4 096 segments and 32 768 access units, counted on the device by the index --
 * is decoded by the wave-cooperative kernel instead (csrc/mlp_coop.h: one wave per (segment, substream), the
 * bit-serial symbol scan in scalar registers, residuals / filter / rematrix at the width they have): BASELINE
 * configs[3]'s 1 024 single access units, a single title, a streaming-tier packet.  64 forces that kernel for any
 * batch; 1 / 2 never use it; 3 = the lane kernels picked per batch as under 0, never the cooperative kernel (tests
 * and soaks that want the lane kernels on small batches). */
int dvda_mlp_hip_set_lanes_per_segment(dvda_mlp_hip_ctx *ctx, uint32_t lanes);

/* Which form the chain passes behind the parse pass take (segments that continue the FIR history of the one before
 * them, src/mlp.c:297-304, 1302; csrc/mlp_chain.h).  0 (default): chosen per batch -- few deferred segments (one title,
 * a small batch): the recursion in place on the planes and a parallel rematrix pass behind it; many: ONE walk over
 * the planes by two-wave workgroups (k_chain_fused).  1 / 2 force the fused / the two-pass form (tests).  Same PCM. */
int dvda_mlp_hip_set_chain_form(dvda_mlp_hip_ctx *ctx, uint32_t form);

/* Per-segment results of the last decode (blocks on `stream`). */
typedef struct dvda_mlp_segment_info {
    uint64_t offset, end;     /* byte range of the segment in the input buffer */
    uint32_t stream, mlp_frames, pcm_frames, status;
} dvda_mlp_segment_info;
int dvda_mlp_hip_segment_info(dvda_mlp_hip_ctx *ctx, uint32_t segment, dvda_mlp_segment_info *info,
                              void *stream);

/* FIR history (reference struct filter_parameters.state, src/mlp.c:73-77, never cleared there)
 * at the END of `segment`, for both substreams: host_fir[2][48] = [substream][channel slot * 8 + tap],
 * tap 0 = most recent output.  Blocks on `stream`. */
int dvda_mlp_hip_segment_fir(dvda_mlp_hip_ctx *ctx, uint32_t segment, int32_t *host_fir, void *stream);

/* FIR history the streams START with (device pointer, [n_streams][2][48], or NULL = fresh
 * decoders).  Lets a caller decode a stream in pieces cut at major syncs. */
int dvda_mlp_hip_set_initial_fir(dvda_mlp_hip_ctx *ctx, const int32_t *d_init_fir);

const char *dvda_mlp_hip_version(void);

/* Self-test: runs the kernels' arithmetic code-book decode on the device for every (book 0..3,
 * 9-bit peek) and returns host_out[book * 512 + peek] = value | length << 8 (value 0xFF = invalid
 * code; book 0 = no code = 0).  The tables it must equal are reference src/mlp_codebook{1,2,3}.json. */
int dvda_mlp_hip_selftest_huff(int device, uint32_t *host_out);

/* Self-test: reads n fields from host_bytes with the kernels' MSB-first bit reader on the device --
 * widths[i] > 0: unsigned field of that many bits (0..32), < 0: signed field of -widths[i] bits (sign bit
 * first, two's complement), 0: nothing -- into host_out[i].  resident != 0 takes the unsigned fields
 * (< 32 bits) through the row loop's branch-free read.  The contract is reference
 * src/bitstream.c:1077-1111, 1198-1206; its known answers are src/bitstream.c:4864-4868, 4940-4944. */
int dvda_mlp_hip_selftest_bits(int device, const uint8_t *host_bytes, uint32_t n_bytes, const int32_t *host_widths,
                               uint32_t n, int64_t *host_out, uint32_t resident);

/* ------------------------------------------------------------------ PCM tier */
/* Raw-PCM AOB tracks: what reference src/dvd-audio.c:1016-1084 (decode_pcm_audio),
 * src/packet.c:61-188 (pack header / PES walk) and src/pcm.c:99-193 (AOB byte un-swizzle, sign
 * extension) do packet by packet, for n_sectors 2048-byte sectors resident in HBM at once.
 * Output: planar int32, channel c at d_pcm[c * stride + frame] -- the order the reference
 * appends into `samples`.  Only whole 2-frame chunks of a packet are decoded (src/pcm.c:149).
 * d_work: device scratch of dvda_pcm_hip_workspace_words(n_sectors) uint32 words.  After the call
 * it holds, per sector, what the sector contributed and where it starts in the output:
 *   d_work[s]                 PCM frames (or MLP payload bytes, for the demux call) of sector s
 *   d_work[n_sectors + s]     their exclusive prefix sum; d_work[2 * n_sectors] = the total
 * (the disc tier reads the prefix to cut a track at a sector boundary). */
size_t dvda_pcm_hip_workspace_words(uint32_t n_sectors);
int dvda_pcm_hip_decode_sectors(const uint8_t *d_sectors, uint32_t n_sectors, unsigned bits_per_sample,
                                unsigned channels, int32_t *d_pcm, uint64_t stride, uint32_t *d_work,
                                void *stream);
/* blocks on `stream`; *bad_sectors = sectors whose pack / PES / codec header was malformed */
int dvda_pcm_hip_result(const uint32_t *d_work, uint32_t n_sectors, uint64_t *pcm_frames,
                        uint32_t *bad_sectors, void *stream);

/* MLP track demux: AOB sectors of an MLP track -> the contiguous MLP byte
 * stream (every 0xBD packet with codec 0xA1, audio header and pad_2 stripped, in order: what
 * reference src/dvd-audio.c:1151-1227 enqueues packet by packet).  d_work as for the PCM tier;
 * dvda_pcm_hip_result() then returns the byte count in *pcm_frames and the malformed sectors. */
int dvda_mlp_hip_demux_sectors(const uint8_t *d_sectors, uint32_t n_sectors, uint8_t *d_mlp,
                               uint64_t mlp_cap, uint32_t *d_work, void *stream);

/* Output stage: planar int32 PCM -> the interleaved little-endian WAV payload
 * dvda2wav writes: frame-major interleave of reference src/dvd-audio.c:781-792, each value as
 * write_signed(bits) (utils/dvda2wav.c:326-334, src/bitstream.c:2846-2857).  bits = 16 or 24;
 * d_out receives frames * channels * bits/8 bytes. */
int dvda_mlp_hip_pack_wav(const int32_t *d_pcm, uint64_t stride, unsigned channels, uint64_t frames,
                          unsigned bits_per_sample, uint8_t *d_out, void *stream);

/* ------------------------------------------------------------------ tier B */
/* The mlp.h mirror: same three calls, same meaning as reference src/mlp.h:29-42 /
 * src/mlp.c:265-354, with the reference's containers replaced by plain memory:
 *   stream_parameters  -> five unsigned codes (src/stream_parameters.h:23-29)
 *   BitstreamReader*   -> (data, len): the bytes the reference would enqueue (src/mlp.c:349-351)
 *   aa_int* samples    -> planar[c] points at the PCM frames decoded by THIS call for RIFF
 *                         channel c (valid until the next call); the binding appends them to
 *                         samples->_[c]
 * decode_packet returns the number of PCM frames decoded by the call; 0 = nothing decodable
 * yet (bytes stay queued) exactly as in the reference.  Where the reference assert()s, the
 * call returns 0 and dvda_hip_mlpdecoder_status() reports the DVDA_ST_* bits.
 * Compatibility tier: every call is a small GPU batch; use tier A for throughput. */
typedef struct dvda_hip_mlpdecoder dvda_hip_mlpdecoder;

dvda_hip_mlpdecoder *dvda_hip_open_mlpdecoder(unsigned group_0_bps, unsigned group_1_bps,
                                              unsigned group_0_rate, unsigned group_1_rate,
                                              unsigned channel_assignment, int device);
void dvda_hip_close_mlpdecoder(dvda_hip_mlpdecoder *decoder);
unsigned dvda_hip_mlpdecoder_decode_packet(dvda_hip_mlpdecoder *decoder, const uint8_t *data,
                                           size_t len, const int32_t **planar, unsigned *channels);
unsigned dvda_hip_mlpdecoder_status(const dvda_hip_mlpdecoder *decoder);
size_t dvda_hip_mlpdecoder_queued_bytes(const dvda_hip_mlpdecoder *decoder);
/* Which path the decoder is on.  0: its state lives on the device and a call decodes exactly the access units it was
 * given (one workgroup, csrc/mlp_coop.h k_coop<false, true>; csrc/mlp_step.h) -- the usual case.  1: the stream left what
 * that path takes (an access unit of non-standard length, src/mlp.c:719-738, or one larger than 4 KB) and every call
 * decodes from the stream's last major sync on through the batch tier, as rounds 1-3 did for every stream. */
int dvda_hip_mlpdecoder_path(const dvda_hip_mlpdecoder *decoder);

#ifdef __cplusplus
}
#endif
#endif
