/* dvd-audio-hip.h -- tier C of the HIP decoder: the disc-level API (SURVEY.md 8(b) "outer
 * boundary", rows f-1 and f-4), exported by libdvd_audio_hip.so.
 *
 * The 27 entry points below have the names, argument meaning, 1-based numbering, NULL-on-failure
 * and ownership rules of the reference's public header (reference include/dvd-audio.h:59-201),
 * so a program written against that header -- utils/dvda2wav.c for one -- links against this
 * library unchanged.  What differs is the inside: a track's AOB sectors are read, uploaded and decoded
 * on the GPU in batches
 *
 *     sector walk + payload gather   (reference src/packet.c:61-188, src/dvd-audio.c:1151-1248)
 *     major-sync search, end-of-track rule            (src/dvd-audio.c:1167-1194, 1238-1421)
 *     MLP decode (tier A of dvda_mlp_hip.h)  /  PCM un-swizzle      (src/mlp.c, src/pcm.c:99-193)
 *
 * -- a track of at most DVDA_WINDOW_SECTORS sectors (default 8 192 = 16 MiB) as ONE batch when its reader is
 * opened; a longer one, MLP or raw PCM, window by window: a producer thread reads, demultiplexes and decodes
 * window k + 1 into one of two pinned buffers while dvda_read() / dvda_hip_reader_wav_next() hand out window k,
 * so what a reader holds is bounded by the window, not by the track (the reference streams a track packet by
 * packet, src/dvd-audio.c:752-795).  There is no CPU decode path: a
 * track reader cannot be opened without a HIP device.  CPPM-protected discs are not handled
 * (`device` is accepted and ignored; reference src/aob.c:109-127).
 */
#ifndef DVD_AUDIO_HIP_H
#define DVD_AUDIO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define DVDA_HIP_PTS_PER_SECOND 90000

typedef struct DVDA_s DVDA;
typedef struct DVDA_Titleset_s DVDA_Titleset;
typedef struct DVDA_Title_s DVDA_Title;
typedef struct DVDA_Track_s DVDA_Track;
typedef struct DVDA_Track_Reader_s DVDA_Track_Reader;

typedef enum { DVDA_PCM, DVDA_MLP } dvda_codec_t;

/* ---- disc: AUDIO_TS.IFO (reference src/dvd-audio.c:323-360, 896-930) */
DVDA *dvda_open(const char *audio_ts_path, const char *device);
void dvda_close(DVDA *dvda);
unsigned dvda_titleset_count(const DVDA *dvda);

/* ---- titleset: ATS_XX_0.IFO (src/dvd-audio.c:362-424, 932-1014) */
DVDA_Titleset *dvda_open_titleset(DVDA *dvda, unsigned titleset);
void dvda_close_titleset(DVDA_Titleset *titleset);
unsigned dvda_titleset_number(const DVDA_Titleset *titleset);
unsigned dvda_title_count(const DVDA_Titleset *titleset);

/* ---- title: track table and the sector range of every track (src/dvd-audio.c:426-515) */
DVDA_Title *dvda_open_title(DVDA_Titleset *titleset, unsigned title);
void dvda_close_title(DVDA_Title *title);
unsigned dvda_title_number(const DVDA_Title *title);
unsigned dvda_track_count(const DVDA_Title *title);
unsigned dvda_title_pts_length(const DVDA_Title *title);

/* ---- track (src/dvd-audio.c:517-584) */
DVDA_Track *dvda_open_track(DVDA_Title *title, unsigned track);
void dvda_close_track(DVDA_Track *track);
unsigned dvda_track_number(const DVDA_Track *track);
unsigned dvda_track_pts_index(const DVDA_Track *track);
unsigned dvda_track_pts_length(const DVDA_Track *track);
unsigned dvda_track_first_sector(const DVDA_Track *track);
unsigned dvda_track_last_sector(const DVDA_Track *track);

/* ---- track reader (src/dvd-audio.c:586-794): a short track is decoded when its reader is opened, a long one in
 *      windows while it is read (see the top of this file; dvda_hip_reader_windowed()) */
DVDA_Track_Reader *dvda_open_track_reader(const DVDA_Track *track);
void dvda_close_track_reader(DVDA_Track_Reader *reader);
dvda_codec_t dvda_codec(const DVDA_Track_Reader *reader);
unsigned dvda_bits_per_sample(const DVDA_Track_Reader *reader);
unsigned dvda_sample_rate(const DVDA_Track_Reader *reader);
unsigned dvda_channel_count(const DVDA_Track_Reader *reader);
unsigned dvda_riff_wave_channel_mask(const DVDA_Track_Reader *reader);
/* interleaved int samples in RIFF-WAVE channel order; returns the PCM frames delivered, fewer
 * than asked only at the end of the track, 0 afterwards */
unsigned dvda_read(DVDA_Track_Reader *reader, unsigned pcm_frames, int buffer[]);

/* ---- additions of this library (not in the reference API) */
/* HIP device used by track readers opened afterwards (default 0) */
void dvda_hip_set_device(int device);
/* on != 0: MLP track readers opened afterwards are decoded straight into the WAV payload dvda2wav would write
 * (the output stage -- dvda_read's interleave + write_signed at the stream's bit depth -- runs inside the decode
 * kernels, DVDA_PCM_WAV24 / DVDA_PCM_WAV16): no int32 PCM buffer, no packing pass.  Such a reader serves
 * dvda_hip_reader_wav_payload() only; dvda_read() on it returns 0 (dvda_hip_reader_wav_only() tells that case from an
 * empty track).  Default off (the reference API's int samples). */
void dvda_hip_set_wav_output(int on);
/* Both options are per calling THREAD (a host that fans tracks out over several devices sets them in each worker).
 * The per-reader form: opens the track on HIP device `device`, as a WAV-payload-only reader if wav_output != 0,
 * whatever the thread's defaults are, and leaves those alone. */
DVDA_Track_Reader *dvda_hip_open_track_reader_on(const DVDA_Track *track, int device, int wav_output);
/* != 0: the reader holds the WAV payload only (opened with wav_output): dvda_read() on it returns 0 frames -- NOT
 * because the track is empty; take the payload with dvda_hip_reader_wav_payload() */
int dvda_hip_reader_wav_only(const DVDA_Track_Reader *reader);
/* ---- long tracks (round 5).  An MLP track of more than a window of sectors (DVDA_WINDOW_SECTORS in the environment,
 * default 8192 = 16 MiB) is read, demultiplexed and decoded window by window while it is served, as the reference
 * streams it (src/dvd-audio.c:751-795): what is resident on the host and on the device is bounded by the window, not
 * by the track.  dvda_read() works on such a reader as on any other.  Its WAV payload comes piece by piece: */
/* the next piece of the track's WAV data bytes (valid until the next call on this reader); 0 at the end of the track.
 * (A reader that is not windowed hands out its whole payload in one piece, once.) */
unsigned long long dvda_hip_reader_wav_next(DVDA_Track_Reader *reader, const unsigned char **payload);
/* != 0: the reader decodes in windows (dvda_hip_reader_total_frames() is then the count so far: final when the
 * track has been read to its end) */
int dvda_hip_reader_windowed(const DVDA_Track_Reader *reader);
/* != 0: a window of the track could not be read or decoded (what makes dvda_open_track_reader return NULL when it
 * happens in a track's first window): dvda_read() / dvda_hip_reader_wav_next() then end early */
int dvda_hip_reader_failed(const DVDA_Track_Reader *reader);
/* peaks of what a windowed reader held: bytes of host memory it allocated (pinned buffers, the bytes kept between
 * windows), and of device memory in use beyond what was in use when it was opened (hipMemGetInfo, sampled after every
 * window); returns 0 on a reader that is not windowed */
int dvda_hip_reader_memory(const DVDA_Track_Reader *reader, unsigned long long *host_peak, unsigned long long *device_peak);
/* A windowed reader's decode context, device buffers and pinned buffers are kept, when it is closed, by the closing thread
 * for the next windowed reader that thread opens on the same device (one set per thread).  This frees the calling
 * thread's set; a thread that opens no more readers calls it before it ends. */
void dvda_hip_release_cached_buffers(void);
/* status word of the decode behind a reader: DVDA_ST_* bits of dvda_mlp_hip.h (0 = clean) */
unsigned dvda_hip_reader_status(const DVDA_Track_Reader *reader);
/* PCM frames the reader holds in total */
unsigned long long dvda_hip_reader_total_frames(const DVDA_Track_Reader *reader);
/* The rest of the track as the WAV payload dvda2wav would write for it (interleaved,
 * little-endian, write_signed semantics, packed on the GPU): returns the byte count and a
 * pointer valid until the reader is closed; the reader is at its end afterwards. */
unsigned long long dvda_hip_reader_wav_payload(DVDA_Track_Reader *reader, const unsigned char **payload);

#ifdef __cplusplus
}
#endif
#endif
