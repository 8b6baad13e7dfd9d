/* dvda_disc.c -- tier C of the HIP decoder (include/dvd-audio-hip.h): the disc-level API,
 * host side, plain C.  Built into libdvd_audio_hip.so on top of libdvda_mlp_hip.so.
 *
 * Mirrors what reference src/dvd-audio.c + src/aob.c + src/packet.c + src/audio_ts.c do
 * (SURVEY.md 8(b) outer boundary, rows f-1 and f-4), with the per-packet decode loop replaced
 * by one GPU batch per track:
 *
 *   IFO tables          parsed on the host, field for field (src/dvd-audio.c:896-1014)
 *   track sector range  dvda_open_title's rules (src/dvd-audio.c:426-492)
 *   AOB files           ATS_XX_1..9.AOB taken as one sector sequence (src/aob.c:86-190)
 *   codec probe         first 0xBD packet at/after the track's first sector (src/dvd-audio.c:586-655)
 *   MLP track           sectors -> GPU gather of the MLP payload (dvda_mlp_hip_demux_sectors);
 *                       stream start = first major sync found byte by byte (locate_mlp_parameters,
 *                       src/dvd-audio.c:1327-1365); stream end = the first major sync at or after
 *                       the first payload byte of a sector beyond the track's last sector
 *                       (decode_mlp_audio + mlp_data_to_major_sync, src/dvd-audio.c:1167-1194,
 *                       1367-1421); then tier A index + decode of that one stream
 *   PCM track           sectors -> GPU un-swizzle (dvda_pcm_hip_decode_sectors); whole packets are
 *                       delivered until lround(PTS length * rate / 90000) frames are covered
 *                       (open_pcm_track_reader / decode_pcm_audio, src/dvd-audio.c:958-1084)
 *   dvda_read           interleave out of the decoded track (src/dvd-audio.c:757-794)
 *
 * Streams the reference would abort on (assert) make dvda_open_track_reader return NULL here.
 */
#include <ctype.h>
#include <dirent.h>
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include "../../include/dvd-audio-hip.h"
#include "../../include/dvda_mlp_hip.h"

#define SECTOR 2048u
#define MAX_AOBS 9
#define CODEC_PCM 0xA0u
#define CODEC_MLP 0xA1u

/* per THREAD: a host that fans tracks out over several devices (tools/dvda2wav_hip.c --devices) sets them in each
 * of its worker threads; dvda_hip_open_track_reader_on() names both for one reader and touches neither */
static _Thread_local int g_device = 0;

static _Thread_local int g_wav_output = 0;
void dvda_hip_set_device(int device) { g_device = device; }
void dvda_hip_set_wav_output(int on) { g_wav_output = on != 0; }

/* ------------------------------------------------------------------ records */
struct ifo_title {
    unsigned track_count, index_count, pts_length;
    struct {
        unsigned index_number, pts_index, pts_length;
    } track[256];
    struct {
        unsigned first, last;
    } index[256];
};

struct DVDA_s {
    char *dir;
    unsigned titlesets;
};

struct DVDA_Titleset_s {
    char *dir;
    unsigned number, title_count;
    struct ifo_title *title;
};

struct track_span {
    unsigned pts_index, pts_length, first, last;
};

struct DVDA_Title_s {
    char *dir;
    unsigned titleset, number, track_count, pts_length;
    struct track_span t[256];
};

struct DVDA_Track_s {
    char *dir;
    unsigned titleset, title, number;
    struct track_span s;
};

struct mlp_windows;                 /* a long MLP track read window by window (below) */

struct DVDA_Track_Reader_s {
    dvda_codec_t codec;
    struct mlp_windows *win;       /* != NULL: the track is decoded in windows of bounded size as it is read */
    unsigned bps_code[2], rate_code[2], assignment;
    unsigned channels, status;
    uint64_t frames, served, stride;
    int interleaved;           /* MLP tracks: frame-major [frame][channel] = the dvda_read order;
                                  PCM tracks: planar [channel][stride]; RIFF-WAVE channel order */
    int32_t *pcm;              /* host copy of d_pcm, made by the first dvda_read() */
    int32_t *d_pcm;            /* device copy, kept for the GPU WAV packer */
    uint8_t *wav;              /* host payload produced by dvda_hip_reader_wav_payload */
    uint8_t *d_wav;            /* MLP reader opened under dvda_hip_set_wav_output(1): the decode kernels wrote the WAV
                                  payload themselves (DVDA_PCM_WAV24 / WAV16); there is no int32 PCM for dvda_read() */
    uint64_t wav_bytes;
};

/* ------------------------------------------------------------------ files */
static int same_name(const char *a, const char *b)
{
    for (; *a && *b; a++, b++)
        if (toupper((unsigned char)*a) != toupper((unsigned char)*b))
            return 0;
    return *a == *b;
}

/* case-insensitive lookup inside the AUDIO_TS directory (src/audio_ts.c:37-73) */
static char *find_file(const char *dir, const char *name)
{
    DIR *d = opendir(dir);
    struct dirent *e;
    char *path = NULL;
    if (!d)
        return NULL;
    while ((e = readdir(d)) != NULL) {
        if (same_name(name, e->d_name)) {
            const size_t n = strlen(dir) + 1 + strlen(e->d_name) + 1;
            path = malloc(n);
            if (path)
                snprintf(path, n, "%s/%s", dir, e->d_name);
            break;
        }
    }
    closedir(d);
    return path;
}

static unsigned be16(const uint8_t *p) { return ((unsigned)p[0] << 8) | p[1]; }
static unsigned be32(const uint8_t *p)
{
    return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3];
}

static uint8_t *slurp(const char *path, size_t *size)
{
    FILE *f = fopen(path, "rb");
    uint8_t *buf = NULL;
    long n;
    if (!f)
        return NULL;
    if (fseek(f, 0, SEEK_END) == 0 && (n = ftell(f)) >= 0 && fseek(f, 0, SEEK_SET) == 0) {
        buf = malloc((size_t)n + 1);
        if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) {
            free(buf);
            buf = NULL;
        }
        *size = (size_t)n;
    }
    fclose(f);
    return buf;
}

/* the titleset's AOB files as one run of sectors (src/aob.c:86-127, 177-216) */
struct aob_set {
    FILE *f[MAX_AOBS];
    unsigned sectors[MAX_AOBS];
    unsigned n, total;
};

static void aob_close_all(struct aob_set *a)
{
    for (unsigned i = 0; i < a->n; i++)
        fclose(a->f[i]);
    a->n = 0;
}

static void aob_open_all(struct aob_set *a, const char *dir, unsigned titleset)
{
    memset(a, 0, sizeof(*a));
    for (unsigned k = 1; k <= MAX_AOBS; k++) {
        char name[16];
        char *path;
        struct stat st;
        snprintf(name, sizeof(name), "ATS_%2.2u_%1.1u.AOB", titleset % 100, k);
        path = find_file(dir, name);
        if (!path)
            break;
        if (stat(path, &st) != 0 || (a->f[a->n] = fopen(path, "rb")) == NULL) {
            free(path);
            break;
        }
        free(path);
        a->sectors[a->n] = (unsigned)(st.st_size / SECTOR);
        a->total += a->sectors[a->n];
        a->n++;
    }
}

/* reads sectors [first, first + count) into dst; returns the number read (short at the end) */
static unsigned aob_read(struct aob_set *a, unsigned first, unsigned count, uint8_t *dst)
{
    unsigned done = 0, base = 0;
    for (unsigned i = 0; i < a->n && done < count; i++) {
        const unsigned lo = base, hi = base + a->sectors[i];
        base = hi;
        if (first + done >= hi)
            continue;
        const unsigned at = first + done - lo;
        unsigned take = a->sectors[i] - at;
        if (take > count - done)
            take = count - done;
        if (fseek(a->f[i], (long)at * (long)SECTOR, SEEK_SET) != 0)
            break;
        const size_t got = fread(dst + (size_t)done * SECTOR, SECTOR, take, a->f[i]);
        done += (unsigned)got;
        if (got != take)
            break;
    }
    return done;
}

/* ------------------------------------------------------------------ disc / titleset / title / track */
DVDA *dvda_open(const char *audio_ts_path, const char *device)
{
    (void)device;               /* CPPM is not handled */
    if (!audio_ts_path)
        return NULL;
    char *ifo = find_file(audio_ts_path, "AUDIO_TS.IFO");
    if (!ifo)
        return NULL;
    size_t n = 0;
    uint8_t *b = slurp(ifo, &n);
    free(ifo);
    /* "DVDAUDIO-AMG", title set count in byte 63; the reference parses 104 bytes (src/dvd-audio.c:908-913) */
    unsigned count = 0;
    if (b && n >= 104 && memcmp(b, "DVDAUDIO-AMG", 12) == 0)
        count = b[63];
    free(b);
    if (!count)
        return NULL;
    DVDA *d = calloc(1, sizeof(*d));
    if (!d)
        return NULL;
    d->dir = strdup(audio_ts_path);
    d->titlesets = count;
    return d;
}

void dvda_close(DVDA *d)
{
    if (d) {
        free(d->dir);
        free(d);
    }
}

unsigned dvda_titleset_count(const DVDA *d) { return d->titlesets; }

/* one title table of ATS_XX_0.IFO (src/dvd-audio.c:975-1014); returns 0 when it leaves the file */
static int parse_title(const uint8_t *b, size_t n, size_t table, struct ifo_title *t)
{
    if (table + 16 > n)
        return 0;
    t->track_count = b[table + 2];
    t->index_count = b[table + 3];
    t->pts_length = be32(b + table + 4);
    const unsigned ptr_off = be16(b + table + 12);
    size_t p = table + 16;
    for (unsigned i = 0; i < t->track_count; i++, p += 20) {
        if (p + 20 > n)
            return 0;
        t->track[i].index_number = b[p + 4];
        t->track[i].pts_index = be32(b + p + 6);
        t->track[i].pts_length = be32(b + p + 10);
    }
    p = table + ptr_off;
    for (unsigned i = 0; i < t->index_count; i++, p += 12) {
        if (p + 12 > n)
            return 0;
        t->index[i].first = be32(b + p + 4);
        t->index[i].last = be32(b + p + 8);
    }
    return 1;
}

DVDA_Titleset *dvda_open_titleset(DVDA *d, unsigned titleset)
{
    char name[16];
    snprintf(name, sizeof(name), "ATS_%2.2u_0.IFO", titleset > 99 ? 99u : titleset);
    char *path = find_file(d->dir, name);
    if (!path)
        return NULL;
    size_t n = 0;
    uint8_t *b = slurp(path, &n);
    free(path);
    if (!b)
        return NULL;
    DVDA_Titleset *ts = NULL;
    if (n >= SECTOR + 8 && memcmp(b, "DVDAUDIO-ATS", 12) == 0) {
        ts = calloc(1, sizeof(*ts));
        ts->number = titleset;
        ts->title_count = be16(b + SECTOR);
        ts->title = calloc(ts->title_count ? ts->title_count : 1, sizeof(*ts->title));
        int ok = ts->title != NULL;
        for (unsigned i = 0; ok && i < ts->title_count; i++) {
            const size_t e = SECTOR + 8 + (size_t)8 * i;       /* title number 8u, 24p, table offset 32u */
            ok = e + 8 <= n && parse_title(b, n, SECTOR + (size_t)be32(b + e + 4), &ts->title[i]);
        }
        if (!ok) {
            free(ts->title);
            free(ts);
            ts = NULL;
        } else {
            ts->dir = strdup(d->dir);
        }
    }
    free(b);
    return ts;
}

void dvda_close_titleset(DVDA_Titleset *ts)
{
    if (ts) {
        free(ts->dir);
        free(ts->title);
        free(ts);
    }
}

unsigned dvda_titleset_number(const DVDA_Titleset *ts) { return ts->number; }
unsigned dvda_title_count(const DVDA_Titleset *ts) { return ts->title_count; }

static const unsigned *index_of(const struct ifo_title *t, unsigned track, int want_last)
{
    static const unsigned zero = 0;
    const unsigned k = t->track[track].index_number;
    if (k == 0 || k > 256)
        return &zero;
    return want_last ? &t->index[k - 1].last : &t->index[k - 1].first;
}

DVDA_Title *dvda_open_title(DVDA_Titleset *ts, unsigned title)
{
    if (title == 0 || title > ts->title_count)
        return NULL;
    const struct ifo_title *it = &ts->title[title - 1];
    DVDA_Title *t = calloc(1, sizeof(*t));
    if (!t)
        return NULL;
    t->dir = strdup(ts->dir);
    t->titleset = ts->number;
    t->number = title;
    t->track_count = it->track_count;
    t->pts_length = it->pts_length;
    for (unsigned i = 0; i < it->track_count; i++) {
        t->t[i].pts_index = it->track[i].pts_index;
        t->t[i].pts_length = it->track[i].pts_length;
        t->t[i].first = *index_of(it, i, 0);
        const unsigned own_last = *index_of(it, i, 1);
        /* a track runs up to the sector before the next track (of this or the next title);
         * only the very last one ends where its own index says (src/dvd-audio.c:452-488) */
        if (i + 1 < it->track_count) {
            t->t[i].last = *index_of(it, i + 1, 0) - 1;
        } else if (title < ts->title_count && ts->title[title].track_count) {
            const unsigned next_first = *index_of(&ts->title[title], 0, 0) - 1;
            t->t[i].last = next_first > own_last ? next_first : own_last;
        } else {
            t->t[i].last = own_last;
        }
    }
    return t;
}

void dvda_close_title(DVDA_Title *t)
{
    if (t) {
        free(t->dir);
        free(t);
    }
}

unsigned dvda_title_number(const DVDA_Title *t) { return t->number; }
unsigned dvda_track_count(const DVDA_Title *t) { return t->track_count; }
unsigned dvda_title_pts_length(const DVDA_Title *t) { return t->pts_length; }

DVDA_Track *dvda_open_track(DVDA_Title *t, unsigned track)
{
    if (track == 0 || track > t->track_count)
        return NULL;
    DVDA_Track *k = calloc(1, sizeof(*k));
    if (!k)
        return NULL;
    k->dir = strdup(t->dir);
    k->titleset = t->titleset;
    k->title = t->number;
    k->number = track;
    k->s = t->t[track - 1];
    return k;
}

void dvda_close_track(DVDA_Track *k)
{
    if (k) {
        free(k->dir);
        free(k);
    }
}

unsigned dvda_track_number(const DVDA_Track *k) { return k->number; }
unsigned dvda_track_pts_index(const DVDA_Track *k) { return k->s.pts_index; }
unsigned dvda_track_pts_length(const DVDA_Track *k) { return k->s.pts_length; }
unsigned dvda_track_first_sector(const DVDA_Track *k) { return k->s.first; }
unsigned dvda_track_last_sector(const DVDA_Track *k) { return k->s.last; }

/* ------------------------------------------------------------------ code tables (src/dvd-audio.c:1423-1496) */
static unsigned bits_of(unsigned code) { return code == 0 ? 16 : code == 1 ? 20 : code == 2 ? 24 : 0; }

static unsigned rate_of(unsigned code)
{
    switch (code) {
    case 0: return 48000;
    case 1: return 96000;
    case 2: return 192000;
    case 8: return 44100;
    case 9: return 88200;
    case 10: return 176400;
    default: return 0;
    }
}

static unsigned channels_of(unsigned assignment)
{
    static const uint8_t n[21] = {1, 2, 3, 4, 3, 4, 5, 3, 4, 5, 4, 5, 6, 4, 5, 4, 5, 6, 5, 5, 6};
    return assignment < 21 ? n[assignment] : 0;
}

/* ------------------------------------------------------------------ codec probe */
/* First 0xBD packet of a sector: returns 1 and its codec id, pad_2 size and the bytes behind the
 * 4-byte audio header; 0 = no audio packet in this sector; -1 = malformed (src/packet.c:61-188,
 * src/dvd-audio.c:1238-1248). */
static int first_audio_packet(const uint8_t *p, unsigned *codec, unsigned *pad2, const uint8_t **body,
                              unsigned *body_len)
{
    if (p[0] != 0 || p[1] != 0 || p[2] != 1 || p[3] != 0xBA)
        return -1;
    if ((p[4] >> 6) != 1 || !(p[4] & 4) || !(p[6] & 4) || !(p[8] & 4) || !(p[9] & 1) || (p[12] & 3) != 3)
        return -1;
    unsigned pos = 14 + (p[13] & 7);
    while (pos + 6 <= SECTOR) {
        const unsigned id = p[pos + 3], len = be16(p + pos + 4);
        if (p[pos] != 0 || p[pos + 1] != 0 || p[pos + 2] != 1 || pos + 6 + len > SECTOR)
            return -1;
        if (id == 0xBD) {
            const uint8_t *q = p + pos + 6;
            if (len < 3 || len < 7u + q[2])
                return -1;
            const unsigned pad1 = q[2];
            *codec = q[3 + pad1];
            *pad2 = q[6 + pad1];
            *body = q + 7 + pad1;
            *body_len = len - 7 - pad1;
            return 1;
        }
        pos += 6 + len;
    }
    return 0;
}

/* ------------------------------------------------------------------ device helpers */
/* DVDA_DISC_TIMING=1: where opening an MLP track spends its time, on stderr (tools/disc_bench.py; diagnostic) */
#include <time.h>
static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
static int disc_timing(void)
{
    static int on = -1;
    if (on < 0)
        on = getenv("DVDA_DISC_TIMING") != NULL;
    return on;
}
#define T_MARK(what)                                                             \
    do {                                                                         \
        if (disc_timing()) {                                                     \
            (void)hipDeviceSynchronize();                                        \
            const double t_ = now_ms();                                          \
            fprintf(stderr, "  [disc] %-28s %8.2f ms\n", what, t_ - t_mark);     \
            t_mark = t_;                                                         \
        }                                                                        \
    } while (0)

static int dev_alloc(void **p, size_t bytes)
{
    *p = NULL;
    return hipMalloc(p, bytes ? bytes : 16) == hipSuccess;
}

/* offset of the first major-sync pattern (bytes +4..+7 = F8 72 6F BB) at or after `from` whose
 * 8 bytes lie inside [0, size), or -1 (find_major_sync, src/dvd-audio.c:1250-1285) */
static int64_t find_sync(const uint8_t *b, uint64_t from, uint64_t size)
{
    for (uint64_t p = from; p + 8 <= size; p++)
        if (b[p + 4] == 0xF8 && b[p + 5] == 0x72 && b[p + 6] == 0x6F && b[p + 7] == 0xBB)
            return (int64_t)p;
    return -1;
}

/* the same search over device memory: windows of the payload are copied back until the pattern
 * shows up (each window overlaps the previous one by 7 bytes) */
static int64_t find_sync_dev(const uint8_t *d_bytes, uint64_t from, uint64_t size)
{
    enum { WINDOW = 1 << 16 };
    uint8_t *w = malloc(WINDOW);
    int64_t at = -1;
    if (!w)
        return -1;
    while (from + 8 <= size) {
        const uint64_t n = size - from < WINDOW ? size - from : WINDOW;
        if (hipMemcpy(w, d_bytes + from, n, hipMemcpyDeviceToHost) != hipSuccess)
            break;
        const int64_t p = find_sync(w, 0, n);
        if (p >= 0) {
            at = (int64_t)from + p;
            break;
        }
        if (n < WINDOW)
            break;
        from += n - 7;
    }
    free(w);
    return at;
}

static void windows_free(struct mlp_windows *w);

static void reader_free(DVDA_Track_Reader *r)
{
    if (!r)
        return;
    windows_free(r->win);
    free(r->pcm);
    (void)hipFree(r->d_wav);
    free(r->wav);
    if (r->d_pcm)
        (void)hipFree(r->d_pcm);
    free(r);
}

/* ------------------------------------------------------------------ MLP track */
static DVDA_Track_Reader *open_mlp(struct aob_set *aobs, const DVDA_Track *k)
{
    DVDA_Track_Reader *r = NULL;
    uint8_t *h_sec = NULL, *d_sec = NULL, *d_mlp = NULL, *d_stream = NULL;
    uint32_t *d_work = NULL, *h_base = NULL;
    uint64_t *d_meta = NULL;
    int32_t *d_pcm = NULL;
    dvda_mlp_hip_ctx *ctx = NULL;
    const unsigned first = k->s.first;
    const unsigned in_track = k->s.last >= first ? k->s.last - first + 1 : 1;
    unsigned extra = 8;
    uint64_t total = 0, begin = 0, end = 0;
    double t_mark = now_ms();

    for (;;) {
        /* sectors of the track plus a few behind it: the stream runs on to the next major sync */
        unsigned want = in_track + extra;
        if (first + want > aobs->total || first + want < first)
            want = aobs->total - first;
        free(h_sec);
        free(h_base);
        h_sec = NULL;
        h_base = NULL;
        (void)hipFree(d_sec);
        (void)hipFree(d_mlp);
        (void)hipFree(d_work);
        d_sec = d_mlp = NULL;
        d_work = NULL;
        h_sec = malloc((size_t)want * SECTOR);
        if (!h_sec)
            goto fail;
        const unsigned got = aob_read(aobs, first, want, h_sec);
        if (got == 0)
            goto fail;
        T_MARK("sectors read from the files");
        const size_t cap = (size_t)got * SECTOR;
        if (!dev_alloc((void **)&d_sec, cap) || !dev_alloc((void **)&d_mlp, cap + 64) ||
            !dev_alloc((void **)&d_work, dvda_pcm_hip_workspace_words(got) * sizeof(uint32_t)))
            goto fail;
        if (hipMemcpy(d_sec, h_sec, cap, hipMemcpyHostToDevice) != hipSuccess)
            goto fail;
        if (dvda_mlp_hip_demux_sectors(d_sec, got, d_mlp, cap, d_work, NULL) != DVDA_HIP_OK)
            goto fail;
        uint32_t bad = 0;
        if (dvda_pcm_hip_result(d_work, got, &total, &bad, NULL) != DVDA_HIP_OK)
            goto fail;
        /* (`bad` also counts the sectors of a following track of the other codec inside the look-ahead window:
           they contribute no bytes, which is what is wanted of them.  A sector with more audio packets than
           the kernels keep is refused by scan and gather alike, csrc/pcm_unswizzle.h.) */
        (void)bad;
        /* workspace words [got, 2*got]: payload offset of every sector, then the total */
        h_base = malloc(((size_t)got + 1) * sizeof(uint32_t));
        if (!h_base)
            goto fail;
        if (hipMemcpy(h_base, d_work + got, ((size_t)got + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
            goto fail;
        T_MARK("to the device + demux");
        const int64_t s0 = find_sync_dev(d_mlp, 0, total);
        if (s0 < 0) {
            if (got < want || first + got >= aobs->total)
                goto fail;                       /* no major sync anywhere: the reference asserts */
            extra *= 4;
            continue;
        }
        begin = (uint64_t)s0;
        if (got <= in_track) {
            end = total;                         /* the titleset ends with this track */
            break;
        }
        const uint64_t boundary = h_base[in_track];
        const int64_t s1 = find_sync_dev(d_mlp, boundary > begin ? boundary : begin, total);
        if (s1 >= 0) {
            end = (uint64_t)s1;
            break;
        }
        if (first + got >= aobs->total) {
            /* packets behind the track but no further sync: the reference gives up 7 bytes
             * short of the data it has (find_major_sync needs 8) */
            end = total - boundary >= 8 ? total - 7 : boundary;
            break;
        }
        extra *= 4;
    }
    if (end <= begin)
        goto fail;
    T_MARK("major syncs at both ends");

    {
        const uint64_t len = end - begin;
        const uint64_t padded = (len + 15) & ~(uint64_t)15;
        uint64_t meta[4] = {0, len, 0, 0};
        dvda_mlp_stream_info info;
        uint32_t segs = (uint32_t)(len / 2048 + 256), found = 0;
        if (!dev_alloc((void **)&d_stream, padded + 64) || !dev_alloc((void **)&d_meta, sizeof(meta)))
            goto fail;
        if (hipMemset(d_stream, 0, padded + 64) != hipSuccess ||
            hipMemcpy(d_stream, d_mlp + begin, len, hipMemcpyDeviceToDevice) != hipSuccess ||
            hipMemcpy(d_meta, meta, sizeof(meta), hipMemcpyHostToDevice) != hipSuccess)
            goto fail;
        (void)hipFree(d_sec);
        (void)hipFree(d_mlp);
        d_sec = d_mlp = NULL;
        for (int attempt = 0; attempt < 2; attempt++) {
            if (dvda_mlp_hip_create(&ctx, g_device, 1, segs) != DVDA_HIP_OK ||
                dvda_mlp_hip_set_pcm_layout(ctx, DVDA_PCM_INTERLEAVED) != DVDA_HIP_OK)
                goto fail;
            if (dvda_mlp_hip_index(ctx, d_stream, padded, d_meta + 0, d_meta + 1, 1, NULL) != DVDA_HIP_OK)
                goto fail;
            const int rc = dvda_mlp_hip_segment_count(ctx, &found, NULL);
            if (rc == DVDA_HIP_OK)
                break;
            if (rc != DVDA_HIP_ECAPACITY || attempt)
                goto fail;
            dvda_mlp_hip_destroy(ctx);
            ctx = NULL;
            segs = found + 16;
        }
        if (dvda_mlp_hip_stream_info(ctx, &info, 1, NULL) != DVDA_HIP_OK || info.channels == 0)
            goto fail;
        T_MARK("context + index");
        const unsigned rate = rate_of(info.group0_rate);
        const uint64_t per_au = rate == 48000 || rate == 44100 ? 40 : rate == 96000 || rate == 88200 ? 80 : 160;
        uint64_t stride = info.mlp_frames * per_au;
        stride = (stride + 3) & ~(uint64_t)3;
        if (stride == 0)
            stride = 4;
        /* dvda_hip_set_wav_output(1): the decode writes the payload dvda2wav would write -- interleaved, little-endian,
         * write_signed at the stream's own bit depth -- and nothing else: no int32 PCM, no separate packing pass
         * (SURVEY 8(f-3) fused into the decode, DVDA_PCM_WAV24 / WAV16) */
        const unsigned wbits = g_wav_output ? bits_of(info.group0_bps) : 0;
        const int direct = wbits == 16 || wbits == 24;
        if (direct && dvda_mlp_hip_set_pcm_layout(ctx, wbits == 24 ? DVDA_PCM_WAV24 : DVDA_PCM_WAV16) != DVDA_HIP_OK)
            goto fail;
        for (int attempt = 0; attempt < 2; attempt++) {
            meta[3] = stride;
            /* (as packed bytes the track takes 3/4 or 1/2 of the int32 words; the buffer is sized in int32 words
             *  either way: the decode's offsets are in those units) */
            const uint64_t words = direct ? (stride * info.channels * (wbits / 8) + 3) / 4 + 4 : stride * info.channels;
            if (!dev_alloc((void **)&d_pcm, words * sizeof(int32_t)) ||
                hipMemcpy(d_meta, meta, sizeof(meta), hipMemcpyHostToDevice) != hipSuccess)
                goto fail;
            if (dvda_mlp_hip_decode(ctx, d_pcm, d_meta + 2, d_meta + 3, NULL) != DVDA_HIP_OK ||
                dvda_mlp_hip_stream_info(ctx, &info, 1, NULL) != DVDA_HIP_OK)
                goto fail;
            if (!(info.status & DVDA_ST_OVERFLOW))
                break;
            /* access units longer than the standard length: pcm_frames is the size needed */
            if (attempt)
                goto fail;
            (void)hipFree(d_pcm);
            d_pcm = NULL;
            stride = (info.pcm_frames + 3) & ~(uint64_t)3;
        }
        T_MARK("decode");
        if (info.status & ~(uint32_t)DVDA_ST_BENIGN)
            goto fail;                       /* the reference assert()s on such a stream */
        r = calloc(1, sizeof(*r));
        if (!r)
            goto fail;
        r->codec = DVDA_MLP;
        r->bps_code[0] = info.group0_bps;
        r->bps_code[1] = info.group1_bps;
        r->rate_code[0] = info.group0_rate;
        r->rate_code[1] = info.group1_rate;
        r->assignment = info.assignment;
        r->channels = channels_of(info.assignment);
        r->status = info.status;
        r->frames = info.pcm_frames;
        r->stride = stride;
        r->interleaved = 1;            /* decoded frame-major: dvda_read() copies frames straight out */
        if (r->channels == 0 || r->channels != info.channels)
            goto fail;
        if (direct) {
            r->d_wav = (uint8_t *)d_pcm;
            r->wav_bytes = r->frames * r->channels * (wbits / 8);
        } else {
            r->d_pcm = d_pcm;          /* the host copy is made by the first dvda_read() */
        }
        d_pcm = NULL;
    }
    goto done;
fail:
    reader_free(r);
    r = NULL;
done:
    if (ctx)
        dvda_mlp_hip_destroy(ctx);
    free(h_sec);
    free(h_base);
    (void)hipFree(d_sec);
    (void)hipFree(d_mlp);
    (void)hipFree(d_work);
    (void)hipFree(d_stream);
    (void)hipFree(d_meta);
    (void)hipFree(d_pcm);
    return r;
}

/* ------------------------------------------------------------------ MLP track, in windows (round 5)
 *
 * The reference streams a track of any length in O(packet) memory (src/dvd-audio.c:751-795, 1151-1227).  open_mlp()
 * above takes the whole track as ONE batch -- its sectors, its bytes and its PCM all resident at once: a 74-minute
 * 6-channel track is ~10 GB on the device and again on the host.  A track of more than WINDOW_SECTORS sectors is
 * therefore read and decoded window by window:
 *
 *   window = the next WINDOW_SECTORS sectors of the track -> GPU demux -> [bytes kept from the window before | new bytes]
 *   cut    = the LAST major sync of that stream at which a segment starts AND every substream restarts
 *            (win_unit_restarts: a window cannot begin at a sync that restarts nothing) (the index says where: restart segments
 *            are the units of parallel decode): everything in front of it is whole segments and is decoded
 *            now; what follows is kept for the next window.  The first window starts at the first major-sync
 *            pattern, the last one ends as open_mlp()'s whole track does (src/dvd-audio.c:1167-1194).
 *   carry  = between two windows nothing but those bytes and the FIR history at the cut (the reference never clears a
 *            channel's history, src/mlp.c:297-304: dvda_mlp_hip_segment_fir / dvda_mlp_hip_set_initial_fir) -- every
 *            other decoder field is set again by the restart header at a major sync.
 *
 * A producer thread fills a ring of two pinned host buffers (window k + 1 is read, demultiplexed and decoded while
 * dvda_read() / dvda_hip_reader_wav_next() serve window k); one decode context, one set of device buffers and the two
 * pinned buffers serve the whole track.  What is resident is bounded by the window, not by the track:
 * dvda_hip_reader_memory() reports the peaks (tests/test_disc_api.py asserts them).
 */
#define WIN_SLOTS 2
static unsigned window_sectors(void)
{
    /* DVDA_WINDOW_SECTORS: window size in 2048-byte sectors (tests use small ones); default 8192 = 16 MiB of sectors,
       which decode to 32..70 MB of PCM */
    const char *e = getenv("DVDA_WINDOW_SECTORS");
    const long v = e ? strtol(e, NULL, 10) : 0;
    return v >= 64 ? (unsigned)v : 8192u;
}

struct win_slot {
    uint8_t *host;              /* pinned: int32 frames [frame][channel], or the packed WAV payload */
    size_t cap;
    uint64_t frames;
    uint64_t stride;            /* != 0: int32 PLANAR [channel][stride] (raw-PCM windows: the order the un-swizzle writes) */
};

struct mlp_windows {
    struct aob_set aobs;        /* the reader's own open files */
    unsigned first, last;       /* the track's sector range */
    unsigned next;              /* next sector to read */
    unsigned window;
    int device, wav_bits;       /* wav_bits != 0: the decode writes the payload (DVDA_PCM_WAV24 / WAV16) */
    int wav_request, wav_decided;   /* the opener asked for the payload; decided (from the first window's major sync:
                                       only a 16- or 24-bit stream is decoded straight into it) */
    int started, finished, failed;
    /* a raw-PCM track read in windows (round 6): sectors decode independently of each other (src/pcm.c:149: whole chunks
       per packet), so a window is a run of sectors and nothing crosses a cut but the count of frames delivered so far */
    int is_pcm;
    unsigned pcm_bits, pcm_channels;
    uint64_t pcm_want;          /* the track's length in PCM frames (its PTS length); whole packets until it is covered */
    uint64_t pcm_done;
    uint8_t *pack_tmp;          /* host: a planar window packed for dvda_hip_reader_wav_next() on an int32 reader */
    size_t pack_cap;
    uint8_t *carry;             /* host: bytes from the last cut on */
    size_t carry_len, carry_cap;
    int32_t fir[2 * 48];
    int have_fir;
    /* device side, kept for the whole track and grown when a window needs more */
    dvda_mlp_hip_ctx *ctx;
    uint32_t ctx_segs;
    uint8_t *h_sec, *d_sec, *d_mlp, *d_stream;
    uint32_t *d_work, *h_base;
    uint64_t *d_meta;
    int32_t *d_pcm, *d_fir;
    size_t cap_sec, cap_stream, cap_pcm;        /* sectors, bytes, bytes */
    /* what the stream is (first window) */
    dvda_mlp_stream_info info;
    unsigned status;            /* (with frames_total and `finished`: published under `mu`, win_commit()) */
    uint64_t frames_total;
    /* what win_produce() found, private to the thread that runs it until win_commit() publishes it under `mu` -- in the
       same critical section that queues the window (round 6: `finished` used to be set by win_produce itself, without
       the lock, BEFORE the producer queued the last window: a consumer that looked in between saw "no window, finished"
       and reported the end of the track with the last window still unqueued) */
    int p_final;
    unsigned p_status;
    uint64_t p_frames;
    /* producer / consumer */
    pthread_t th;
    int th_started, stop;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    struct win_slot slot[WIN_SLOTS];
    unsigned head, tail, count;
    uint64_t served_in_slot;    /* frames of slot[tail] already handed out */
    uint8_t *whole;             /* dvda_hip_reader_wav_payload() on a windowed reader: every window appended (unbounded) */
    /* accounting */
    size_t host_now, host_peak, dev_free0, dev_peak;
    size_t dev_base;            /* device memory the reader took over from the thread's cache (already allocated at dev_free0) */
};

static void win_host_add(struct mlp_windows *w, size_t now_more)
{
    w->host_now += now_more;
    if (w->host_now > w->host_peak)
        w->host_peak = w->host_now;
}
static void win_dev_sample(struct mlp_windows *w)
{
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) == hipSuccess && w->dev_free0 > fr && w->dev_base + (w->dev_free0 - fr) > w->dev_peak)
        w->dev_peak = w->dev_base + (w->dev_free0 - fr);
}
static int win_grow_dev(void **p, size_t *cap, size_t need)
{
    if (need <= *cap && *p)
        return 1;
    (void)hipFree(*p);
    *p = NULL;
    *cap = 0;
    need += need / 4 + 4096;
    if (hipMalloc(p, need) != hipSuccess)
        return 0;
    *cap = need;
    return 1;
}

/* What a windowed reader allocated -- the decode context, the device buffers, the pinned sector buffer and window slots --
 * kept by the thread that closes it for the next windowed reader the same thread opens on the same device (round 5: a
 * worker of dvda2wav_hip extracts track after track; allocating and freeing all of it was 50 of a track's 90 ms,
 * tools/probe/r05_disc_time.sh).  One set per thread; dvda_hip_release_cached_buffers() frees the caller's. */
struct win_cache {
    int valid, device;
    dvda_mlp_hip_ctx *ctx;
    uint32_t ctx_segs;
    uint8_t *h_sec, *d_sec, *d_mlp, *d_stream, *carry;
    uint32_t *d_work, *h_base;
    uint64_t *d_meta;
    int32_t *d_pcm, *d_fir;
    size_t cap_sec, cap_stream, cap_pcm, carry_cap, host_bytes, dev_bytes;
    struct win_slot slot[WIN_SLOTS];
};
static __thread struct win_cache t_win_cache;

static void win_cache_free(struct win_cache *c)
{
    if (!c->valid)
        return;
    (void)hipSetDevice(c->device);
    if (c->ctx)
        dvda_mlp_hip_destroy(c->ctx);
    for (int i = 0; i < WIN_SLOTS; i++)
        if (c->slot[i].host)
            (void)hipHostFree(c->slot[i].host);
    if (c->h_sec)
        (void)hipHostFree(c->h_sec);
    free(c->h_base);
    free(c->carry);
    (void)hipFree(c->d_sec);
    (void)hipFree(c->d_mlp);
    (void)hipFree(c->d_stream);
    (void)hipFree(c->d_work);
    (void)hipFree(c->d_meta);
    (void)hipFree(c->d_pcm);
    (void)hipFree(c->d_fir);
    memset(c, 0, sizeof(*c));
}

void dvda_hip_release_cached_buffers(void) { win_cache_free(&t_win_cache); }

/* ... and a thread that ends without calling it does not keep them: a key whose destructor frees the thread's set */
static pthread_key_t g_win_cache_key;
static pthread_once_t g_win_cache_once = PTHREAD_ONCE_INIT;
static void win_cache_at_thread_exit(void *p)
{
    win_cache_free((struct win_cache *)p);
}
static void win_cache_make_key(void) { (void)pthread_key_create(&g_win_cache_key, win_cache_at_thread_exit); }
static void win_cache_arm(struct win_cache *c)
{
    (void)pthread_once(&g_win_cache_once, win_cache_make_key);
    (void)pthread_setspecific(g_win_cache_key, c);
}

static void windows_free(struct mlp_windows *w)
{
    if (!w)
        return;
    if (w->th_started) {
        pthread_mutex_lock(&w->mu);
        w->stop = 1;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
        pthread_join(w->th, NULL);
    }
    /* the buffers go to this thread's cache (whatever it held before is freed: the newer set fits the newer tracks) */
    struct win_cache *c = &t_win_cache;
    win_cache_free(c);
    c->valid = 1;
    win_cache_arm(c);
    c->device = w->device;
    c->ctx = w->ctx;
    c->ctx_segs = w->ctx_segs;
    c->h_sec = w->h_sec;
    c->d_sec = w->d_sec;
    c->d_mlp = w->d_mlp;
    c->d_stream = w->d_stream;
    c->carry = w->carry;
    c->d_work = w->d_work;
    c->h_base = w->h_base;
    c->d_meta = w->d_meta;
    c->d_pcm = w->d_pcm;
    c->d_fir = w->d_fir;
    c->cap_sec = w->cap_sec;
    c->cap_stream = w->cap_stream;
    c->cap_pcm = w->cap_pcm;
    c->carry_cap = w->carry_cap;
    c->host_bytes = w->host_now;
    c->dev_bytes = w->dev_peak;
    for (int i = 0; i < WIN_SLOTS; i++) {
        c->slot[i] = w->slot[i];
        c->slot[i].frames = 0;
    }
    free(w->whole);
    free(w->pack_tmp);
    aob_close_all(&w->aobs);
    pthread_mutex_destroy(&w->mu);
    pthread_cond_destroy(&w->cv);
    free(w);
}

/* Does every substream of the sync unit at `off` of the stream [0, len) in d_stream open with a restart header?  (Reference:
 * decode_block src/mlp.c:748-753 -- two flags in front of a block, "parameters present" and "restart header present";
 * the unit's layout: 4 bytes of frame header, the 28-byte major sync with substream_count 128 bits in, src/mlp.c:621-632,
 * one directory word per substream and one more behind it when its top bit is set, src/mlp.c:463-468, 661-667.)
 * 1 = yes, 0 = no, -1 = the device read failed.  A window may begin at such a unit: the restart header sets every
 * parameter the lanes do not carry across a cut, the FIR history is carried (win_produce). */
static int win_unit_restarts(struct mlp_windows *w, uint64_t off, uint64_t len)
{
    uint8_t h[4 + 28 + 8 + 1];
    const uint64_t have = len - off < sizeof(h) ? len - off : sizeof(h);
    if (have < 4 + 28 + 2 + 1)
        return 0;
    if (hipMemcpy(h, w->d_stream + off, have, hipMemcpyDeviceToHost) != hipSuccess)
        return -1;
    if (h[4] != 0xF8 || h[5] != 0x72 || h[6] != 0x6F || h[7] != 0xBB)
        return 0;
    const unsigned S = h[4 + 16] >> 4;
    if (S != 1 && S != 2)
        return 0;
    uint64_t p = 4 + 28, end0 = 0;
    for (unsigned s = 0; s < S; s++) {
        if (p + 2 > have)
            return 0;
        const unsigned e = ((unsigned)h[p] << 8) | h[p + 1];
        if (s == 0)
            end0 = (uint64_t)(e & 0xFFFu) * 2;
        p += (e & 0x8000u) ? 4 : 2;
    }
    if (p >= have || (h[p] & 0xC0) != 0xC0)
        return 0;
    if (S == 2) {
        uint8_t b = 0;
        if (off + p + end0 >= len)
            return 0;
        if (hipMemcpy(&b, w->d_stream + off + p + end0, 1, hipMemcpyDeviceToHost) != hipSuccess)
            return -1;
        if ((b & 0xC0) != 0xC0)
            return 0;
    }
    return 1;
}

/* the last segment of the index (a live one: a sync pattern inside another segment's frames is not a cut) -> its number
 * and byte offset; 0 when the stream has no segment start behind its first byte, -1 when a device call failed (that is
 * not "no cut": taken as one, the carried bytes would grow window after window).  restart_len != 0: the last one a
 * window can BEGIN at -- a major sync does not oblige the substreams to restart (the reference reads its parameters and
 * decodes on, src/mlp.c:449-460), and a lane that starts there would have no parameters; such a unit stays inside its
 * window, where the sequential pass reaches it with the state of the units before it.  restart_len = the stream's length. */
static int win_last_cut(struct mlp_windows *w, uint32_t n_seg, uint64_t restart_len, uint32_t *seg_out, uint64_t *off_out)
{
    for (uint32_t s = n_seg; s-- > 0;) {
        dvda_mlp_segment_info si;
        if (dvda_mlp_hip_segment_info(w->ctx, s, &si, NULL) != DVDA_HIP_OK)
            return -1;
        if (si.status & DVDA_ST_FALSE_SYNC)
            continue;
        if (restart_len && si.offset != 0) {
            const int r = win_unit_restarts(w, si.offset, restart_len);
            if (r < 0)
                return -1;
            if (!r)
                continue;
        }
        *seg_out = s;
        *off_out = si.offset;
        return si.offset != 0;
    }
    return 0;
}

/* index of stream bytes [0, len) in d_stream; grows the context when it has too few segments */
static int win_index(struct mlp_windows *w, uint64_t len, uint32_t *n_seg)
{
    const uint64_t padded = (len + 15) & ~(uint64_t)15;
    const uint64_t meta[4] = {0, len, 0, 0};
    if (hipMemsetAsync(w->d_stream + len, 0, padded + 64 - len, NULL) != hipSuccess ||
        hipMemcpy(w->d_meta, meta, sizeof(meta), hipMemcpyHostToDevice) != hipSuccess)
        return 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        if (!w->ctx) {
            if (dvda_mlp_hip_create(&w->ctx, w->device, 1, w->ctx_segs) != DVDA_HIP_OK)
                return 0;
        }
        if (dvda_mlp_hip_index(w->ctx, w->d_stream, padded, w->d_meta + 0, w->d_meta + 1, 1, NULL) != DVDA_HIP_OK)
            return 0;
        const int rc = dvda_mlp_hip_segment_count(w->ctx, n_seg, NULL);
        if (rc == DVDA_HIP_OK)
            return 1;
        if (rc != DVDA_HIP_ECAPACITY || attempt)
            return 0;
        dvda_mlp_hip_destroy(w->ctx);
        w->ctx = NULL;
        w->ctx_segs = *n_seg + *n_seg / 2 + 64;
    }
    return 0;
}

/* One window into `out`: 1 = out holds frames (possibly none), 0 = failure.  Leaves "this was the track's last window",
 * the window's status bits and its frame count in w->p_*: the caller publishes them (win_commit, under w->mu once a
 * consumer exists).  Runs in the opener's thread for the first window, in the producer thread afterwards. */
static int win_produce(struct mlp_windows *w, struct win_slot *out)
{
    out->frames = 0;
    out->stride = 0;
    unsigned extra = 8;
    for (;;) {
        const int has_track = w->next <= w->last;
        unsigned in_track = has_track ? w->last - w->next + 1 : 0;
        int final = in_track <= w->window;
        if (!final)
            in_track = w->window;
        unsigned want = in_track + (final ? extra : 0);
        if (w->next + want > w->aobs.total || w->next + want < w->next)
            want = w->aobs.total - w->next;
        /* ---- sectors -> device -> MLP bytes */
        if (want > w->cap_sec) {
            if (w->h_sec) {
                (void)hipHostFree(w->h_sec);
                w->host_now -= w->cap_sec * SECTOR;
            }
            (void)hipFree(w->d_sec);
            (void)hipFree(w->d_mlp);
            (void)hipFree(w->d_work);
            free(w->h_base);
            w->h_sec = w->d_sec = w->d_mlp = NULL;
            w->d_work = w->h_base = NULL;
            w->cap_sec = 0;
            const size_t cap = (size_t)want * SECTOR;
            if (hipHostMalloc((void **)&w->h_sec, cap, hipHostMallocDefault) != hipSuccess)
                return 0;
            win_host_add(w, cap);
            if (!dev_alloc((void **)&w->d_sec, cap) || !dev_alloc((void **)&w->d_mlp, cap + 64) ||
                !dev_alloc((void **)&w->d_work, dvda_pcm_hip_workspace_words(want) * sizeof(uint32_t)) ||
                (w->h_base = malloc(((size_t)want + 1) * sizeof(uint32_t))) == NULL)
                return 0;
            w->cap_sec = want;
        }
        const unsigned got = want ? aob_read(&w->aobs, w->next, want, w->h_sec) : 0;
        uint64_t total = 0;
        uint32_t bad = 0;
        if (got) {
            if (hipMemcpy(w->d_sec, w->h_sec, (size_t)got * SECTOR, hipMemcpyHostToDevice) != hipSuccess ||
                dvda_mlp_hip_demux_sectors(w->d_sec, got, w->d_mlp, (size_t)w->cap_sec * SECTOR, w->d_work, NULL) != DVDA_HIP_OK ||
                dvda_pcm_hip_result(w->d_work, got, &total, &bad, NULL) != DVDA_HIP_OK ||
                hipMemcpy(w->h_base, w->d_work + got, ((size_t)got + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
                return 0;
        } else if (!final) {
            return 0;                            /* the files end inside the track */
        }
        /* ---- where this window's new bytes begin and end */
        uint64_t begin = 0, end = total;
        if (!w->started) {
            const int64_t s0 = find_sync_dev(w->d_mlp, 0, total);
            if (s0 < 0) {
                if (!final || got < want || w->next + got >= w->aobs.total)
                    return 0;                    /* no major sync anywhere: the reference asserts */
                extra *= 4;
                continue;
            }
            begin = (uint64_t)s0;
        }
        if (final) {
            /* the end-of-track rule of open_mlp(), on the window that holds the track's last sector */
            if (got > in_track) {
                const uint64_t boundary = w->h_base[in_track];
                const int64_t s1 = find_sync_dev(w->d_mlp, boundary > begin ? boundary : begin, total);
                if (s1 >= 0) {
                    end = (uint64_t)s1;
                } else if (w->next + got >= w->aobs.total) {
                    end = total - boundary >= 8 ? total - 7 : boundary;
                } else {
                    extra *= 4;
                    continue;
                }
            }
        }
        if (end < begin)
            end = begin;
        /* ---- the stream of this window: what was kept + the new bytes */
        uint64_t len = w->carry_len + (end - begin);
        if (!win_grow_dev((void **)&w->d_stream, &w->cap_stream, ((len + 15) & ~(uint64_t)15) + 64 + 16) ||
            (!w->d_meta && !dev_alloc((void **)&w->d_meta, 4 * sizeof(uint64_t))))
            return 0;
        if ((w->carry_len && hipMemcpy(w->d_stream, w->carry, w->carry_len, hipMemcpyHostToDevice) != hipSuccess) ||
            (end > begin && hipMemcpy(w->d_stream + w->carry_len, w->d_mlp + begin, end - begin, hipMemcpyDeviceToDevice) != hipSuccess))
            return 0;
        w->next += final ? got : in_track;
        w->started = 1;
        if (len == 0) {
            w->p_final = final;
            return final;                        /* an empty track does not open */
        }
        uint32_t n_seg = 0, cut_seg = 0;
        uint64_t cut = len;
        if (w->ctx_segs == 0)
            w->ctx_segs = (uint32_t)(len / 2048 + 256);
        if (!win_index(w, len, &n_seg))
            return 0;
        if (!final) {
            const int have_cut = win_last_cut(w, n_seg, len, &cut_seg, &cut);
            if (have_cut < 0)
                return 0;
            if (!have_cut) {
                /* no segment starts inside this window: all of it waits for the next one */
                cut = 0;
            }
            /* what follows the cut is kept (host copy: the device buffers are the next window's) */
            const uint64_t keep = len - cut;
            if (keep > w->carry_cap) {
                free(w->carry);
                w->host_now -= w->carry_cap;
                w->carry_cap = keep + keep / 2 + 4096;
                w->carry = malloc(w->carry_cap);
                if (!w->carry) {
                    w->carry_cap = 0;
                    return 0;
                }
                win_host_add(w, w->carry_cap);
            }
            if (keep && hipMemcpy(w->carry, w->d_stream + cut, keep, hipMemcpyDeviceToHost) != hipSuccess)
                return 0;
            w->carry_len = keep;
            if (cut == 0)
                return 1;                        /* nothing to decode yet (out->frames == 0) */
            if (!win_index(w, cut, &n_seg))      /* the index of what is decoded now: whole segments */
                return 0;
        } else {
            w->carry_len = 0;
        }
        /* ---- decode [0, cut) from the history the window before left */
        dvda_mlp_stream_info info;
        if (dvda_mlp_hip_stream_info(w->ctx, &info, 1, NULL) != DVDA_HIP_OK || info.channels == 0)
            return 0;
        if (w->have_fir) {
            if ((!w->d_fir && !dev_alloc((void **)&w->d_fir, sizeof(w->fir))) ||
                hipMemcpy(w->d_fir, w->fir, sizeof(w->fir), hipMemcpyHostToDevice) != hipSuccess)
                return 0;
        }
        if (dvda_mlp_hip_set_initial_fir(w->ctx, w->have_fir ? w->d_fir : NULL) != DVDA_HIP_OK)
            return 0;
        const unsigned rate = rate_of(info.group0_rate);
        const uint64_t per_au = rate == 48000 || rate == 44100 ? 40 : rate == 96000 || rate == 88200 ? 80 : 160;
        uint64_t stride = ((info.mlp_frames * per_au + 3) & ~(uint64_t)3);
        if (stride == 0)
            stride = 4;
        if (w->wav_request && !w->wav_decided) {
            /* the bit depth is in the index's stream record: known before anything is decoded (round 5: the first
               window was decoded twice, once as int32 to learn it) */
            const unsigned b = bits_of(info.group0_bps);
            w->wav_bits = (b == 16 || b == 24) ? (int)b : 0;
            w->wav_decided = 1;
        }
        const unsigned wbits = w->wav_bits;
        if (dvda_mlp_hip_set_pcm_layout(w->ctx, wbits == 24 ? DVDA_PCM_WAV24 : wbits == 16 ? DVDA_PCM_WAV16 : DVDA_PCM_INTERLEAVED) != DVDA_HIP_OK)
            return 0;
        for (int attempt = 0;; attempt++) {
            uint64_t meta[4] = {0, cut, 0, stride};
            const uint64_t words = wbits ? (stride * info.channels * (wbits / 8) + 3) / 4 + 4 : stride * info.channels;
            if (!win_grow_dev((void **)&w->d_pcm, &w->cap_pcm, words * sizeof(int32_t)) ||
                hipMemcpy(w->d_meta, meta, sizeof(meta), hipMemcpyHostToDevice) != hipSuccess)
                return 0;
            if (dvda_mlp_hip_decode(w->ctx, w->d_pcm, w->d_meta + 2, w->d_meta + 3, NULL) != DVDA_HIP_OK ||
                dvda_mlp_hip_stream_info(w->ctx, &info, 1, NULL) != DVDA_HIP_OK)
                return 0;
            if (!(info.status & DVDA_ST_OVERFLOW))
                break;
            if (attempt)
                return 0;
            stride = (info.pcm_frames + 3) & ~(uint64_t)3;       /* access units longer than the standard length */
        }
        w->p_status |= info.status;
        if (info.status & ~(uint32_t)DVDA_ST_BENIGN)
            return 0;                            /* the reference assert()s on such a stream */
        if (!w->info.channels)
            w->info = info;
        /* ---- the history at the cut, for the next window */
        if (!final) {
            uint32_t last_seg = 0;
            uint64_t dummy = 0;
            if (win_last_cut(w, n_seg, 0, &last_seg, &dummy) < 0)
                return 0;
            if (dvda_mlp_hip_segment_fir(w->ctx, last_seg, w->fir, NULL) != DVDA_HIP_OK)
                return 0;
            w->have_fir = 1;
        }
        /* ---- PCM (or payload) to the host buffer */
        const size_t bytes = (size_t)info.pcm_frames * info.channels * (wbits ? wbits / 8 : 4);
        if (bytes > out->cap) {
            if (out->host) {
                (void)hipHostFree(out->host);
                w->host_now -= out->cap;
            }
            out->host = NULL;
            out->cap = bytes + bytes / 4 + 4096;
            if (hipHostMalloc((void **)&out->host, out->cap, hipHostMallocDefault) != hipSuccess) {
                out->cap = 0;
                return 0;
            }
            win_host_add(w, out->cap);
        }
        if (bytes && hipMemcpy(out->host, w->d_pcm, bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return 0;
        out->frames = info.pcm_frames;
        w->p_frames += info.pcm_frames;
        win_dev_sample(w);
        w->p_final = final;
        return 1;
    }
}

/* One window of a raw-PCM track into `out` (reference: src/dvd-audio.c:1017-1083 decode_pcm_audio packet by packet,
 * src/pcm.c:99-193): the next run of sectors -> device -> k_pcm_scan / k_pcm_unswizzle_t -> the pinned slot, planar int32
 * or the packed payload.  Whole packets are delivered until the track's PTS length is covered (open_pcm's rule); a
 * track that spills over its sector range reads on.  1 = ok (out->frames may be 0), 0 = failure. */
static int win_produce_pcm(struct mlp_windows *w, struct win_slot *out)
{
    out->frames = 0;
    out->stride = 0;
    const unsigned bits = w->pcm_bits, ch = w->pcm_channels;
    unsigned want = w->window;
    if (w->next >= w->aobs.total) {
        w->p_final = 1;
        return w->started;
    }
    if (w->next + want > w->aobs.total || w->next + want < w->next)
        want = w->aobs.total - w->next;
    if (want > w->cap_sec) {
        if (w->h_sec) {
            (void)hipHostFree(w->h_sec);
            w->host_now -= w->cap_sec * SECTOR;
        }
        (void)hipFree(w->d_sec);
        (void)hipFree(w->d_work);
        free(w->h_base);
        w->h_sec = w->d_sec = NULL;
        w->d_work = w->h_base = NULL;
        w->cap_sec = 0;
        const size_t cap = (size_t)want * SECTOR;
        if (hipHostMalloc((void **)&w->h_sec, cap, hipHostMallocDefault) != hipSuccess)
            return 0;
        win_host_add(w, cap);
        if (!dev_alloc((void **)&w->d_sec, cap) ||
            !dev_alloc((void **)&w->d_work, dvda_pcm_hip_workspace_words(want) * sizeof(uint32_t)) ||
            (w->h_base = malloc(((size_t)want + 1) * sizeof(uint32_t))) == NULL)
            return 0;
        w->cap_sec = want;
    }
    const unsigned got = aob_read(&w->aobs, w->next, want, w->h_sec);
    if (!got) {
        w->p_final = 1;
        return w->started;                       /* the files end: what was delivered stands */
    }
    /* a sector holds at most 2013 payload bytes: upper bound of the PCM frames */
    uint64_t stride = (uint64_t)got * (2013 / (ch * (bits / 8) * 2)) * 2;
    stride = (stride + 3) & ~(uint64_t)3;
    uint64_t total = 0;
    uint32_t bad = 0;
    if (!win_grow_dev((void **)&w->d_pcm, &w->cap_pcm, stride * ch * sizeof(int32_t)) ||
        hipMemcpy(w->d_sec, w->h_sec, (size_t)got * SECTOR, hipMemcpyHostToDevice) != hipSuccess ||
        dvda_pcm_hip_decode_sectors(w->d_sec, got, bits, ch, w->d_pcm, stride, w->d_work, NULL) != DVDA_HIP_OK ||
        dvda_pcm_hip_result(w->d_work, got, &total, &bad, NULL) != DVDA_HIP_OK ||
        hipMemcpy(w->h_base, w->d_work + got, ((size_t)got + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
        return 0;
    (void)bad;                                   /* (open_pcm: foreign sectors contribute nothing) */
    uint64_t deliver = total;
    int final = 0;
    if (w->pcm_want == 0) {
        deliver = w->h_base[1] < total ? w->h_base[1] : total;      /* the opening packet is decoded regardless */
        final = 1;
    } else {
        for (unsigned s = 0; s < got; s++) {
            if (w->pcm_done + w->h_base[s + 1] >= w->pcm_want) {
                deliver = w->h_base[s + 1];
                final = 1;
                break;
            }
        }
    }
    w->next += got;
    w->started = 1;
    if (!final && (got < want || w->next >= w->aobs.total))
        final = 1;                               /* the files end inside the track */
    if (deliver) {
        if (w->wav_bits) {
            const size_t bytes = (size_t)deliver * ch * (bits / 8);
            if (!win_grow_dev((void **)&w->d_stream, &w->cap_stream, bytes + 64) ||
                dvda_mlp_hip_pack_wav(w->d_pcm, stride, ch, deliver, bits, w->d_stream, NULL) != DVDA_HIP_OK)
                return 0;
            if (bytes > out->cap) {
                if (out->host) {
                    (void)hipHostFree(out->host);
                    w->host_now -= out->cap;
                }
                out->host = NULL;
                out->cap = bytes + bytes / 4 + 4096;
                if (hipHostMalloc((void **)&out->host, out->cap, hipHostMallocDefault) != hipSuccess) {
                    out->cap = 0;
                    return 0;
                }
                win_host_add(w, out->cap);
            }
            if (hipMemcpy(out->host, w->d_stream, bytes, hipMemcpyDeviceToHost) != hipSuccess)
                return 0;
        } else {
            const size_t bytes = (size_t)deliver * ch * sizeof(int32_t);
            if (bytes > out->cap) {
                if (out->host) {
                    (void)hipHostFree(out->host);
                    w->host_now -= out->cap;
                }
                out->host = NULL;
                out->cap = bytes + bytes / 4 + 4096;
                if (hipHostMalloc((void **)&out->host, out->cap, hipHostMallocDefault) != hipSuccess) {
                    out->cap = 0;
                    return 0;
                }
                win_host_add(w, out->cap);
            }
            /* the window's frames of every channel, the planes packed to `deliver` frames each */
            if (hipMemcpy2D(out->host, (size_t)deliver * sizeof(int32_t), w->d_pcm, (size_t)stride * sizeof(int32_t),
                            (size_t)deliver * sizeof(int32_t), ch, hipMemcpyDeviceToHost) != hipSuccess)
                return 0;
            out->stride = deliver;
        }
        out->frames = deliver;
    }
    w->pcm_done += deliver;
    w->p_frames += deliver;
    w->p_final = final;
    win_dev_sample(w);
    return 1;
}

/* publishes what the last win_produce() left (the caller holds w->mu, or no consumer exists yet) */
static void win_commit(struct mlp_windows *w)
{
    w->status |= w->p_status;
    w->frames_total += w->p_frames;
    w->p_frames = 0;
    if (w->p_final)
        w->finished = 1;
}

static void *win_thread(void *arg)
{
    struct mlp_windows *w = arg;
    if (hipSetDevice(w->device) != hipSuccess) {
        pthread_mutex_lock(&w->mu);
        w->failed = 1;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
        return NULL;
    }
    for (;;) {
        pthread_mutex_lock(&w->mu);
        while (!w->stop && w->count == WIN_SLOTS)
            pthread_cond_wait(&w->cv, &w->mu);
        if (w->stop || w->finished || w->failed) {
            pthread_mutex_unlock(&w->mu);
            break;
        }
        struct win_slot *out = &w->slot[w->head];
        pthread_mutex_unlock(&w->mu);
        const int ok = w->is_pcm ? win_produce_pcm(w, out) : win_produce(w, out);
        pthread_mutex_lock(&w->mu);
        win_commit(w);                          /* status, frames and "finished" together with the window itself */
        if (!ok)
            w->failed = 1;
        else if (out->frames) {
            w->head = (w->head + 1) % WIN_SLOTS;
            w->count++;
        }
        pthread_cond_broadcast(&w->cv);
        const int done = w->finished || w->failed;
        pthread_mutex_unlock(&w->mu);
        if (done)
            break;
    }
    return NULL;
}

/* the window the consumer reads from: waits for the producer; NULL at the end of the track (or on failure) */
static struct win_slot *win_current(struct mlp_windows *w)
{
    struct win_slot *s = NULL;
    pthread_mutex_lock(&w->mu);
    while (w->count == 0 && !w->finished && !w->failed)
        pthread_cond_wait(&w->cv, &w->mu);
    if (w->count)
        s = &w->slot[w->tail];
    pthread_mutex_unlock(&w->mu);
    return s;
}
static void win_release(struct mlp_windows *w)
{
    pthread_mutex_lock(&w->mu);
    w->tail = (w->tail + 1) % WIN_SLOTS;
    w->count--;
    w->served_in_slot = 0;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
}

static DVDA_Track_Reader *open_mlp_windowed(const DVDA_Track *k)
{
    struct mlp_windows *w = calloc(1, sizeof(*w));
    DVDA_Track_Reader *r = calloc(1, sizeof(*r));
    size_t tot = 0;
    if (!w || !r) {
        free(w);
        free(r);
        return NULL;
    }
    pthread_mutex_init(&w->mu, NULL);
    pthread_cond_init(&w->cv, NULL);
    r->win = w;
    r->codec = DVDA_MLP;
    w->device = g_device;
    w->first = w->next = k->s.first;
    w->last = k->s.last >= k->s.first ? k->s.last : k->s.first;
    w->window = window_sectors();
    {
        /* what the last windowed reader of this thread left (same device): its context and buffers serve this track too */
        struct win_cache *c = &t_win_cache;
        if (c->valid && c->device != w->device)
            win_cache_free(c);
        if (c->valid) {
            w->ctx = c->ctx;
            w->ctx_segs = c->ctx_segs;
            w->h_sec = c->h_sec;
            w->d_sec = c->d_sec;
            w->d_mlp = c->d_mlp;
            w->d_stream = c->d_stream;
            w->carry = c->carry;
            w->d_work = c->d_work;
            w->h_base = c->h_base;
            w->d_meta = c->d_meta;
            w->d_pcm = c->d_pcm;
            w->d_fir = c->d_fir;
            w->cap_sec = c->cap_sec;
            w->cap_stream = c->cap_stream;
            w->cap_pcm = c->cap_pcm;
            w->carry_cap = c->carry_cap;
            for (int i = 0; i < WIN_SLOTS; i++)
                w->slot[i] = c->slot[i];
            w->host_now = w->host_peak = c->host_bytes;
            w->dev_base = c->dev_bytes;
            w->dev_peak = c->dev_bytes;
            memset(c, 0, sizeof(*c));
        }
    }
    (void)hipMemGetInfo(&w->dev_free0, &tot);
    aob_open_all(&w->aobs, k->dir, k->titleset);
    if (w->aobs.n == 0)
        goto fail;
    /* the first window in this thread: the stream's parameters are known when the reader is handed out (and the output
       layout with them: win_produce decides it from the first window's index) */
    w->wav_bits = 0;
    w->wav_request = g_wav_output ? 1 : 0;
    {
        struct win_slot *out = &w->slot[0];
        for (;;) {
            const int ok = win_produce(w, out);
            win_commit(w);                      /* (no consumer yet: no lock needed) */
            if (!ok)
                goto fail;
            if (out->frames || w->finished)
                break;
        }
        if (!w->info.channels)
            goto fail;
        if (out->frames) {
            w->head = 1 % WIN_SLOTS;
            w->count = 1;
        }
    }
    r->bps_code[0] = w->info.group0_bps;
    r->bps_code[1] = w->info.group1_bps;
    r->rate_code[0] = w->info.group0_rate;
    r->rate_code[1] = w->info.group1_rate;
    r->assignment = w->info.assignment;
    r->channels = channels_of(w->info.assignment);
    r->status = w->status;
    r->interleaved = 1;
    if (r->channels == 0 || r->channels != w->info.channels)
        goto fail;
    if (!w->finished) {
        if (pthread_create(&w->th, NULL, win_thread, w) != 0)
            goto fail;
        w->th_started = 1;
    }
    return r;
fail:
    reader_free(r);
    return NULL;
}

/* ------------------------------------------------------------------ PCM track */
static DVDA_Track_Reader *open_pcm(struct aob_set *aobs, const DVDA_Track *k, const uint8_t *params)
{
    /* 9-byte parameter block: first_audio_frame 16u, 8p, bps 4u 4u, rate 4u 4u, 8p, assignment 8u,
     * 8p, crc 8u (src/pcm.c:80-96) */
    DVDA_Track_Reader *r = calloc(1, sizeof(*r));
    uint8_t *h_sec = NULL, *d_sec = NULL;
    uint32_t *d_work = NULL, *h_base = NULL;
    int32_t *d_pcm = NULL;
    if (!r)
        return NULL;
    r->codec = DVDA_PCM;
    r->bps_code[0] = params[3] >> 4;
    r->bps_code[1] = params[3] & 15;
    r->rate_code[0] = params[4] >> 4;
    r->rate_code[1] = params[4] & 15;
    r->assignment = params[6];
    r->channels = channels_of(r->assignment);
    const unsigned bits = bits_of(r->bps_code[0]), rate = rate_of(r->rate_code[0]);
    if (!r->channels || (bits != 16 && bits != 24) || !rate)
        goto fail;
    const uint64_t want_frames = (uint64_t)lround((double)k->s.pts_length * (double)rate / DVDA_HIP_PTS_PER_SECOND);
    const unsigned first = k->s.first;
    unsigned count = k->s.last >= first ? k->s.last - first + 1 : 1;
    for (;;) {
        if (first + count > aobs->total || first + count < first)
            count = aobs->total - first;
        free(h_sec);
        free(h_base);
        (void)hipFree(d_sec);
        (void)hipFree(d_work);
        (void)hipFree(d_pcm);
        h_sec = NULL;
        h_base = NULL;
        d_sec = NULL;
        d_work = NULL;
        d_pcm = NULL;
        h_sec = malloc((size_t)count * SECTOR);
        if (!h_sec)
            goto fail;
        const unsigned got = aob_read(aobs, first, count, h_sec);
        if (!got)
            goto fail;
        /* a sector holds at most 2013 payload bytes: upper bound of the PCM frames */
        uint64_t stride = (uint64_t)got * (2013 / (r->channels * (bits / 8) * 2)) * 2;
        stride = (stride + 3) & ~(uint64_t)3;
        if (!dev_alloc((void **)&d_sec, (size_t)got * SECTOR) ||
            !dev_alloc((void **)&d_work, dvda_pcm_hip_workspace_words(got) * sizeof(uint32_t)) ||
            !dev_alloc((void **)&d_pcm, stride * r->channels * sizeof(int32_t)))
            goto fail;
        if (hipMemcpy(d_sec, h_sec, (size_t)got * SECTOR, hipMemcpyHostToDevice) != hipSuccess)
            goto fail;
        if (dvda_pcm_hip_decode_sectors(d_sec, got, bits, r->channels, d_pcm, stride, d_work, NULL) != DVDA_HIP_OK)
            goto fail;
        uint64_t total = 0;
        uint32_t bad = 0;
        if (dvda_pcm_hip_result(d_work, got, &total, &bad, NULL) != DVDA_HIP_OK)
            goto fail;
        (void)bad;              /* see open_mlp: foreign sectors inside the window count as bad and contribute nothing */
        h_base = malloc(((size_t)got + 1) * sizeof(uint32_t));
        if (!h_base ||
            hipMemcpy(h_base, d_work + got, ((size_t)got + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess)
            goto fail;
        /* whole packets are delivered until the track's length is covered */
        uint64_t deliver = total;
        int covered = 0;
        for (unsigned s = 0; s < got; s++) {
            if (h_base[s + 1] >= want_frames) {
                deliver = h_base[s + 1];
                covered = 1;
                break;
            }
        }
        if (!covered && got == count && first + got < aobs->total) {
            count *= 2;                          /* the track spills over its sector range */
            continue;
        }
        if (want_frames == 0)
            deliver = h_base[1] < total ? h_base[1] : total;   /* the opening packet is decoded regardless */
        r->frames = deliver;
        r->stride = stride;
        r->d_pcm = d_pcm;
        d_pcm = NULL;
        break;
    }
    goto done;
fail:
    reader_free(r);
    r = NULL;
done:
    free(h_sec);
    free(h_base);
    (void)hipFree(d_sec);
    (void)hipFree(d_work);
    (void)hipFree(d_pcm);
    return r;
}

/* A raw-PCM track of more sectors than a window: read, un-swizzled and handed out window by window (round 6; the
 * reference streams a track of any length packet by packet, src/dvd-audio.c:752-795, 1017-1083).  The first window in
 * the opener's thread, the rest by the producer thread into the two pinned slots: what the reader holds is bounded by
 * the window, not by the track. */
static DVDA_Track_Reader *open_pcm_windowed(const DVDA_Track *k, const uint8_t *params)
{
    struct mlp_windows *w = calloc(1, sizeof(*w));
    DVDA_Track_Reader *r = calloc(1, sizeof(*r));
    size_t tot = 0;
    if (!w || !r) {
        free(w);
        free(r);
        return NULL;
    }
    pthread_mutex_init(&w->mu, NULL);
    pthread_cond_init(&w->cv, NULL);
    r->win = w;
    r->codec = DVDA_PCM;
    r->bps_code[0] = params[3] >> 4;
    r->bps_code[1] = params[3] & 15;
    r->rate_code[0] = params[4] >> 4;
    r->rate_code[1] = params[4] & 15;
    r->assignment = params[6];
    r->channels = channels_of(r->assignment);
    r->interleaved = 1;
    const unsigned bits = bits_of(r->bps_code[0]), rate = rate_of(r->rate_code[0]);
    w->device = g_device;
    w->is_pcm = 1;
    w->first = w->next = k->s.first;
    w->last = k->s.last >= k->s.first ? k->s.last : k->s.first;
    w->window = window_sectors();
    if (!r->channels || (bits != 16 && bits != 24) || !rate)
        goto fail;
    w->pcm_bits = bits;
    w->pcm_channels = r->channels;
    w->pcm_want = (uint64_t)lround((double)k->s.pts_length * (double)rate / DVDA_HIP_PTS_PER_SECOND);
    w->wav_request = g_wav_output ? 1 : 0;
    w->wav_bits = w->wav_request ? (int)bits : 0;
    w->wav_decided = 1;
    {
        /* the buffers the thread's last windowed reader left serve this one too (same device) */
        struct win_cache *c = &t_win_cache;
        if (c->valid && c->device != w->device)
            win_cache_free(c);
        if (c->valid) {
            w->ctx = c->ctx;
            w->ctx_segs = c->ctx_segs;
            w->h_sec = c->h_sec;
            w->d_sec = c->d_sec;
            w->d_mlp = c->d_mlp;
            w->d_stream = c->d_stream;
            w->carry = c->carry;
            w->d_work = c->d_work;
            w->h_base = c->h_base;
            w->d_meta = c->d_meta;
            w->d_pcm = c->d_pcm;
            w->d_fir = c->d_fir;
            w->cap_sec = c->cap_sec;
            w->cap_stream = c->cap_stream;
            w->cap_pcm = c->cap_pcm;
            w->carry_cap = c->carry_cap;
            for (int i = 0; i < WIN_SLOTS; i++)
                w->slot[i] = c->slot[i];
            w->host_now = w->host_peak = c->host_bytes;
            w->dev_base = c->dev_bytes;
            w->dev_peak = c->dev_bytes;
            memset(c, 0, sizeof(*c));
        }
    }
    (void)hipMemGetInfo(&w->dev_free0, &tot);
    aob_open_all(&w->aobs, k->dir, k->titleset);
    if (w->aobs.n == 0)
        goto fail;
    {
        struct win_slot *out = &w->slot[0];
        for (;;) {
            const int ok = win_produce_pcm(w, out);
            win_commit(w);                      /* (no consumer yet: no lock needed) */
            if (!ok)
                goto fail;
            if (out->frames || w->finished)
                break;
        }
        if (out->frames) {
            w->head = 1 % WIN_SLOTS;
            w->count = 1;
        }
    }
    if (!w->finished) {
        if (pthread_create(&w->th, NULL, win_thread, w) != 0)
            goto fail;
        w->th_started = 1;
    }
    return r;
fail:
    reader_free(r);
    return NULL;
}

/* ------------------------------------------------------------------ track reader */
DVDA_Track_Reader *dvda_hip_open_track_reader_on(const DVDA_Track *k, int device, int wav_output)
{
    /* the options of THIS reader: the calling thread's defaults are put back whatever happens */
    const int keep_device = g_device, keep_wav = g_wav_output;
    g_device = device;
    g_wav_output = wav_output != 0;
    DVDA_Track_Reader *r = dvda_open_track_reader(k);
    g_device = keep_device;
    g_wav_output = keep_wav;
    return r;
}

int dvda_hip_reader_wav_only(const DVDA_Track_Reader *r) { return r && (r->d_wav != NULL || (r->win && r->win->wav_bits)); }

DVDA_Track_Reader *dvda_open_track_reader(const DVDA_Track *k)
{
    struct aob_set aobs;
    DVDA_Track_Reader *r = NULL;
    uint8_t sec[SECTOR];
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= g_device) {
        fprintf(stderr, "libdvd_audio_hip: no HIP device %d (the decode path is GPU-only)\n", g_device);
        return NULL;
    }
    if (hipSetDevice(g_device) != hipSuccess)
        return NULL;
    aob_open_all(&aobs, k->dir, k->titleset);
    if (aobs.n == 0 || k->s.first >= aobs.total)
        goto out;
    /* the first audio packet at or after the track's first sector names the codec */
    for (unsigned s = k->s.first; s < aobs.total; s++) {
        unsigned codec = 0, pad2 = 0, body_len = 0;
        const uint8_t *body = NULL;
        if (aob_read(&aobs, s, 1, sec) != 1)
            break;
        const int rc = first_audio_packet(sec, &codec, &pad2, &body, &body_len);
        if (rc < 0)
            break;
        if (rc == 0)
            continue;
        if (codec == CODEC_MLP) {
            /* a long track is read and decoded window by window (bounded memory); a short one as one batch */
            const unsigned in_track = k->s.last >= k->s.first ? k->s.last - k->s.first + 1 : 1;
            r = in_track > window_sectors() ? open_mlp_windowed(k) : open_mlp(&aobs, k);
        } else if (codec == CODEC_PCM && body_len >= 9 && pad2 >= 9) {
            const unsigned in_track = k->s.last >= k->s.first ? k->s.last - k->s.first + 1 : 1;
            r = in_track > window_sectors() ? open_pcm_windowed(k, body) : open_pcm(&aobs, k, body);
        }
        break;
    }
out:
    aob_close_all(&aobs);
    return r;
}

void dvda_close_track_reader(DVDA_Track_Reader *r) { reader_free(r); }

dvda_codec_t dvda_codec(const DVDA_Track_Reader *r) { return r->codec; }
unsigned dvda_bits_per_sample(const DVDA_Track_Reader *r) { return bits_of(r->bps_code[0]); }
unsigned dvda_sample_rate(const DVDA_Track_Reader *r) { return rate_of(r->rate_code[0]); }
unsigned dvda_channel_count(const DVDA_Track_Reader *r) { return channels_of(r->assignment); }
unsigned dvda_hip_reader_status(const DVDA_Track_Reader *r)
{
    if (!r->win)
        return r->status;
    pthread_mutex_lock(&r->win->mu);
    const unsigned st = r->win->status;
    pthread_mutex_unlock(&r->win->mu);
    return st;
}
/* (a track read in windows knows its length when its last window has been decoded: until then, the frames so far) */
unsigned long long dvda_hip_reader_total_frames(const DVDA_Track_Reader *r)
{
    if (!r->win)
        return r->frames;
    pthread_mutex_lock(&r->win->mu);
    const unsigned long long n = r->win->frames_total;
    pthread_mutex_unlock(&r->win->mu);
    return n;
}
int dvda_hip_reader_windowed(const DVDA_Track_Reader *r) { return r && r->win != NULL; }
int dvda_hip_reader_failed(const DVDA_Track_Reader *r)
{
    int f = 0;
    if (r && r->win) {
        pthread_mutex_lock(&r->win->mu);
        f = r->win->failed;
        pthread_mutex_unlock(&r->win->mu);
    }
    return f;
}
int dvda_hip_reader_memory(const DVDA_Track_Reader *r, unsigned long long *host_peak, unsigned long long *device_peak)
{
    if (!r || !r->win)
        return 0;
    if (host_peak)
        *host_peak = r->win->host_peak;
    if (device_peak)
        *device_peak = r->win->dev_peak;
    return 1;
}

unsigned dvda_riff_wave_channel_mask(const DVDA_Track_Reader *r)
{
    /* speaker bits per channel assignment (src/dvd-audio.c:693-755) */
    enum { FL = 0x1, FR = 0x2, FC = 0x4, LF = 0x8, BL = 0x10, BR = 0x20, BC = 0x100 };
    static const unsigned mask[21] = {
        FC, FL | FR, FL | FR | BC, FL | FR | BL | BR, FL | FR | LF, FL | FR | LF | BC, FL | FR | LF | BL | BR,
        FL | FR | FC, FL | FR | FC | BC, FL | FR | FC | BL | BR, FL | FR | FC | LF, FL | FR | FC | LF | BC,
        FL | FR | FC | LF | BL | BR, FL | FR | FC | BC, FL | FR | FC | BL | BR, FL | FR | FC | LF,
        FL | FR | FC | LF | BC, FL | FR | FC | LF | BL | BR, FL | FR | BL | BR | LF, FL | FR | BL | BR | FC,
        FL | FR | BL | BR | FC | LF};
    return r->assignment < 21 ? mask[r->assignment] : 0;
}

unsigned dvda_read(DVDA_Track_Reader *r, unsigned pcm_frames, int buffer[])
{
    if (r->win) {
        /* a track read in windows: frames out of the window in hand, the next one when it is used up */
        struct mlp_windows *w = r->win;
        unsigned done = 0;
        if (w->wav_bits)
            return 0;           /* payload only (dvda_hip_reader_wav_next) */
        while (done < pcm_frames) {
            struct win_slot *s = win_current(w);
            if (!s)
                break;
            const uint64_t left = s->frames - w->served_in_slot;
            const unsigned n = left < pcm_frames - done ? (unsigned)left : pcm_frames - done;
            if (s->stride) {
                /* a raw-PCM window: planes of `stride` frames, interleaved here (src/dvd-audio.c:781-792) */
                for (unsigned c = 0; c < r->channels; c++) {
                    const int32_t *src = (const int32_t *)s->host + (size_t)c * s->stride + w->served_in_slot;
                    int *dst = buffer + (size_t)done * r->channels + c;
                    for (unsigned i = 0; i < n; i++)
                        dst[(size_t)i * r->channels] = src[i];
                }
            } else
            memcpy(buffer + (size_t)done * r->channels,
                   (const int32_t *)s->host + (size_t)w->served_in_slot * r->channels, (size_t)n * r->channels * sizeof(int32_t));
            w->served_in_slot += n;
            done += n;
            if (w->served_in_slot == s->frames)
                win_release(w);
        }
        r->served += done;
        return done;
    }
    if (r->d_wav)
        return 0;               /* opened under dvda_hip_set_wav_output(1): the track exists as WAV payload only */
    if (!r->pcm) {
        /* PCM of the whole track, fetched once */
        const size_t bytes = r->stride * r->channels * sizeof(int32_t);
        r->pcm = malloc(bytes ? bytes : 1);
        if (!r->pcm || hipMemcpy(r->pcm, r->d_pcm, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
            free(r->pcm);
            r->pcm = NULL;
            return 0;
        }
    }
    const uint64_t left = r->frames - r->served;
    const unsigned n = left < pcm_frames ? (unsigned)left : pcm_frames;
    if (r->interleaved) {
        memcpy(buffer, r->pcm + (size_t)r->served * r->channels, (size_t)n * r->channels * sizeof(int32_t));
    } else {
        for (unsigned c = 0; c < r->channels; c++) {
            const int32_t *src = r->pcm + (size_t)c * r->stride + r->served;
            for (unsigned i = 0; i < n; i++)
                buffer[(size_t)i * r->channels + c] = src[i];
        }
    }
    r->served += n;
    return n;
}

/* The payload window by window: *payload = the next piece of the track's WAV data bytes (valid until the next call on
 * this reader), returns its size, 0 at the end of the track.  On a reader that is not windowed: the whole payload, once.
 * An int32 windowed reader (no wav output) is packed on the host, write_signed per value (src/bitstream.c:2846-2857). */
unsigned long long dvda_hip_reader_wav_next(DVDA_Track_Reader *r, const unsigned char **payload)
{
    *payload = NULL;
    if (!r->win)
        return r->served == 0 ? dvda_hip_reader_wav_payload(r, payload) : 0;
    struct mlp_windows *w = r->win;
    const unsigned bits = bits_of(r->bps_code[0]);
    if (bits != 16 && bits != 24)
        return 0;
    if (w->served_in_slot)                       /* the piece handed out by the call before: done with */
        win_release(w);
    struct win_slot *s = win_current(w);
    if (!s)
        return 0;
    const size_t nb = bits / 8;
    if (!w->wav_bits && s->stride) {
        /* a planar int32 window (raw PCM read without the payload option): packed into a buffer of its own */
        const size_t bytes = (size_t)s->frames * r->channels * nb;
        if (bytes > w->pack_cap) {
            free(w->pack_tmp);
            w->pack_cap = bytes + bytes / 4;
            w->pack_tmp = malloc(w->pack_cap);
            if (!w->pack_tmp) {
                w->pack_cap = 0;
                return 0;
            }
        }
        const uint32_t sign = 1u << (bits - 1);
        for (uint64_t i = 0; i < s->frames; i++)
            for (unsigned c = 0; c < r->channels; c++) {
                const int32_t v = ((const int32_t *)s->host)[(size_t)c * s->stride + i];
                const uint32_t u = ((uint32_t)v & (sign - 1)) | (v < 0 ? sign : 0u);
                for (size_t b = 0; b < nb; b++)
                    w->pack_tmp[(i * r->channels + c) * nb + b] = (uint8_t)(u >> (8 * b));
            }
        w->served_in_slot = s->frames;
        r->served += s->frames;
        *payload = w->pack_tmp;
        return (unsigned long long)bytes;
    }
    if (!w->wav_bits) {
        /* int32 frames -> payload, in place (the packed form is shorter) */
        const int32_t *src = (const int32_t *)s->host;
        uint8_t *dst = s->host;
        const uint64_t n = s->frames * r->channels;
        const uint32_t sign = 1u << (bits - 1);
        for (uint64_t i = 0; i < n; i++) {
            const int32_t v = src[i];
            const uint32_t u = ((uint32_t)v & (sign - 1)) | (v < 0 ? sign : 0u);
            for (size_t b = 0; b < nb; b++)
                dst[i * nb + b] = (uint8_t)(u >> (8 * b));
        }
    }
    w->served_in_slot = s->frames;               /* released by the next call */
    r->served += s->frames;
    *payload = s->host;
    return (unsigned long long)s->frames * r->channels * nb;
}

unsigned long long dvda_hip_reader_wav_payload(DVDA_Track_Reader *r, const unsigned char **payload)
{
    if (r->win) {
        /* the whole payload of a windowed reader, for callers of the one-piece interface: every window appended (this
           is the one call on such a reader whose memory grows with the track) */
        struct mlp_windows *w = r->win;
        size_t have = 0, cap = 0;
        const unsigned char *piece = NULL;
        unsigned long long n;
        *payload = NULL;
        while ((n = dvda_hip_reader_wav_next(r, &piece)) != 0) {
            if (have + n > cap) {
                cap = (have + n) * 2;
                uint8_t *g = realloc(w->whole, cap);
                if (!g)
                    return 0;
                w->whole = g;
            }
            memcpy(w->whole + have, piece, n);
            have += n;
        }
        *payload = w->whole;
        return have;
    }
    const unsigned bits = bits_of(r->bps_code[0]);
    const uint64_t left = r->frames - r->served;
    const uint64_t bytes = left * r->channels * (bits / 8);
    uint8_t *d_out = NULL;
    *payload = NULL;
    if ((bits != 16 && bits != 24) || left == 0)
        return 0;
    if (r->d_wav) {
        /* the decode wrote the payload: one copy to the host */
        double t_mark = now_ms();
        free(r->wav);
        r->wav = malloc(r->wav_bytes ? r->wav_bytes : 1);
        if (!r->wav || r->served != 0 ||
            hipMemcpy(r->wav, r->d_wav, r->wav_bytes, hipMemcpyDeviceToHost) != hipSuccess)
            return 0;
        T_MARK("payload to the host");
        r->served = r->frames;
        *payload = r->wav;
        return r->wav_bytes;
    }
    free(r->wav);
    r->wav = malloc(bytes);
    if (!r->wav || !dev_alloc((void **)&d_out, bytes))
        return 0;
    /* planar: planes start at r->served inside each channel (shifted base, same stride);
     * frame-major: the values already are in payload order = one "channel" of left * channels values */
    const int rc = r->interleaved
        ? dvda_mlp_hip_pack_wav(r->d_pcm + r->served * r->channels, 4, 1, left * r->channels, bits, d_out, NULL)
        : dvda_mlp_hip_pack_wav(r->d_pcm + r->served, r->stride, r->channels, left, bits, d_out, NULL);
    if (rc != DVDA_HIP_OK ||
        hipMemcpy(r->wav, d_out, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipFree(d_out);
        return 0;
    }
    (void)hipFree(d_out);
    r->served = r->frames;
    *payload = r->wav;
    return bytes;
}
