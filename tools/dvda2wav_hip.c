/* dvda2wav_hip -- extracts the tracks of a DVD-Audio disc to WAV files on the GPU.
 *
 * Same command line and the same output files (names, 68-byte WAVE_FORMAT_EXTENSIBLE header,
 * data bytes) as the reference's utility (reference utils/dvda2wav.c:57-186, 257-397), built on
 * libdvd_audio_hip.so only: IFO walk, sector demux, MLP decode / PCM un-swizzle and the
 * write_signed packing of the data chunk all run through include/dvd-audio-hip.h; this file does
 * argument parsing, the header and fwrite().
 *
 *   cc -O2 -o dvda2wav_hip tools/dvda2wav_hip.c -Iinclude -Llibdvd-audio_amd -ldvd_audio_hip \
 *      -ldvda_mlp_hip -Wl,-rpath,'$ORIGIN/../libdvd-audio_amd'
 */
#include <getopt.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "dvd-audio-hip.h"

static void put16(uint8_t *p, unsigned v) { p[0] = v & 0xFF; p[1] = (v >> 8) & 0xFF; }
static void put32(uint8_t *p, unsigned v) { put16(p, v & 0xFFFF); put16(p + 2, v >> 16); }

/* RIFF/WAVE_FORMAT_EXTENSIBLE header as the reference writes it (utils/dvda2wav.c:352-397):
 * note that its RIFF size field counts the 12 + 8 + 40 + 8 header bytes themselves */
static void wave_header(uint8_t h[68], unsigned rate, unsigned channels, unsigned mask, unsigned bits,
                        unsigned frames)
{
    static const uint8_t guid[16] = {1, 0, 0, 0, 0, 0, 16, 0, 128, 0, 0, 170, 0, 56, 155, 113};
    const unsigned bytes = bits / 8, data = bytes * channels * frames;
    memcpy(h, "RIFF", 4);
    put32(h + 4, 12 + 40 + 8 + data + (data % 2));
    memcpy(h + 8, "WAVEfmt ", 8);
    put32(h + 16, 40);
    put16(h + 20, 0xFFFE);
    put16(h + 22, channels);
    put32(h + 24, rate);
    put32(h + 28, rate * channels * bytes);
    put16(h + 32, channels * bytes);
    put16(h + 34, bits);
    put16(h + 36, 22);
    put16(h + 38, bits);
    put32(h + 40, mask);
    memcpy(h + 44, guid, 16);
    memcpy(h + 60, "data", 4);
    put32(h + 64, data);
}

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + ts.tv_nsec * 1e-9;
}

static int extract(DVDA_Title *title, unsigned track_num, const char *dir, int device, int fused_wav)
{
    const int timing = getenv("DVDA_TOOL_TIMING") != NULL;       /* diagnostic: where a track's wall-clock time goes */
    const double t_begin = now_s();
    double t_write = 0;
    DVDA_Track *track = dvda_open_track(title, track_num);
    if (!track) {
        fprintf(stderr, "*** Error: unable to open track %u\n", track_num);
        return 0;
    }
    /* this tool only ever writes WAV files: MLP tracks are decoded straight into the payload */
    DVDA_Track_Reader *r = dvda_hip_open_track_reader_on(track, device, fused_wav);
    if (!r) {
        fprintf(stderr, "*** Error: unable to open track %u for reading\n", track_num);
        dvda_close_track(track);
        return 0;
    }
    const double t_opened = now_s();
    char path[4096];
    const size_t n = strlen(dir);
    snprintf(path, sizeof(path), "%s%strack-%2.2u-%2.2u.wav", dir, n && dir[n - 1] == '/' ? "" : "/",
             dvda_title_number(title), dvda_track_number(track));
    dvda_close_track(track);
    FILE *f = fopen(path, "wb");
    if (!f) {
        fprintf(stderr, "*** Error: unable to open \"%s\" for writing\n", path);
        dvda_close_track_reader(r);
        return 0;
    }
    const unsigned channels = dvda_channel_count(r), bits = dvda_bits_per_sample(r), rate = dvda_sample_rate(r);
    printf("* Extracting %s track  %u channels  %u Hz  %u bps\n", dvda_codec(r) == DVDA_MLP ? "MLP" : "PCM",
           channels, rate, bits);
    /* the header first, as the reference does (utils/dvda2wav.c:316-343: a placeholder, the data chunk as it is read, the
       finished header written over it) -- a long track comes window by window and knows its length at its end; a short
       one hands out its whole payload as the one and only piece */
    uint8_t h[68];
    const unsigned mask = dvda_riff_wave_channel_mask(r);
    wave_header(h, rate, channels, mask, bits, 0);
    int ok = fwrite(h, 1, sizeof(h), f) == sizeof(h);
    unsigned long long bytes = 0, piece;
    const unsigned char *payload = NULL;
    while (ok && (piece = dvda_hip_reader_wav_next(r, &payload)) != 0) {
        const double tw = now_s();
        ok = fwrite(payload, 1, piece, f) == piece;
        t_write += now_s() - tw;
        bytes += piece;
    }
    const double t_pieces = now_s();
    const unsigned long long frames = dvda_hip_reader_total_frames(r);
    if (dvda_hip_reader_windowed(r)) {
        /* a track read in windows: what it held at most (tools/disc_bench.py reads this line) */
        unsigned long long hp = 0, dp = 0;
        dvda_hip_reader_memory(r, &hp, &dp);
        printf("* Read in windows: host peak %.1f MB, device peak %.1f MB, payload %.1f MB\n", hp / 1e6, dp / 1e6, bytes / 1e6);
    }
    if (ok) {
        wave_header(h, rate, channels, mask, bits, (unsigned)frames);
        ok = fseek(f, 0, SEEK_SET) == 0 && fwrite(h, 1, sizeof(h), f) == sizeof(h);
    }
    ok = fclose(f) == 0 && ok;
    /* (a window that could not be read or decoded ends the pieces early: that is a failed track, not a short one) */
    ok = ok && bytes == frames * channels * (bits / 8) && !dvda_hip_reader_failed(r);
    if (ok)
        printf("* Wrote: \"%s\"\n", path);
    else
        fprintf(stderr, "*** Error: writing \"%s\"\n", path);
    dvda_close_track_reader(r);
    if (timing)
        fprintf(stderr, "timing: track %u  open %.1f ms  pieces %.1f ms (of which fwrite %.1f)  finish %.1f ms  at %.1f ms\n", track_num,
                (t_opened - t_begin) * 1e3, (t_pieces - t_opened) * 1e3, t_write * 1e3, (now_s() - t_pieces) * 1e3,
                now_s() * 1e3);
    return ok;
}

/* ---- tracks fanned out over device entries (SURVEY 8(e) from the C host: tracks are independent -- the reference
 *      extracts them one after the other through one decoder, utils/dvda2wav.c:287-350, src/dvd-audio.c:597-657):
 *      one worker thread per entry of --devices, each takes the next track of the job list, opens it on ITS device,
 *      fetches the payload and writes the file.  A device named more than once is that many workers on it: one
 *      worker's file read and WAV write overlap another's decode. */
struct job {
    DVDA_Title *title;
    unsigned track;
};
struct pool {
    struct job *jobs;
    unsigned n_jobs;
    atomic_uint next, failed, done;     /* done: workers through with the job list */
    const char *dir;
    int fused_wav;
    int park;                           /* a worker that is through parks instead of ending (the fast way out) */
};
struct worker {
    struct pool *pool;
    int device;
    int threaded;                       /* runs in a thread of its own (not in main's) */
};

static void *work(void *arg)
{
    struct worker *w = arg;
    for (;;) {
        const unsigned i = atomic_fetch_add(&w->pool->next, 1);
        if (i >= w->pool->n_jobs)
            break;
        if (!extract(w->pool->jobs[i].title, w->pool->jobs[i].track, w->pool->dir, w->device, w->pool->fused_wav))
            atomic_fetch_add(&w->pool->failed, 1);
    }
    /* (what this thread's last windowed reader left for a next one would be freed when the thread ends -- 30 ms a worker.
       main() does not wait for that: it leaves with _exit as soon as every worker is through with the list, and a worker
       that has counted itself done PARKS here instead of ending, so that no thread is inside a HIP call -- its buffers'
       release -- when the process goes.  DVDA_TOOL_FULL_TEARDOWN=1: the orderly way, threads joined, everything freed) */
    atomic_fetch_add(&w->pool->done, 1);
    if (w->pool->park && w->threaded)
        for (;;)
            pause();
    return NULL;
}

static void usage(const char *prog)
{
    printf("*** Usage : %s -A [AUDIO_TS] [OPTIONS]\n"
           "Options:\n"
           "  -h, --help                show this help message and exit\n"
           "  -A PATH, --audio_ts=PATH  path to disc's AUDIO_TS directory\n"
           "  -S SET, --titleset=SET    title set number (default 1)\n"
           "  -T TITLE, --title=TITLE   title number to extract (default: all)\n"
           "  -t TRACK, --track=TRACK   track number to extract (default: all)\n"
           "  -d DIR, --dir=DIR         output directory (default: the working directory)\n"
           "  -g N, --gpu=N             HIP device (default 0)\n"
           "  -D LIST, --devices=LIST   comma-separated HIP devices: one worker thread per entry takes tracks in turn\n"
           "                            (default: up to four workers on the device of -g)\n"
           "                            (a device may be named more than once)\n", prog);
}

int main(int argc, char *argv[])
{
    static struct option longopts[] = {{"audio_ts", required_argument, 0, 'A'}, {"cdrom", required_argument, 0, 'c'},
                                       {"titleset", required_argument, 0, 'S'}, {"title", required_argument, 0, 'T'},
                                       {"track", required_argument, 0, 't'},    {"dir", required_argument, 0, 'd'},
                                       {"gpu", required_argument, 0, 'g'},      {"help", no_argument, 0, 'h'},
                                       {"devices", required_argument, 0, 'D'},
                                       {0, 0, 0, 0}};
    const char *audio_ts = NULL, *dir = ".", *cdrom = NULL;
    unsigned titleset_num = 1, title_num = 0, track_num = 0;
    int devices[64], n_devices = 0, one_device = 0;
    int c;
    if (getenv("DVDA_TOOL_TIMING"))
        fprintf(stderr, "timing: main at %.1f ms\n", now_s() * 1e3);
    while ((c = getopt_long(argc, argv, "A:c:S:T:t:d:g:D:h", longopts, NULL)) != -1) {
        switch (c) {
        case 'A': audio_ts = optarg; break;
        case 'c': cdrom = optarg; break;
        case 'S': titleset_num = (unsigned)strtoul(optarg, NULL, 10); break;
        case 'T': title_num = (unsigned)strtoul(optarg, NULL, 10); break;
        case 't': track_num = (unsigned)strtoul(optarg, NULL, 10); break;
        case 'd': dir = optarg; break;
        case 'g': one_device = atoi(optarg); break;
        case 'D':
            for (char *tok = strtok(optarg, ","); tok && n_devices < 64; tok = strtok(NULL, ","))
                devices[n_devices++] = atoi(tok);
            break;
        case 'h': usage(argv[0]); return 0;
        default: return 1;
        }
    }
    if (!audio_ts) {
        usage(argv[0]);
        return 0;
    }
    /* (DVDA_NO_FUSED_WAV=1 in the environment brings the int32 decode + packing pass back, for comparison) */
    const int fused_wav = getenv("DVDA_NO_FUSED_WAV") == NULL;
    DVDA *dvda = dvda_open(audio_ts, cdrom);
    DVDA_Titleset *ts = dvda ? dvda_open_titleset(dvda, titleset_num) : NULL;
    if (!ts) {
        fprintf(stderr, "*** Error: \"%s\" does not appear to be a valid AUDIO_TS path\n", audio_ts);
        if (dvda)
            dvda_close(dvda);
        return 1;
    }
    int rc = 0;
    const unsigned t_lo = title_num ? title_num : 1, t_hi = title_num ? title_num : dvda_title_count(ts);
    /* the job list: every (title, track) asked for, in the reference tool's order */
    DVDA_Title *titles[256];
    unsigned n_titles = 0, n_jobs = 0, cap_jobs = 0;
    struct job *jobs = NULL;
    for (unsigned t = t_lo; t <= t_hi && !rc && n_titles < 256; t++) {
        DVDA_Title *title = dvda_open_title(ts, t);
        if (!title) {
            fprintf(stderr, "*** Error: unable to open title %u\n", t);
            rc = 1;
            break;
        }
        titles[n_titles++] = title;
        const unsigned k_lo = track_num ? track_num : 1, k_hi = track_num ? track_num : dvda_track_count(title);
        for (unsigned k = k_lo; k <= k_hi; k++) {
            if (n_jobs == cap_jobs) {
                cap_jobs = cap_jobs ? 2 * cap_jobs : 64;
                jobs = realloc(jobs, cap_jobs * sizeof(*jobs));
                if (!jobs)
                    return 1;
            }
            jobs[n_jobs].title = title;
            jobs[n_jobs++].track = k;
        }
    }
    if (n_devices == 0) {
        /* no --devices: a pool of up to four workers on the one device (-g, default 0) -- one worker's file read and
           WAV write overlap the others' decodes; a single track needs no pool */
        const unsigned w = n_jobs >= 4 ? 4 : (n_jobs ? n_jobs : 1);
        for (unsigned i = 0; i < w; i++)
            devices[n_devices++] = one_device;
    }
    struct pool pool = {jobs, n_jobs, 0, 0, 0, dir, fused_wav, getenv("DVDA_TOOL_FULL_TEARDOWN") == NULL};
    struct worker workers[64];
    pthread_t th[64];
    int started[64];
    for (int i = 0; i < n_devices; i++) {
        workers[i].pool = &pool;
        workers[i].device = devices[i];
        workers[i].threaded = n_devices > 1;
        started[i] = n_devices > 1 && pthread_create(&th[i], NULL, work, &workers[i]) == 0;
        if (!started[i]) {
            workers[i].threaded = 0;
            work(&workers[i]);              /* one entry (or no thread to be had): here, in turn */
        }
    }
    {
        unsigned n_started = 0;
        for (int i = 0; i < n_devices; i++)
            n_started += started[i] ? 1u : 0u;
        if (getenv("DVDA_TOOL_FULL_TEARDOWN")) {
            for (int i = 0; i < n_devices; i++)
                if (started[i])
                    pthread_join(th[i], NULL);
        } else {
            /* every file is closed when a worker counts itself done; its buffers' release is not waited for */
            while (atomic_load(&pool.done) < n_started) {
                struct timespec ts = {0, 200000};
                nanosleep(&ts, NULL);
            }
        }
    }
    /* a track that could not be extracted (no such device, a read error, a decode the library reports) fails the run */
    if (atomic_load(&pool.failed)) {
        fprintf(stderr, "*** Error: %u of %u tracks could not be extracted\n", atomic_load(&pool.failed), n_jobs);
        rc = 1;
    }
    /* every file is written and closed.  What is left -- the titles' tables, and the HIP runtime's own teardown at exit
       (some 60-80 ms, measured: tools/probe/r05_disc_time.sh) -- is the operating system's to reclaim: a command-line
       tool that is done leaves */
    fflush(stdout);
    fflush(stderr);
    if (!getenv("DVDA_TOOL_FULL_TEARDOWN"))
        _exit(rc);
    for (unsigned i = 0; i < n_titles; i++)
        dvda_close_title(titles[i]);
    free(jobs);
    dvda_close_titleset(ts);
    dvda_close(dvda);
    return rc;
}
