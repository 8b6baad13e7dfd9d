// mlp_multi.cpp -- the title list of ONE host process dealt to several GPUs (include/dvda_mlp_hip.h, "multi" calls).
//
// the path shards with no exchange step -- titles are independent (the reference decodes one track
// at a time through one decoder, src/dvd-audio.c:597-657; nothing is shared between two of them) -- so a host that
// holds a list of streams gives every GPU its own sub-list and adds up a small summary.  This is the C restatement
// of libdvd-audio_amd/shard.py (what `bench.py --gpus N` does with one PROCESS per GPU and one RCCL all-reduce):
// here it is one host THREAD per device entry, each with its own decode context, device buffers and HIP stream.
// The summary -- the path's one reduction -- goes over RCCL when the device list names every device once (round 6:
// one communicator per device, ncclCommInitAll; two small all-reduces, a sum and a max, issued for all devices by the
// calling thread inside one group once every part is through; librccl is opened at run time, the library does not
// link it); a list that names a device twice (two contexts and two host threads on one GPU -- how the tests run it
// on a one-GPU box, and a way to overlap one batch's copies with another's decode), a box without librccl, or
// DVDA_MULTI_RCCL=0 reduce on the host, the same sums.  dvda_mlp_multi_summary.reduction says which it was.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <new>
#include <vector>

#include "../../include/dvda_mlp_hip.h"

// the four RCCL calls the summary needs (rccl.h: ncclCommInitAll, ncclCommDestroy, ncclAllReduce, ncclGroupStart / End)
struct RcclApi {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*AllReduce)(const void *send, void *recv, size_t count, int datatype, int op, void *comm, hipStream_t st) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
constexpr int NCCL_UINT64 = 5, NCCL_SUM = 0, NCCL_MAX = 2;      // ncclDataType_t / ncclRedOp_t
constexpr int RED_SUMS = 4, RED_MAXS = 3, RED_WORDS = 2 * (RED_SUMS + RED_MAXS);

struct dvda_mlp_hip_multi {
    std::vector<int> devices;
    std::vector<dvda_mlp_hip_ctx *> ctx;
    uint32_t max_streams, max_segments;
    std::vector<double> last_ms;            // per device entry: wall time of the last decode_multi
    std::vector<uint64_t> last_bytes;       // ... and the compressed bytes it was dealt
    // the summary over RCCL: a communicator, a stream and a few device words per device entry
    RcclApi rccl;
    bool use_rccl = false;
    std::vector<void *> comms;
    std::vector<hipStream_t> red_stream;
    std::vector<uint64_t *> d_red;          // [RED_WORDS]: sums to send | maxima to send | sums received | maxima received
};

static void rccl_close(dvda_mlp_hip_multi *m)
{
    for (size_t p = 0; p < m->comms.size(); p++) {
        (void)hipSetDevice(m->devices[p]);
        if (m->comms[p] && m->rccl.CommDestroy)
            (void)m->rccl.CommDestroy(m->comms[p]);
        if (p < m->red_stream.size() && m->red_stream[p])
            (void)hipStreamDestroy(m->red_stream[p]);
        if (p < m->d_red.size())
            (void)hipFree(m->d_red[p]);
    }
    m->comms.clear();
    m->red_stream.clear();
    m->d_red.clear();
    if (m->rccl.lib)
        dlclose(m->rccl.lib);
    m->rccl = RcclApi();
    m->use_rccl = false;
}

// every device named once, librccl there, the communicators made: the summary goes over RCCL; anything else: the host
static void rccl_open(dvda_mlp_hip_multi *m)
{
    const char *e = getenv("DVDA_MULTI_RCCL");
    if (e && atoi(e) == 0)
        return;
    std::vector<int> d = m->devices;
    std::sort(d.begin(), d.end());
    if (std::adjacent_find(d.begin(), d.end()) != d.end())
        return;                             // a device named twice: two ranks of one communicator cannot share a GPU
    RcclApi &r = m->rccl;
    r.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!r.lib)
        r.lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!r.lib)
        return;
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(dlsym(r.lib, "ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    const size_t n = m->devices.size();
    int keep = 0;
    (void)hipGetDevice(&keep);
    bool ok = r.CommInitAll && r.CommDestroy && r.AllReduce && r.GroupStart && r.GroupEnd;
    if (ok) {
        m->comms.assign(n, nullptr);
        ok = r.CommInitAll(m->comms.data(), (int)n, m->devices.data()) == 0;
    }
    m->red_stream.assign(n, nullptr);
    m->d_red.assign(n, nullptr);
    for (size_t p = 0; ok && p < n; p++)
        ok = hipSetDevice(m->devices[p]) == hipSuccess &&
             hipStreamCreateWithFlags(&m->red_stream[p], hipStreamNonBlocking) == hipSuccess &&
             hipMalloc((void **)&m->d_red[p], RED_WORDS * sizeof(uint64_t)) == hipSuccess;
    (void)hipSetDevice(keep);
    if (!ok) {
        rccl_close(m);
        return;
    }
    m->use_rccl = true;
}

// shard.shard_titles: greedy longest-processing-time on the compressed size -- streams by size descending (ties:
// lower index first), each to the part with the least bytes so far (ties: lower part) -- deterministic, so that any
// two callers (this one, the Python launcher) compute the same partition
extern "C" int dvda_mlp_hip_shard(const uint64_t *sizes, uint32_t n, uint32_t parts, uint32_t *part_of)
{
    if (!sizes || !part_of || parts == 0)
        return DVDA_HIP_EINVAL;
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++)
        order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return sizes[a] > sizes[b]; });
    std::vector<uint64_t> load(parts, 0);
    for (uint32_t k = 0; k < n; k++) {
        uint32_t best = 0;
        for (uint32_t p = 1; p < parts; p++)
            if (load[p] < load[best])
                best = p;
        part_of[order[k]] = best;
        load[best] += sizes[order[k]];
    }
    return DVDA_HIP_OK;
}

extern "C" int dvda_mlp_hip_create_multi(dvda_mlp_hip_multi **out, const int *devices, uint32_t n_devices,
                                         uint32_t max_streams, uint32_t max_segments)
{
    if (!out || !devices || n_devices == 0 || max_streams == 0 || max_segments == 0)
        return DVDA_HIP_EINVAL;
    dvda_mlp_hip_multi *m = new (std::nothrow) dvda_mlp_hip_multi();
    if (!m)
        return DVDA_HIP_ENOMEM;
    m->max_streams = max_streams;
    m->max_segments = max_segments;
    for (uint32_t d = 0; d < n_devices; d++) {
        dvda_mlp_hip_ctx *c = nullptr;
        const int rc = dvda_mlp_hip_create(&c, devices[d], max_streams, max_segments);
        if (rc != DVDA_HIP_OK) {
            for (dvda_mlp_hip_ctx *x : m->ctx)
                dvda_mlp_hip_destroy(x);
            delete m;
            return rc;                      // (no device, no decode: there is no CPU path behind this)
        }
        m->devices.push_back(devices[d]);
        m->ctx.push_back(c);
    }
    m->last_ms.assign(n_devices, 0.0);
    m->last_bytes.assign(n_devices, 0);
    rccl_open(m);
    *out = m;
    return DVDA_HIP_OK;
}

extern "C" void dvda_mlp_hip_destroy_multi(dvda_mlp_hip_multi *m)
{
    if (!m)
        return;
    rccl_close(m);
    for (dvda_mlp_hip_ctx *x : m->ctx)
        dvda_mlp_hip_destroy(x);
    delete m;
}

extern "C" uint32_t dvda_mlp_hip_multi_devices(const dvda_mlp_hip_multi *m) { return m ? (uint32_t)m->ctx.size() : 0u; }

extern "C" int dvda_mlp_hip_multi_device_time(const dvda_mlp_hip_multi *m, uint32_t entry, double *ms, uint64_t *bytes)
{
    if (!m || entry >= m->ctx.size())
        return DVDA_HIP_EINVAL;
    if (ms)
        *ms = m->last_ms[entry];
    if (bytes)
        *bytes = m->last_bytes[entry];
    return DVDA_HIP_OK;
}

namespace {

struct Job {
    dvda_mlp_hip_multi *m;
    uint32_t part;
    const uint8_t *const *streams;
    const uint64_t *lengths;
    const uint32_t *part_of;
    uint32_t n_streams;
    uint32_t layout;
    void *const *pcm;
    const uint64_t *capacity;
    dvda_mlp_stream_info *infos;
    int rc;
    uint64_t bytes;             // compressed bytes this part decoded
    double ms;                  // wall time of this part (its thread's clock)
    bool in_caller;             // no thread to be had: the part runs in the caller's thread, whose device is put back
};

inline double now_ms()
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

template <class T>
struct DevBuf {
    T *p = nullptr;
    ~DevBuf() { (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc((void **)&p, (n ? n : 1) * sizeof(T)) == hipSuccess; }
};

// bytes one PCM frame of a stream takes in the caller's buffer / one value in the device buffer, per layout
inline size_t value_bytes(uint32_t layout) { return layout == DVDA_PCM_WAV24 ? 3 : layout == DVDA_PCM_WAV16 ? 2 : 4; }

void *worker(void *arg)
{
    Job &j = *static_cast<Job *>(arg);
    dvda_mlp_hip_ctx *ctx = j.m->ctx[j.part];
    j.rc = DVDA_HIP_OK;
    j.bytes = 0;
    j.ms = 0;
    const double t_begin = now_ms();
    // (a part that runs in the caller's thread leaves the caller's current device as it found it)
    int caller_device = -1;
    if (j.in_caller && hipGetDevice(&caller_device) != hipSuccess)
        caller_device = -1;
    struct Restore {
        int dev;
        ~Restore() { if (dev >= 0) (void)hipSetDevice(dev); }
    } restore{caller_device};
    std::vector<uint32_t> mine;
    for (uint32_t i = 0; i < j.n_streams; i++)
        if (j.part_of[i] == j.part)
            mine.push_back(i);
    if (mine.empty())
        return nullptr;
    if (hipSetDevice(j.m->devices[j.part]) != hipSuccess) {
        j.rc = DVDA_HIP_ENODEV;
        return nullptr;
    }
    hipStream_t st = nullptr;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
        j.rc = DVDA_HIP_ENODEV;
        return nullptr;
    }
    const uint32_t n = (uint32_t)mine.size();
    // the part's streams back to back, each 16-byte aligned, 64 readable bytes behind the last one; its PCM the same
    // way, every stream at the capacity (PCM frames per channel) the caller gave it and 6 channels wide -- the widest
    // a DVD-Audio assignment is (src/dvd-audio.c:1459-1496); what the stream really has is known after the index
    std::vector<uint64_t> off(n), len(n), ooff(n), ostr(n);
    uint64_t total = 0, words = 0;
    const size_t vb = value_bytes(j.layout);
    for (uint32_t k = 0; k < n; k++) {
        const uint32_t i = mine[k];
        off[k] = total;
        len[k] = j.lengths[i];
        total += (j.lengths[i] + 15) & ~15ull;
        ooff[k] = words;
        ostr[k] = j.capacity[i];
        words += (j.capacity[i] * 6 * vb + 3) / 4 + 4;
        j.bytes += j.lengths[i];
    }
    DevBuf<uint8_t> d_bytes;
    DevBuf<uint64_t> d_meta;
    DevBuf<int32_t> d_pcm;
    std::vector<dvda_mlp_stream_info> info(n);
    uint8_t *h_stage = nullptr;
    // (the source of an asynchronous copy: it lives until the stream has been waited for, on every way out)
    std::vector<uint64_t> meta(4 * (size_t)n);
    do {
        if (!d_bytes.alloc(total + 64) || !d_meta.alloc(4 * (size_t)n) || !d_pcm.alloc(words) ||
            hipHostMalloc((void **)&h_stage, total + 64, hipHostMallocDefault) != hipSuccess) {
            j.rc = DVDA_HIP_ENOMEM;
            break;
        }
        memset(h_stage, 0, total + 64);
        for (uint32_t k = 0; k < n; k++)
            memcpy(h_stage + off[k], j.streams[mine[k]], len[k]);
        for (uint32_t k = 0; k < n; k++) {
            meta[k] = off[k];
            meta[n + k] = len[k];
            meta[2 * (size_t)n + k] = ooff[k];
            meta[3 * (size_t)n + k] = ostr[k];
        }
        if (hipMemcpyAsync(d_bytes.p, h_stage, total + 64, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_meta.p, meta.data(), meta.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st) != hipSuccess) {
            j.rc = DVDA_HIP_ENODEV;
            break;
        }
        if ((j.rc = dvda_mlp_hip_set_pcm_layout(ctx, j.layout)) != DVDA_HIP_OK ||
            (j.rc = dvda_mlp_hip_index(ctx, d_bytes.p, total, d_meta.p, d_meta.p + n, n, st)) != DVDA_HIP_OK ||
            (j.rc = dvda_mlp_hip_decode(ctx, d_pcm.p, d_meta.p + 2 * (size_t)n, d_meta.p + 3 * (size_t)n, st)) != DVDA_HIP_OK ||
            (j.rc = dvda_mlp_hip_stream_info(ctx, info.data(), n, st)) != DVDA_HIP_OK)
            break;
        // PCM back: planar [channel][capacity] int32, or frame-major / packed payload in one run per stream
        for (uint32_t k = 0; k < n && j.rc == DVDA_HIP_OK; k++) {
            const uint32_t i = mine[k];
            j.infos[i] = info[k];
            const uint64_t frames = info[k].pcm_frames < ostr[k] ? info[k].pcm_frames : ostr[k];
            const uint32_t ch = info[k].channels;
            if (!j.pcm[i] || !frames || !ch)
                continue;
            hipError_t e;
            if (j.layout == DVDA_PCM_PLANAR) {
                e = hipMemcpy2DAsync(j.pcm[i], ostr[k] * 4, d_pcm.p + ooff[k], ostr[k] * 4, frames * 4, ch,
                                     hipMemcpyDeviceToHost, st);
            } else {
                e = hipMemcpyAsync(j.pcm[i], d_pcm.p + ooff[k], frames * ch * vb, hipMemcpyDeviceToHost, st);
            }
            if (e != hipSuccess)
                j.rc = DVDA_HIP_ENODEV;
        }
    } while (0);
    // nothing is freed -- staging, meta, device buffers -- while a copy or a kernel may still use it: the stream is
    // waited for whichever way the block above was left
    if (hipStreamSynchronize(st) != hipSuccess && j.rc == DVDA_HIP_OK)
        j.rc = DVDA_HIP_ENODEV;
    if (h_stage)
        (void)hipHostFree(h_stage);
    (void)hipStreamDestroy(st);
    j.ms = now_ms() - t_begin;
    return nullptr;
}

} // namespace

extern "C" int dvda_mlp_hip_decode_multi(dvda_mlp_hip_multi *m, const uint8_t *const *streams, const uint64_t *lengths,
                                         uint32_t n_streams, uint32_t layout, void *const *pcm,
                                         const uint64_t *capacity_frames, dvda_mlp_stream_info *infos,
                                         dvda_mlp_multi_summary *summary)
{
    if (!m || !streams || !lengths || !pcm || !capacity_frames || !infos || n_streams == 0 || layout > DVDA_PCM_WAV16)
        return DVDA_HIP_EINVAL;
    if (n_streams > m->max_streams)
        return DVDA_HIP_ECAPACITY;
    const uint32_t parts = (uint32_t)m->ctx.size();
    std::vector<uint32_t> part_of(n_streams);
    int rc = dvda_mlp_hip_shard(lengths, n_streams, parts, part_of.data());
    if (rc != DVDA_HIP_OK)
        return rc;
    memset(infos, 0, sizeof(dvda_mlp_stream_info) * n_streams);
    std::vector<Job> jobs(parts);
    std::vector<pthread_t> th(parts);
    std::vector<char> started(parts, 0);
    for (uint32_t p = 0; p < parts; p++) {
        jobs[p] = Job{m, p, streams, lengths, part_of.data(), n_streams, layout, pcm, capacity_frames, infos, 0, 0, 0.0, false};
        started[p] = pthread_create(&th[p], nullptr, worker, &jobs[p]) == 0;
        if (!started[p]) {
            jobs[p].in_caller = true;       // (no thread to be had: the part is decoded here, in turn)
            worker(&jobs[p]);
        }
    }
    for (uint32_t p = 0; p < parts; p++)
        if (started[p])
            pthread_join(th[p], nullptr);
    for (uint32_t p = 0; p < parts; p++) {
        if (jobs[p].rc != DVDA_HIP_OK)
            rc = jobs[p].rc;
        m->last_ms[p] = jobs[p].ms;
        m->last_bytes[p] = jobs[p].bytes;
    }
    if (summary) {
        // the path's one reduction (bench.py does the same with one all-reduce over RCCL): sums, and the balance.
        // Every part's own contribution first: {PCM frames, samples, streams with errors, compressed bytes} to add,
        // {compressed bytes, wall time in us, ~wall time in us} to take the maximum of (the last one is the minimum's)
        memset(summary, 0, sizeof(*summary));
        summary->devices = parts;
        std::vector<uint64_t> loc((size_t)parts * (RED_SUMS + RED_MAXS), 0);
        for (uint32_t i = 0; i < n_streams; i++) {
            uint64_t *l = &loc[(size_t)part_of[i] * (RED_SUMS + RED_MAXS)];
            l[0] += infos[i].pcm_frames;
            l[1] += infos[i].pcm_frames * infos[i].channels;
            l[2] += (infos[i].status & ~(uint32_t)DVDA_ST_BENIGN) ? 1u : 0u;
        }
        for (uint32_t p = 0; p < parts; p++) {
            uint64_t *l = &loc[(size_t)p * (RED_SUMS + RED_MAXS)];
            l[3] = jobs[p].bytes;
            l[RED_SUMS + 0] = jobs[p].bytes;
            // (an entry that got no stream did nothing: it neither sets the slowest nor the fastest time)
            const uint64_t us = (uint64_t)(jobs[p].ms * 1e3);
            l[RED_SUMS + 1] = jobs[p].bytes ? us : 0;
            l[RED_SUMS + 2] = jobs[p].bytes ? ~us : 0;
        }
        uint64_t tot[RED_SUMS + RED_MAXS] = {0, 0, 0, 0, 0, 0, 0};
        bool reduced = false;
        if (m->use_rccl && rc == DVDA_HIP_OK) {
            // two small all-reduces over RCCL, issued for every device by this thread inside one group (every part is
            // through: no rank can be missing); any failure falls back to the host's sums below
            int keep = 0;
            (void)hipGetDevice(&keep);
            bool ok = true;
            for (uint32_t p = 0; ok && p < parts; p++)
                ok = hipSetDevice(m->devices[p]) == hipSuccess &&
                     hipMemcpyAsync(m->d_red[p], &loc[(size_t)p * (RED_SUMS + RED_MAXS)], (RED_SUMS + RED_MAXS) * sizeof(uint64_t),
                                    hipMemcpyHostToDevice, m->red_stream[p]) == hipSuccess;
            if (ok) {
                ok = m->rccl.GroupStart() == 0;
                for (uint32_t p = 0; ok && p < parts; p++) {
                    uint64_t *d = m->d_red[p];
                    ok = hipSetDevice(m->devices[p]) == hipSuccess &&
                         m->rccl.AllReduce(d, d + RED_SUMS + RED_MAXS, RED_SUMS, NCCL_UINT64, NCCL_SUM, m->comms[p], m->red_stream[p]) == 0 &&
                         m->rccl.AllReduce(d + RED_SUMS, d + 2 * RED_SUMS + RED_MAXS, RED_MAXS, NCCL_UINT64, NCCL_MAX, m->comms[p],
                                           m->red_stream[p]) == 0;
                }
                ok = (m->rccl.GroupEnd() == 0) && ok;
            }
            std::vector<uint64_t> back((size_t)parts * (RED_SUMS + RED_MAXS), 0);
            for (uint32_t p = 0; ok && p < parts; p++)
                ok = hipSetDevice(m->devices[p]) == hipSuccess &&
                     hipMemcpyAsync(&back[(size_t)p * (RED_SUMS + RED_MAXS)], m->d_red[p] + RED_SUMS + RED_MAXS,
                                    (RED_SUMS + RED_MAXS) * sizeof(uint64_t), hipMemcpyDeviceToHost, m->red_stream[p]) == hipSuccess &&
                     hipStreamSynchronize(m->red_stream[p]) == hipSuccess;
            // (every rank holds the same result; they are compared -- a summary is not worth a silent disagreement)
            for (uint32_t p = 1; ok && p < parts; p++)
                ok = memcmp(&back[0], &back[(size_t)p * (RED_SUMS + RED_MAXS)], (RED_SUMS + RED_MAXS) * sizeof(uint64_t)) == 0;
            (void)hipSetDevice(keep);
            if (ok) {
                memcpy(tot, back.data(), sizeof(tot));
                reduced = true;
            } else {
                (void)hipGetLastError();
            }
        }
        if (!reduced) {
            for (uint32_t p = 0; p < parts; p++) {
                const uint64_t *l = &loc[(size_t)p * (RED_SUMS + RED_MAXS)];
                for (int k = 0; k < RED_SUMS; k++)
                    tot[k] += l[k];
                for (int k = RED_SUMS; k < RED_SUMS + RED_MAXS; k++)
                    tot[k] = l[k] > tot[k] ? l[k] : tot[k];
            }
        }
        summary->pcm_frames = tot[0];
        summary->samples = tot[1];
        summary->streams_with_errors = (uint32_t)tot[2];
        summary->compressed_bytes = tot[3];
        summary->compressed_bytes_max_device = tot[RED_SUMS + 0];
        summary->device_ms_max = (double)tot[RED_SUMS + 1] * 1e-3;
        summary->device_ms_min = tot[RED_SUMS + 2] ? (double)(~tot[RED_SUMS + 2]) * 1e-3 : 0.0;
        summary->imbalance = tot[3] ? (double)tot[RED_SUMS + 0] * parts / (double)tot[3] : 1.0;
        summary->reduction = reduced ? 1u : 0u;
    }
    return rc;
}
