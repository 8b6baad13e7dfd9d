/* oracle/cpu_pool.c -- TEST / BENCH INFRASTRUCTURE ONLY (never linked into the product path).
 *
 * A plain pthread pool around the CPU decoder for bench.py's `cpu_baseline` legs: one decoder
 * instance per title, titles handed out by an atomic counter, no Python (no GIL) between two decodes.
 * Compiled twice by oracle/Makefile:
 *   -DPOOL_REF    into oracle/_ref/libdvda_ref.so : the workers call ref_mlp_decode(), i.e. the REAL
 *                 reference (src/mlp.c through its mlp.h entry points, see oracle/ref_driver.c)
 *   -DPOOL_ORACLE into oracle/libmlp_oracle.so    : the workers call this repo's restatement
 * Decoder instances share nothing (no globals but const tables), which is what makes
 * "N cores = N independent decoders over disjoint titles" the fair all-core figure.
 */
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdlib.h>
#include <time.h>

#if defined(POOL_REF)
long ref_mlp_decode(const uint8_t *data, size_t len, size_t chunk, unsigned g0_bps, unsigned g1_bps,
                    unsigned g0_rate, unsigned g1_rate, unsigned channel_assignment, unsigned nch,
                    int32_t *out, size_t cap);
#else
long mlp_oracle_decode(const uint8_t *data, size_t len, size_t chunk, unsigned nch, int32_t *out, size_t cap,
                       unsigned *status);
#endif

struct pool {
    const uint8_t *base;
    const uint64_t *offs, *sizes;
    uint32_t n;
    unsigned bps, rate, assignment, nch;
    int32_t *out;                 /* [n][nch][frames] */
    size_t frames;
    double deadline;              /* CLOCK_MONOTONIC seconds; keep going round until then */
    atomic_ulong next;
    atomic_ulong done;
    atomic_int failed;
};

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void *worker(void *arg)
{
    struct pool *p = (struct pool *)arg;
    int32_t *scratch = NULL;
    for (;;) {
        const unsigned long idx = atomic_fetch_add(&p->next, 1);
        /* every title at least once; after that only while the time budget lasts */
        if (idx >= p->n && now_s() > p->deadline)
            break;
        {
            const uint32_t i = (uint32_t)(idx % p->n);
            /* the first decode of a title is the one the caller compares with; later rounds (timing only) go to a
               buffer of the thread's own -- two threads never write one title's PCM at the same time */
            int32_t *dst = idx < p->n ? p->out + (size_t)i * p->nch * p->frames : scratch;
            long r;
            if (idx >= p->n && !scratch) {
                scratch = (int32_t *)malloc((size_t)p->nch * p->frames * sizeof(int32_t));
                if (!scratch) {
                    atomic_store(&p->failed, 1);
                    break;
                }
                dst = scratch;
            }
#if defined(POOL_REF)
            r = ref_mlp_decode(p->base + p->offs[i], (size_t)p->sizes[i], 0, p->bps, p->bps, p->rate, p->rate,
                               p->assignment, p->nch, dst, p->frames);
#else
            unsigned st = 0;
            r = mlp_oracle_decode(p->base + p->offs[i], (size_t)p->sizes[i], 0, p->nch, dst, p->frames, &st);
            if (st)
                r = -2;
#endif
            if (r != (long)p->frames)
                atomic_store(&p->failed, 1);
            /* (every finished decode counts, also one that was running at the deadline: *seconds is taken after
               the threads have joined, so its time is in the denominator too -- counted and timed together) */
            atomic_fetch_add(&p->done, 1);
        }
    }
    free(scratch);
    return NULL;
}

/* Decodes titles i = 0..n-1 (bytes base + offs[i], sizes[i] long; every title `frames` PCM frames of
 * nch channels) into out[i][channel][frame] on `threads` threads: each title at least once, then round
 * and round until budget_s seconds have passed.  Returns the number of title decodes finished, 0 if a
 * decode failed or came out with another length; *seconds = wall time of the whole run. */
unsigned long cpu_pool_decode(const uint8_t *base, const uint64_t *offs, const uint64_t *sizes, uint32_t n,
                              unsigned bps_code, unsigned rate_code, unsigned assignment, unsigned nch,
                              int32_t *out, size_t frames, unsigned threads, double budget_s, double *seconds)
{
    struct pool p;
    pthread_t *th;
    unsigned t, started = 0;
    const double t0 = now_s();
    if (!threads)
        threads = 1;
    p.base = base;
    p.offs = offs;
    p.sizes = sizes;
    p.n = n;
    p.bps = bps_code;
    p.rate = rate_code;
    p.assignment = assignment;
    p.nch = nch;
    p.out = out;
    p.frames = frames;
    p.deadline = t0 + budget_s;
    atomic_init(&p.next, 0);
    atomic_init(&p.done, 0);
    atomic_init(&p.failed, 0);
    th = (pthread_t *)malloc(sizeof(pthread_t) * threads);
    if (!th)
        return 0;
    for (t = 0; t < threads; t++) {
        if (pthread_create(&th[t], NULL, worker, &p) != 0)
            break;
        started++;
    }
    if (!started)
        worker(&p);
    for (t = 0; t < started; t++)
        pthread_join(th[t], NULL);
    free(th);
    if (seconds)
        *seconds = now_s() - t0;
    return atomic_load(&p.failed) ? 0 : atomic_load(&p.done);
}
