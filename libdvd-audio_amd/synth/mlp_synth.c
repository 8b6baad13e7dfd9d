/* mlp_synth.c -- synthetic MLP stream generator (see mlp_synth.h).
 *
 * Syntax written here, with the reference parser that reads it back:
 *   frame header "4p 12u 16p"                 reference src/mlp.c:392-394
 *   major sync (28 bytes)                     reference src/mlp.c:621-639
 *   substream info "1u 1u 1u 1p 12u" (+16p)   reference src/mlp.c:660-667, 465-467
 *   parity / CRC-8 trailer                    reference src/mlp.c:675-706, 1360-1399
 *   block = [params] rows last_block_bit      reference src/mlp.c:714-807
 *   restart header                            reference src/mlp.c:822-851
 *   decoding parameters                       reference src/mlp.c:866-990
 *   matrix / FIR / IIR parameters             reference src/mlp.c:1003-1023, 1033-1119
 *   residual rows                             reference src/mlp.c:1194-1238
 * The code books are the data of reference src/mlp_codebook{1,2,3}.json
 * re-expressed as (value -> code, length).
 */
#include "mlp_synth.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define MAXCH 8
#define MAXMAT 6

/* ------------------------------------------------------------------ rng */
typedef struct { uint32_t x; } rng_t;

static inline uint32_t rnd(rng_t *r)
{
    r->x = r->x * 1664525u + 1013904223u;
    return r->x >> 8; /* 24 useful bits */
}
static inline uint32_t rnd_below(rng_t *r, uint32_t n) { return n ? rnd(r) % n : 0; }
static inline int rnd_range(rng_t *r, int lo, int hi) /* inclusive */
{
    return lo + (int)rnd_below(r, (uint32_t)(hi - lo + 1));
}
static inline int rnd_chance(rng_t *r, unsigned percent)
{
    return rnd_below(r, 100) < percent;
}

/* ----------------------------------------------------------- bit writer */
typedef struct {
    uint8_t *buf;
    size_t cap;    /* bytes */
    size_t bits;   /* bits written */
    int overflow;
} bw_t;

static void bw_init(bw_t *w, uint8_t *buf, size_t cap)
{
    w->buf = buf;
    w->cap = cap;
    w->bits = 0;
    w->overflow = 0;
    memset(buf, 0, cap);
}

static void bw_put(bw_t *w, unsigned n, uint32_t v) /* MSB first, n <= 32 */
{
    while (n) {
        size_t byte = w->bits >> 3;
        unsigned room = 8 - (unsigned)(w->bits & 7);
        unsigned take = n < room ? n : room;
        uint32_t chunk;
        if (byte >= w->cap) {
            w->overflow = 1;
            return;
        }
        chunk = (n == 32 && take == 32) ? v : ((v >> (n - take)) & ((1u << take) - 1u));
        w->buf[byte] |= (uint8_t)(chunk << (room - take));
        w->bits += take;
        n -= take;
    }
}

static void bw_put_signed(bw_t *w, unsigned n, int v) /* two's complement, n bits */
{
    bw_put(w, n, (uint32_t)v & (n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u)));
}

static void bw_align(bw_t *w, unsigned bits) /* zero pad to a multiple of `bits` */
{
    while (w->bits % bits)
        bw_put(w, 1, 0);
}

/* ------------------------------------------------------------ codebooks */
/* value -> (code, length); reference src/mlp_codebook{1,2,3}.json */
typedef struct { uint16_t code; uint8_t len; } hcode_t;

static const hcode_t BOOK1[18] = {
    {0x001, 9}, {0x001, 8}, {0x001, 7}, {0x001, 6}, {0x001, 5}, {0x001, 4}, {0x001, 3},
    {0x004, 3}, {0x005, 3}, {0x006, 3}, {0x007, 3}, {0x003, 3},
    {0x005, 4}, {0x009, 5}, {0x011, 6}, {0x021, 7}, {0x041, 8}, {0x081, 9}};
static const hcode_t BOOK2[16] = {
    {0x001, 9}, {0x001, 8}, {0x001, 7}, {0x001, 6}, {0x001, 5}, {0x001, 4}, {0x001, 3},
    {0x002, 2}, {0x003, 2}, {0x003, 3},
    {0x005, 4}, {0x009, 5}, {0x011, 6}, {0x021, 7}, {0x041, 8}, {0x081, 9}};
static const hcode_t BOOK3[15] = {
    {0x001, 9}, {0x001, 8}, {0x001, 7}, {0x001, 6}, {0x001, 5}, {0x001, 4}, {0x001, 3},
    {0x001, 1}, {0x003, 3},
    {0x005, 4}, {0x009, 5}, {0x011, 6}, {0x021, 7}, {0x041, 8}, {0x081, 9}};

static const hcode_t *const BOOKS[4] = {NULL, BOOK1, BOOK2, BOOK3};
static const unsigned BOOK_MAX[4] = {0, 17, 15, 14};

/* ------------------------------------------------------------- CRC-8 */
static uint8_t CRC8T[256];
static pthread_once_t crc_once = PTHREAD_ONCE_INIT;
static void crc_init(void)
{
    unsigned i, k;
    for (i = 0; i < 256; i++) {
        unsigned c = i;
        for (k = 0; k < 8; k++)
            c = (c & 0x80) ? ((c << 1) ^ 0x63) : (c << 1);
        CRC8T[i] = (uint8_t)c;
    }
}

/* ------------------------------------------------------- stream tables */
unsigned mlp_synth_channels(uint32_t a)
{
    static const uint8_t n[21] = {1, 2, 3, 4, 3, 4, 5, 3, 4, 5, 4, 5, 6, 4, 5, 4, 5, 6, 5, 5, 6};
    return a < 21 ? n[a] : 0;
}

unsigned mlp_synth_rows_per_au(uint32_t rate_code)
{
    switch (rate_code) {
    case 0: case 8:  return 40;
    case 1: case 9:  return 80;
    case 2: case 10: return 160;
    default:         return 40;
    }
}

/* ------------------------------------------------- decoder-state model */
typedef struct {
    unsigned order;
    unsigned shift;
} filt_t;

typedef struct {
    filt_t fir, iir;
    int huff_off;
    unsigned codebook, lsbs;
} chan_t;

typedef struct {
    unsigned min_ch, max_ch, max_matrix_ch;
    unsigned flags[8];
    unsigned block_size;
    unsigned matrix_len;
    unsigned bypass[MAXMAT];
    unsigned qss[MAXCH];
    chan_t ch[MAXCH];
    int have_restart;
} ss_t;

typedef struct {
    const mlp_synth_cfg *cfg;
    rng_t rng;
    unsigned nch;
    unsigned rows_per_au;
    ss_t ss[2];
    unsigned leadin; /* current block must use order-0 filters */
} gen_t;

static const int RECIPE_FIR[8] = {8192, -4096, 2048, -1024, 512, -256, 128, -64};

static unsigned bits_for_signed(int maxabs_pos, int minneg)
{
    /* smallest n such that [-2^(n-1), 2^(n-1)-1] holds both extremes */
    unsigned n = 1;
    while (n < 16 && (maxabs_pos > (1 << (n - 1)) - 1 || minneg < -(1 << (n - 1))))
        n++;
    return n;
}

/* ---- restart header: reference src/mlp.c:822-851 */
static void put_restart_header(gen_t *g, bw_t *w, ss_t *s, unsigned au_index)
{
    const unsigned f = g->cfg->profile ? g->cfg->features : 0;
    unsigned c;
    unsigned noise_shift = (f & MLP_SF_NOISE) ? rnd_below(&g->rng, 8) : rnd_below(&g->rng, 3);
    bw_put(w, 13, 0x18F5);
    bw_put(w, 1, 0);                                   /* noise_type */
    bw_put(w, 16, (au_index * g->rows_per_au) & 0xFFFF); /* output_timestamp */
    bw_put(w, 4, s->min_ch);
    bw_put(w, 4, s->max_ch);
    bw_put(w, 4, s->max_matrix_ch);
    bw_put(w, 4, noise_shift);
    bw_put(w, 23, rnd(&g->rng) & 0x7FFFFF);            /* noise_gen_seed */
    bw_put(w, 19, rnd(&g->rng) & 0x7FFFF);             /* unknown */
    bw_put(w, 1, rnd(&g->rng) & 1);                    /* check_data_present (ignored) */
    bw_put(w, 8, rnd(&g->rng) & 0xFF);                 /* lossless_check (ignored) */
    bw_put(w, 16, rnd(&g->rng) & 0xFFFF);              /* unknown */
    for (c = 0; c <= s->max_matrix_ch; c++) {
        unsigned a = c;
        if (f & MLP_SF_FLAGS)
            a = rnd_below(&g->rng, s->max_matrix_ch + 1);
        bw_put(w, 6, a);                               /* channel_assignment */
    }
    bw_put(w, 8, rnd(&g->rng) & 0xFF);                 /* checksum (ignored) */
    s->have_restart = 1;
}

/* ---- matrices: reference src/mlp.c:1003-1023 */
static void put_matrices(gen_t *g, bw_t *w, ss_t *s, unsigned maxlen)
{
    const mlp_synth_cfg *cfg = g->cfg;
    const unsigned f = cfg->profile ? cfg->features : 0;
    const unsigned ncoef = s->max_matrix_ch + 3;
    unsigned m, c;
    unsigned len = cfg->n_matrices;
    if (f & MLP_SF_MATRIXRAND)
        len = rnd_below(&g->rng, MAXMAT + 1);
    if (len > MAXMAT)
        len = MAXMAT;
    /* In a non-first block of a frame the matrix count must not grow: the
       reference rematrixes the whole frame with the final count and would index
       bypassed-LSB arrays that earlier blocks never filled. */
    if (len > maxlen)
        len = maxlen;
    s->matrix_len = len;
    bw_put(w, 4, len);
    for (m = 0; m < len; m++) {
        unsigned out_ch = (f & MLP_SF_MATRIXRAND) ? rnd_below(&g->rng, s->max_matrix_ch + 1)
                                                  : (m % (s->max_matrix_ch + 1));
        unsigned frac = (f & MLP_SF_MATRIXRAND) ? (unsigned)rnd_range(&g->rng, 6, 14) : 14;
        unsigned bypass = (f & MLP_SF_MATRIXRAND) ? (rnd(&g->rng) & 1) : (m == 1);
        bw_put(w, 4, out_ch);
        bw_put(w, 4, frac);
        bw_put(w, 1, bypass);
        s->bypass[m] = bypass;
        for (c = 0; c < ncoef; c++) {
            int q14; /* desired coefficient in Q2.14 */
            int present = 1;
            if (c == out_ch) {
                q14 = 16384;
                if (f & MLP_SF_MATRIXRAND)
                    q14 = 16384 - (int)rnd_below(&g->rng, 4096);
            } else if (c <= s->max_matrix_ch) {
                int mag = 1000 + (int)rnd_below(&g->rng, 1501);
                q14 = (rnd(&g->rng) & 1) ? mag : -mag;
                if (f & MLP_SF_MATRIXRAND)
                    present = rnd_chance(&g->rng, 70);
            } else {
                q14 = (f & MLP_SF_NOISE) ? rnd_range(&g->rng, -600, 600) : 5;
                if (f & MLP_SF_MATRIXRAND)
                    present = rnd_chance(&g->rng, 60);
            }
            bw_put(w, 1, (uint32_t)present);
            if (present) {
                /* value is read_signed(frac + 2) << (14 - frac) */
                int v = q14 >> (14 - frac);
                int lim = (1 << (frac + 1)) - 1;
                if (v > lim) v = lim;
                if (v < -lim - 1) v = -lim - 1;
                bw_put_signed(w, frac + 2, v);
            }
        }
    }
}

/* ---- FIR: reference src/mlp.c:1033-1068.  Feedback taps on past OUTPUTS. */
static void put_fir(gen_t *g, bw_t *w, chan_t *ch, unsigned order_limit, int allow,
                    unsigned forced_shift)
{
    const mlp_synth_cfg *cfg = g->cfg;
    const unsigned f = cfg->profile ? cfg->features : 0;
    unsigned order, i;
    int coef[8];
    unsigned shift;

    if (!allow) {
        order = 0;
    } else if (f & MLP_SF_FIRRAND) {
        order = rnd_below(&g->rng, order_limit + 1);
    } else {
        order = cfg->fir_order <= order_limit ? cfg->fir_order : order_limit;
    }
    bw_put(w, 4, order);
    ch->fir.order = order;
    if (order == 0) {
        ch->fir.shift = 0;
        return;
    }
    if (f & MLP_SF_FIRRAND) {
        /* geometric decay, sum |c| <= 0.75 * 2^shift keeps the recursion bounded */
        int budget;
        shift = forced_shift ? forced_shift : (unsigned)rnd_range(&g->rng, 8, 15);
        budget = (3 << shift) >> 2;
        for (i = 0; i < order; i++) {
            int mag = (int)rnd_below(&g->rng, (uint32_t)(budget / 2 + 1));
            if (mag > 32767) mag = 32767;
            budget -= mag;
            coef[i] = (rnd(&g->rng) & 1) ? mag : -mag;
        }
    } else {
        /* the recipe filter is defined at shift 14; rescale it if an IIR that stays
           in force dictates another shift (both must agree) */
        shift = forced_shift ? forced_shift : 14;
        for (i = 0; i < order; i++)
            coef[i] = shift >= 14 ? RECIPE_FIR[i] * (1 << (shift - 14))
                                  : RECIPE_FIR[i] / (1 << (14 - shift));
    }
    {
        /* pick coeff_shift = common trailing zeros (<= 7), coeff_bits minimal */
        unsigned cshift = 7, cbits;
        int mx = 0, mn = 0;
        for (i = 0; i < order; i++) {
            unsigned tz = 0;
            int v = coef[i];
            if (v == 0) continue;
            while (tz < 7 && !((v >> tz) & 1)) tz++;
            if (tz < cshift) cshift = tz;
        }
        for (i = 0; i < order; i++) {
            int v = coef[i] >> cshift; /* exact: low bits are zero */
            if (v > mx) mx = v;
            if (v < mn) mn = v;
        }
        cbits = bits_for_signed(mx, mn);
        while (cbits + cshift > 16) { /* cannot happen for |c| <= 32767, be safe */
            cshift--;
        }
        bw_put(w, 4, shift);
        bw_put(w, 5, cbits);
        bw_put(w, 3, cshift);
        for (i = 0; i < order; i++)
            bw_put_signed(w, cbits, coef[i] >> cshift);
        bw_put(w, 1, 0); /* must be 0, reference src/mlp.c:1056 */
    }
    ch->fir.shift = shift;
}

/* ---- IIR: reference src/mlp.c:1075-1119.  Taps on past (output - prediction). */
static void put_iir(gen_t *g, bw_t *w, chan_t *ch, int allow)
{
    unsigned order = 0, i;
    unsigned shift;
    unsigned limit = 8 - ch->fir.order;
    if (allow && limit > 0)
        order = rnd_below(&g->rng, (limit > 4 ? 4 : limit) + 1);
    bw_put(w, 4, order);
    ch->iir.order = order;
    if (order == 0) {
        ch->iir.shift = 0;
        return;
    }
    /* both shifts > 0 must agree (reference src/mlp.c:1262-1265) */
    shift = ch->fir.shift ? ch->fir.shift : (unsigned)rnd_range(&g->rng, 8, 15);
    {
        int coef[8];
        int budget = 1 << shift; /* feed-forward gain <= 2 */
        unsigned cbits;
        int mx = 0, mn = 0;
        for (i = 0; i < order; i++) {
            int mag = (int)rnd_below(&g->rng, (uint32_t)(budget / 2 + 1));
            if (mag > 32767) mag = 32767;
            budget -= mag;
            coef[i] = (rnd(&g->rng) & 1) ? mag : -mag;
            if (coef[i] > mx) mx = coef[i];
            if (coef[i] < mn) mn = coef[i];
        }
        cbits = bits_for_signed(mx, mn);
        bw_put(w, 4, shift);
        bw_put(w, 5, cbits);
        bw_put(w, 3, 0);
        for (i = 0; i < order; i++)
            bw_put_signed(w, cbits, coef[i]);
        /* state must be present (reference indexes an emptied array otherwise,
           SURVEY.md A.4) */
        {
            unsigned sbits = (unsigned)rnd_range(&g->rng, 2, 12);
            unsigned sshift = rnd_below(&g->rng, 5);
            bw_put(w, 1, 1);
            bw_put(w, 4, sbits);
            bw_put(w, 4, sshift);
            for (i = 0; i < order; i++)
                bw_put_signed(w, sbits, rnd_range(&g->rng, -(1 << (sbits - 1)), (1 << (sbits - 1)) - 1));
        }
    }
    ch->iir.shift = shift;
}

/* ---- decoding parameters: reference src/mlp.c:866-990.
 * `restart`  : header_present
 * `first`    : first block of the frame (matrix-class changes are always legal there)
 * `recipe2`  : BASELINE recipe's second block of a restart AU (switch on the filters) */
static void put_decoding_params(gen_t *g, bw_t *w, ss_t *s, int restart, int first,
                                int recipe2, unsigned want_block_size)
{
    const mlp_synth_cfg *cfg = g->cfg;
    const unsigned f = cfg->profile ? cfg->features : 0;
    const int fuzz = cfg->profile != 0;
    const int matrix_ok = first || (f & MLP_SF_MIDMATRIX);
    unsigned c;

    /* flags */
    if (restart) {
        if ((f & MLP_SF_FLAGS) && rnd_chance(&g->rng, 50)) {
            bw_put(w, 1, 1);
            for (c = 0; c < 8; c++) {
                /* keep block-size (7) and per-channel (3) flags mostly on so the
                   stream can still carry its structure */
                unsigned v = rnd_chance(&g->rng, 80);
                if (c == 7) v = 1;
                s->flags[c] = v;
                bw_put(w, 1, v);
            }
        } else {
            bw_put(w, 1, 0);
            for (c = 0; c < 8; c++)
                s->flags[c] = 1;
        }
    } else if (s->flags[0]) {
        if ((f & MLP_SF_FLAGS) && rnd_chance(&g->rng, 20)) {
            bw_put(w, 1, 1);
            for (c = 0; c < 8; c++) {
                unsigned v = rnd_chance(&g->rng, 80);
                if (c == 7) v = 1;
                s->flags[c] = v;
                bw_put(w, 1, v);
            }
        } else {
            bw_put(w, 1, 0);
        }
    }

    /* block size */
    if (s->flags[7]) {
        if (want_block_size != s->block_size || (restart && want_block_size != 8)) {
            bw_put(w, 1, 1);
            bw_put(w, 9, want_block_size);
            s->block_size = want_block_size;
        } else if (restart) {
            /* equal to the default 8: either form is legal */
            bw_put(w, 1, 0);
            s->block_size = 8;
        } else {
            bw_put(w, 1, 0);
        }
    } else if (restart) {
        s->block_size = 8;
    }

    /* matrices */
    if (s->flags[6]) {
        int send = restart ? 1 : (fuzz && matrix_ok && (f & MLP_SF_PARAMBLOCKS) && rnd_chance(&g->rng, 30));
        if (restart && (f & MLP_SF_MATRIXRAND) && rnd_chance(&g->rng, 15))
            send = 0;
        bw_put(w, 1, (uint32_t)send);
        if (send)
            put_matrices(g, w, s, first ? MAXMAT : s->matrix_len);
        else if (restart)
            s->matrix_len = 0;
    } else if (restart) {
        s->matrix_len = 0;
    }

    /* output shifts (4s each, channels 0..max_matrix) */
    if (s->flags[5]) {
        int send = (f & MLP_SF_OUTSHIFT) && matrix_ok && (restart || ((f & MLP_SF_PARAMBLOCKS) && rnd_chance(&g->rng, 25)));
        bw_put(w, 1, (uint32_t)send);
        if (send)
            for (c = 0; c <= s->max_matrix_ch; c++)
                bw_put_signed(w, 4, (int)rnd_below(&g->rng, 3));
    }

    /* quant step sizes (4u each, channels 0..max_channel -- sic) */
    if (s->flags[4]) {
        int send = (f & MLP_SF_QSS) && matrix_ok && (restart || ((f & MLP_SF_PARAMBLOCKS) && rnd_chance(&g->rng, 25)));
        bw_put(w, 1, (uint32_t)send);
        if (send) {
            for (c = 0; c <= s->max_ch; c++) {
                unsigned q = rnd_below(&g->rng, 4);
                s->qss[c] = q;
                bw_put(w, 4, q);
            }
        } else if (restart) {
            for (c = 0; c < MAXCH; c++)
                s->qss[c] = 0;
        }
    } else if (restart) {
        for (c = 0; c < MAXCH; c++)
            s->qss[c] = 0;
    }

    /* per-channel parameters */
    for (c = s->min_ch; c <= s->max_ch; c++) {
        chan_t *ch = &s->ch[c];
        int filters_ok = !g->leadin;
        int send;
        if (restart)
            send = fuzz ? rnd_chance(&g->rng, 85) : 0;
        else if (recipe2)
            send = 1;
        else
            send = rnd_chance(&g->rng, (f & MLP_SF_DISC) ? 85 : 60);
        /* a channel whose current lsbs would fall below a freshly drawn qss must
           be re-sent */
        if (!send && !restart && ch->lsbs < s->qss[c])
            send = 1;
        if (restart && !send && 24 < s->qss[c])
            send = 1; /* unreachable: qss <= 3 */
        bw_put(w, 1, (uint32_t)send);
        if (!send) {
            if (restart) {
                ch->fir.order = ch->fir.shift = 0;
                ch->iir.order = ch->iir.shift = 0;
                ch->huff_off = 0;
                ch->codebook = 0;
                ch->lsbs = 24;
            }
            continue;
        }
        /* FIR */
        if (s->flags[3]) {
            int sf = recipe2 ? 1 : (fuzz ? rnd_chance(&g->rng, restart ? 60 : ((f & MLP_SF_DISC) ? 80 : 50)) : 0);
            if (!filters_ok && !restart && ch->fir.order)
                sf = 1; /* cannot happen: lead-in is always a restart block */
            bw_put(w, 1, (uint32_t)sf);
            if (sf) {
                /* keep FIR + IIR <= 8 and equal shifts with an IIR that stays in
                   force (on a restart the IIR is re-decided just below) */
                unsigned lim = restart ? 8 : 8 - ch->iir.order;
                unsigned fs = (!restart && ch->iir.order) ? ch->iir.shift : 0;
                put_fir(g, w, ch, lim, filters_ok, fs);
            } else if (restart) {
                ch->fir.order = ch->fir.shift = 0;
            }
        } else if (restart) {
            ch->fir.order = ch->fir.shift = 0;
        }
        /* IIR */
        if (s->flags[2]) {
            int want_iir = (f & MLP_SF_IIR) && filters_ok && rnd_chance(&g->rng, 40);
            int si = want_iir;
            /* shifts must agree when both are non-zero; orders must fit in 8 */
            if (!si && !restart && ch->iir.order &&
                (ch->fir.order + ch->iir.order > 8 ||
                 (ch->fir.shift && ch->iir.shift && ch->fir.shift != ch->iir.shift)))
                si = 1;
            bw_put(w, 1, (uint32_t)si);
            if (si)
                put_iir(g, w, ch, want_iir);
            else if (restart)
                ch->iir.order = ch->iir.shift = 0;
        } else {
            if (restart)
                ch->iir.order = ch->iir.shift = 0;
            /* IIR cannot be re-sent: FIR draw above was limited by ch->iir.order,
               but a shift clash is still possible -- resolved by the caller
               never enabling MLP_SF_IIR together with flags[2] == 0 updates */
        }
        /* huffman offset (15s) */
        if (s->flags[1]) {
            int so = (f & MLP_SF_HUFFOFF) && rnd_chance(&g->rng, 50);
            bw_put(w, 1, (uint32_t)so);
            if (so) {
                ch->huff_off = rnd_range(&g->rng, -300, 300);
                bw_put_signed(w, 15, ch->huff_off);
            } else if (restart) {
                ch->huff_off = 0;
            }
        } else if (restart) {
            ch->huff_off = 0;
        }
        /* codebook + huffman_lsbs */
        {
            unsigned cb, lsbs;
            if (g->leadin || (!recipe2 && !fuzz)) {
                cb = 0;
                lsbs = 24;
            } else if (f & MLP_SF_MIXBOOKS) {
                cb = rnd_below(&g->rng, 4);
                if (cb)
                    lsbs = s->qss[c] + rnd_below(&g->rng, 15 - 3);
                else
                    lsbs = s->qss[c] + (unsigned)rnd_range(&g->rng, 0, 20);
            } else if (recipe2 || fuzz) {
                cb = cfg->codebook;
                lsbs = cfg->huffman_lsbs;
                if (lsbs < s->qss[c]) lsbs = s->qss[c];
            } else {
                cb = 0;
                lsbs = 24;
            }
            if (lsbs > 24) lsbs = 24;
            ch->codebook = cb;
            ch->lsbs = lsbs;
            bw_put(w, 2, cb);
            bw_put(w, 5, lsbs);
        }
    }
}

/* ---- one block's residual rows: reference src/mlp.c:1194-1238 */
static void put_rows(gen_t *g, bw_t *w, const ss_t *s, unsigned rows)
{
    const int fuzz = g->cfg->profile != 0;
    unsigned i, m, c;
    for (i = 0; i < rows; i++) {
        for (m = 0; m < s->matrix_len; m++)
            if (s->bypass[m])
                bw_put(w, 1, rnd(&g->rng) & 1);
        for (c = s->min_ch; c <= s->max_ch; c++) {
            const chan_t *ch = &s->ch[c];
            const unsigned lsb_bits = ch->lsbs - s->qss[c];
            if (ch->codebook) {
                unsigned v;
                const hcode_t *hc;
                uint32_t r = rnd(&g->rng);
                if (fuzz && (r & 0xFF) < 6) {
                    /* rare extremes exercise the 8- and 9-bit codes */
                    v = (r & 0x100) ? 0 : BOOK_MAX[ch->codebook];
                    if ((r & 0x600) == 0x600) v = rnd_below(&g->rng, BOOK_MAX[ch->codebook] + 1);
                } else {
                    uint32_t r2 = rnd(&g->rng);
                    v = 7 + (r % 5) - (r2 % 3); /* 5..11, BASELINE.md recipe */
                }
                hc = &BOOKS[ch->codebook][v];
                bw_put(w, hc->len, hc->code);
                if (lsb_bits) {
                    uint32_t l = rnd(&g->rng);
                    if (lsb_bits > 24) return; /* not generated */
                    bw_put(w, lsb_bits, l & ((1u << lsb_bits) - 1u));
                }
            } else if (lsb_bits) {
                /* raw: residual = LSB - 2^(lsb_bits-1) (+offset), keep it within
                   +-2^15 for wide fields (BASELINE.md recipe), full range otherwise */
                uint32_t l;
                if (lsb_bits > 17) {
                    int v = (int)(rnd(&g->rng) & 0xFFFF) - 32768;
                    l = (uint32_t)((1 << (lsb_bits - 1)) + v);
                } else {
                    l = rnd(&g->rng) & ((1u << lsb_bits) - 1u);
                }
                bw_put(w, lsb_bits, l);
            }
        }
    }
}

/* splits `rows` into `n` blocks, each >= 8 */
static unsigned split_rows(gen_t *g, unsigned rows, unsigned n, unsigned *out, int random)
{
    unsigned i;
    if (n < 1) n = 1;
    while (n > 1 && rows / n < 8) n--;
    if (!random) {
        for (i = 0; i < n; i++)
            out[i] = rows / n;
        out[n - 1] += rows - (rows / n) * n;
        return n;
    }
    {
        unsigned left = rows;
        for (i = 0; i < n; i++) {
            unsigned remaining_blocks = n - i - 1;
            unsigned maxv = left - 8 * remaining_blocks;
            unsigned v = (remaining_blocks == 0) ? left : (unsigned)rnd_range(&g->rng, 8, (int)maxv);
            if (v > 511) v = 511;
            out[i] = v;
            left -= v;
        }
        if (left) {
            /* clipped at 511: give the rest to additional capacity of the last block */
            out[n - 1] += left > (511 - out[n - 1]) ? (511 - out[n - 1]) : left;
        }
        return n;
    }
}

/* builds one substream payload for one AU; returns byte length (even, includes
 * the parity/CRC trailer when `check`). */
static size_t build_substream(gen_t *g, ss_t *s, unsigned au, int restart_au, int last_au,
                              const unsigned *blk, unsigned nblk, int check,
                              uint8_t *buf, size_t cap)
{
    const mlp_synth_cfg *cfg = g->cfg;
    const unsigned f = cfg->profile ? cfg->features : 0;
    const int fuzz = cfg->profile != 0;
    const int chained = (f & MLP_SF_CHAINED) && au != 0;
    bw_t w;
    unsigned b;
    size_t nbytes;

    bw_init(&w, buf, cap);
    for (b = 0; b < nblk; b++) {
        int restart = (restart_au && b == 0) ||
                      (b > 0 && (f & MLP_SF_MIDRESTART) && rnd_chance(&g->rng, 10));
        int recipe2 = 0;
        int params;
        /* BASELINE recipe: the block after the raw lead-in switches the filters on.
           With a single block per AU that is the first block of the next AU. */
        if (!fuzz) {
            if (nblk > 1)
                recipe2 = restart_au && b == 1;
            else
                recipe2 = (!restart_au) && s->ch[s->min_ch].codebook == 0 && s->ch[s->min_ch].lsbs == 24;
        }
        g->leadin = restart && !chained && (b == 0);
        if (restart && b > 0)
            g->leadin = 1; /* mid-frame restarts also start with a raw block */
        params = restart || recipe2 || blk[b] != s->block_size ||
                 (fuzz && (f & MLP_SF_PARAMBLOCKS) && rnd_chance(&g->rng, 35));
        if (fuzz && (f & MLP_SF_DISC))
            params = 1;
        /* after a lead-in block the filters have to be (re)enabled at some point;
           in fuzz mode do it on the next block */
        if (fuzz && !restart && b > 0 && s->ch[s->min_ch].codebook == 0 && s->ch[s->min_ch].lsbs == 24 &&
            !(f & MLP_SF_MIXBOOKS))
            params = 1;
        if (fuzz && !restart && b == 0 && !restart_au && rnd_chance(&g->rng, 25))
            params = 1;
        bw_put(&w, 1, (uint32_t)params);
        if (params) {
            bw_put(&w, 1, (uint32_t)restart);
            if (restart)
                put_restart_header(g, &w, s, au);
            put_decoding_params(g, &w, s, restart, b == 0, recipe2, blk[b]);
        }
        put_rows(g, &w, s, s->block_size);
        bw_put(&w, 1, b + 1 == nblk); /* last block of the substream */
    }
    if ((f & MLP_SF_TERMINATOR) && last_au) {
        bw_align(&w, 8);
        bw_put(&w, 32, 0xD234D234u);
    }
    bw_align(&w, 16);
    if (w.overflow)
        return 0;
    nbytes = w.bits >> 3;
    if (check) {
        /* reference src/mlp.c:677-706, 1397-1398 */
        uint8_t parity = 0, crc = 0x3C, final_crc = 0;
        size_t i;
        if (nbytes + 2 > cap)
            return 0;
        for (i = 0; i < nbytes; i++) {
            parity ^= buf[i];
            final_crc = crc ^ buf[i];
            crc = CRC8T[final_crc];
        }
        buf[nbytes] = parity ^ 0xA9;
        buf[nbytes + 1] = final_crc;
        nbytes += 2;
    }
    return nbytes;
}

void mlp_synth_default(mlp_synth_cfg *cfg, uint32_t assignment, uint32_t rate_code,
                       uint32_t n_substreams, uint32_t n_aus)
{
    memset(cfg, 0, sizeof(*cfg));
    cfg->profile = 0;
    cfg->assignment = assignment;
    cfg->rate_code = rate_code;
    cfg->bps_code = 2;
    cfg->n_substreams = n_substreams;
    cfg->ss0_channels = 2;
    cfg->n_aus = n_aus;
    cfg->restart_interval = 8;
    cfg->blocks_per_au = 2;
    cfg->fir_order = 8;
    cfg->codebook = 1;
    cfg->huffman_lsbs = 12;
    cfg->n_matrices = 2;
}

size_t mlp_synth_bound(const mlp_synth_cfg *cfg)
{
    /* worst case per row and channel: 9 + 24 bits, plus 6 bypass bits per row;
       block headers: a few hundred bytes in fuzz mode, far less in the recipe */
    size_t rows = mlp_synth_rows_per_au(cfg->rate_code);
    size_t nch = mlp_synth_channels(cfg->assignment);
    size_t per_au;
    if (cfg->profile && (cfg->features & MLP_SF_VARROWS))
        rows *= 2;
    per_au = (cfg->profile ? 4096 : 320) + (rows * (nch * 33 + 6) + 7) / 8;
    if (per_au > 8190)
        per_au = 8190;
    return per_au * (size_t)cfg->n_aus + 64;
}

size_t mlp_synth_stream(const mlp_synth_cfg *cfg, uint64_t seed, uint8_t *out, size_t cap,
                        uint64_t *pcm_frames)
{
    gen_t g;
    unsigned au;
    size_t pos = 0;
    uint64_t frames = 0;
    uint8_t *ssbuf[2];
    const unsigned f = cfg->profile ? cfg->features : 0;
    const unsigned S = cfg->n_substreams == 2 ? 2 : 1;

    pthread_once(&crc_once, crc_init);
    memset(&g, 0, sizeof(g));
    g.cfg = cfg;
    g.rng.x = (uint32_t)(seed * 2654435761u) ^ (uint32_t)(seed >> 32) ^ 0x9E3779B9u;
    g.nch = mlp_synth_channels(cfg->assignment);
    g.rows_per_au = mlp_synth_rows_per_au(cfg->rate_code);
    if (g.nch == 0 || cfg->restart_interval == 0)
        return 0;
    if (S == 2 && (cfg->ss0_channels == 0 || cfg->ss0_channels >= g.nch))
        return 0;

    if (S == 1) {
        g.ss[0].min_ch = 0;
        g.ss[0].max_ch = g.nch - 1;
        g.ss[0].max_matrix_ch = g.nch - 1;
    } else {
        g.ss[0].min_ch = 0;
        g.ss[0].max_ch = cfg->ss0_channels - 1;
        g.ss[0].max_matrix_ch = cfg->ss0_channels - 1;
        g.ss[1].min_ch = cfg->ss0_channels;
        g.ss[1].max_ch = g.nch - 1;
        g.ss[1].max_matrix_ch = g.nch - 1;
    }
    {
        unsigned s, c;
        for (s = 0; s < 2; s++) {
            for (c = 0; c < 8; c++)
                g.ss[s].flags[c] = 1;
            g.ss[s].block_size = 8;
            for (c = 0; c < MAXCH; c++) {
                g.ss[s].ch[c].lsbs = 24;
            }
        }
    }

    ssbuf[0] = (uint8_t *)malloc(2 * 8192);
    if (!ssbuf[0])
        return 0;
    ssbuf[1] = ssbuf[0] + 8192;

    for (au = 0; au < cfg->n_aus; au++) {
        const int restart_au = (au % cfg->restart_interval) == 0;
        /* (MLP_SF_SYNCONLY: a major sync in front of an access unit whose substreams go on without a restart header) */
        const int sync_au = restart_au || ((f & MLP_SF_SYNCONLY) && rnd_chance(&g.rng, 30));
        const int last_au = au + 1 == cfg->n_aus;
        unsigned rows = g.rows_per_au;
        unsigned blk[2][8];
        unsigned nblk[2];
        size_t sslen[2] = {0, 0};
        unsigned extraword[2] = {0, 0};
        int check = !((f & MLP_SF_NOCHECK) && rnd_chance(&g.rng, 50));
        unsigned s;
        size_t total;
        bw_t hw;
        uint8_t hdr[4 + 28 + 8];

        if (f & MLP_SF_VARROWS)
            rows = (unsigned)rnd_range(&g.rng, 16, (int)(2 * g.rows_per_au));
        for (s = 0; s < S; s++) {
            unsigned want = cfg->blocks_per_au ? cfg->blocks_per_au : 1;
            if (f & MLP_SF_VARBLOCK)
                want = (unsigned)rnd_range(&g.rng, 1, 4);
            nblk[s] = split_rows(&g, rows, want, blk[s], (f & MLP_SF_VARBLOCK) != 0);
            sslen[s] = build_substream(&g, &g.ss[s], au, restart_au, last_au, blk[s], nblk[s],
                                       check, ssbuf[s], 8192);
            if (sslen[s] == 0)
                goto fail;
            if (f & MLP_SF_EXTRAWORD)
                extraword[s] = rnd_chance(&g.rng, 30);
        }
        total = 4 + (sync_au ? 28 : 0) + 2 * S + 2 * (extraword[0] + extraword[1]) + sslen[0] + sslen[1];
        if (total > 8190 || (total & 1))
            goto fail;
        if (pos + total > cap)
            goto fail;

        /* frame header + optional major sync + substream info */
        bw_init(&hw, hdr, sizeof(hdr));
        bw_put(&hw, 4, rnd(&g.rng) & 0xF);
        bw_put(&hw, 12, (uint32_t)(total / 2));
        bw_put(&hw, 16, (au * g.rows_per_au) & 0xFFFF);
        if (sync_au) {
            bw_put(&hw, 24, 0xF8726F);
            bw_put(&hw, 8, 0xBB);
            bw_put(&hw, 4, cfg->bps_code);
            bw_put(&hw, 4, cfg->bps_code);
            bw_put(&hw, 4, cfg->rate_code);
            bw_put(&hw, 4, cfg->rate_code);
            bw_put(&hw, 11, 0);
            bw_put(&hw, 5, cfg->assignment);
            bw_put(&hw, 32, 0);
            bw_put(&hw, 16, 0);
            bw_put(&hw, 1, 1);        /* is_VBR */
            bw_put(&hw, 15, 0x1234);  /* peak bitrate */
            bw_put(&hw, 4, S);
            bw_put(&hw, 32, 0);
            bw_put(&hw, 32, 0);
            bw_put(&hw, 28, 0);
        }
        {
            size_t end = 0;
            for (s = 0; s < S; s++) {
                end += sslen[s];
                bw_put(&hw, 1, extraword[s]);
                bw_put(&hw, 1, !restart_au);         /* nonrestart_substream (ignored) */
                /* (the reference reads substream 1's check bytes when SUBSTREAM 0's flag is set, src/mlp.c:545: what
                   substream 1's own flag says does not matter) */
                bw_put(&hw, 1, (uint32_t)((s == 1 && (f & MLP_SF_CHECKQUIRK)) ? !check : check));
                bw_put(&hw, 1, 0);
                bw_put(&hw, 12, (uint32_t)(end / 2));
                if (extraword[s])
                    bw_put(&hw, 16, rnd(&g.rng) & 0xFFFF);
            }
        }
        memcpy(out + pos, hdr, hw.bits >> 3);
        pos += hw.bits >> 3;
        for (s = 0; s < S; s++) {
            memcpy(out + pos, ssbuf[s], sslen[s]);
            pos += sslen[s];
        }
        {
            /* rows actually coded = sum of block sizes of substream 0 */
            unsigned b, r = 0;
            for (b = 0; b < nblk[0]; b++)
                r += blk[0][b];
            frames += r;
        }
    }
    free(ssbuf[0]);
    if (pcm_frames)
        *pcm_frames = frames;
    return pos;
fail:
    free(ssbuf[0]);
    return 0;
}

/* --------------------------------------------------------------- batch */
typedef struct {
    const mlp_synth_cfg *cfg;
    uint64_t seed0;
    uint32_t n, tid, nthreads;
    size_t slot;
    uint8_t **bufs;
    uint64_t *sizes;
    uint64_t *frames;
    int fail;
} job_t;

static void *batch_worker(void *p)
{
    job_t *j = (job_t *)p;
    uint8_t *tmp = (uint8_t *)malloc(j->slot);
    uint32_t i;
    if (!tmp) {
        j->fail = 1;
        return NULL;
    }
    for (i = j->tid; i < j->n; i += j->nthreads) {
        uint64_t fr = 0;
        size_t sz = mlp_synth_stream(j->cfg, j->seed0 + i, tmp, j->slot, &fr);
        j->sizes[i] = sz;
        j->frames[i] = fr;
        j->bufs[i] = sz ? (uint8_t *)malloc(sz) : NULL;
        if (!sz || !j->bufs[i]) {
            j->fail = 1;
            continue;
        }
        memcpy(j->bufs[i], tmp, sz);
    }
    free(tmp);
    return NULL;
}

size_t mlp_synth_batch(const mlp_synth_cfg *cfg, uint64_t seed0, uint32_t n, uint32_t threads,
                       uint8_t *out, size_t cap, uint64_t *offsets, uint64_t *sizes,
                       uint64_t *frames)
{
    const size_t slot = mlp_synth_bound(cfg);
    uint8_t **bufs;
    job_t jobs[64];
    pthread_t th[64];
    uint32_t t, i;
    size_t pos = 0;
    int fail = 0;

    if (threads < 1) threads = 1;
    if (threads > 64) threads = 64;
    if (threads > n) threads = n ? n : 1;
    bufs = (uint8_t **)calloc(n ? n : 1, sizeof(uint8_t *));
    if (!bufs)
        return 0;
    for (t = 0; t < threads; t++) {
        jobs[t].cfg = cfg;
        jobs[t].seed0 = seed0;
        jobs[t].n = n;
        jobs[t].tid = t;
        jobs[t].nthreads = threads;
        jobs[t].slot = slot;
        jobs[t].bufs = bufs;
        jobs[t].sizes = sizes;
        jobs[t].frames = frames;
        jobs[t].fail = 0;
        if (threads == 1)
            batch_worker(&jobs[t]);
        else if (pthread_create(&th[t], NULL, batch_worker, &jobs[t]))
            jobs[t].fail = 2;
    }
    for (t = 0; t < threads; t++) {
        if (threads > 1 && jobs[t].fail != 2)
            pthread_join(th[t], NULL);
        fail |= jobs[t].fail;
    }
    for (i = 0; i < n && !fail; i++) {
        size_t padded = ((size_t)sizes[i] + 15) & ~(size_t)15;
        if (pos + padded > cap) {
            fail = 1;
            break;
        }
        offsets[i] = pos;
        memcpy(out + pos, bufs[i], (size_t)sizes[i]);
        memset(out + pos + sizes[i], 0, padded - (size_t)sizes[i]);
        pos += padded;
    }
    for (i = 0; i < n; i++)
        free(bufs[i]);
    free(bufs);
    return fail ? 0 : pos;
}
