/* oracle/mlp_oracle.c -- TEST INFRASTRUCTURE ONLY (see mlp_oracle.h).
 *
 * A from-scratch CPU restatement of the algorithm of the reference decoder
 * tuffy/libdvd-audio src/mlp.c (each function cites the lines it follows).
 * It is the checker for the HIP path; it is never the thing shipped or measured
 * (except as bench.py's labelled cpu_baseline).  It reproduces the reference's
 * quirks on purpose: rematrix once per frame with the
 * parameters left by the frame's last block, FIR history never cleared,
 * substream 1 checked with substream 0's checkdata flag, restart-header
 * checksums ignored, the "final_crc" definition of the CRC-8 trailer.
 *
 * Where the reference would abort (assert) or read out of bounds, an error bit
 * is recorded and the frame is abandoned the way the reference's -DNDEBUG build
 * would abandon it.
 */
#include "mlp_oracle.h"

#include <stdlib.h>
#include <string.h>

#define MAX_SUBSTREAMS 2
#define MAX_MATRICES 6
#define MAX_CHANNELS 8
#define HIST 8

/* ----------------------------------------------------------------- tables */

/* (value, length) for a 9-bit peek, built from the code lists of reference
 * src/mlp_codebook{1,2,3}.json; value -1 = the two invalid all-zero-tail codes */
typedef struct { int8_t value; uint8_t len; } huff_entry;
static huff_entry HUFF[4][512];
static uint8_t CRC8[256];
static int tables_ready = 0;

static void add_code(huff_entry *t, unsigned code, unsigned len, int value)
{
    unsigned lo = code << (9 - len), n = 1u << (9 - len), i;
    for (i = 0; i < n; i++) {
        t[lo + i].value = (int8_t)value;
        t[lo + i].len = (uint8_t)len;
    }
}

static void build_tables(void)
{
    unsigned book, z, k, i;
    if (tables_ready)
        return;
    for (book = 1; book <= 3; book++) {
        huff_entry *t = HUFF[book];
        /* "1.."   : book1 1xx -> 7+xx, book2 1x -> 7+x, book3 1 -> 7 */
        unsigned sub = 3 - book, base = (book == 1) ? 11 : (book == 2) ? 9 : 8;
        for (i = 0; i < (1u << sub); i++)
            add_code(t, (1u << sub) | i, sub + 1, (int)(7 + i));
        /* "0^z 1", z = 2..8 -> 8 - z */
        for (z = 2; z <= 8; z++)
            add_code(t, 1, z + 1, (int)(8 - z));
        add_code(t, 0, 9, -1);
        /* "01 0^k 1", k = 0..6 -> base + k */
        for (k = 0; k <= 6; k++)
            add_code(t, (1u << (k + 1)) | 1u, k + 3, (int)(base + k));
        add_code(t, 0x80, 9, -1);
    }
    /* MSB-first CRC-8, polynomial 0x63 (reference table mlp.c:1363-1395, [1] = 0x63) */
    for (i = 0; i < 256; i++) {
        unsigned c = i;
        for (k = 0; k < 8; k++)
            c = (c & 0x80) ? ((c << 1) ^ 0x63) : (c << 1);
        CRC8[i] = (uint8_t)c;
    }
    tables_ready = 1;
}

/* RIFF-WAVE position of MLP channel c per channel assignment (mlp.c:416-438) */
static int wave_channel(unsigned assignment, unsigned c)
{
    static const uint8_t count[21] = {1, 2, 3, 4, 3, 4, 5, 3, 4, 5, 4, 5, 6, 4, 5, 4, 5, 6, 5, 5, 6};
    static const int8_t p12[6] = {0, 1, 3, 4, 2, -1};
    static const int8_t p14[6] = {0, 1, 4, 5, 2, 3};
    if (assignment > 20 || c >= 6)
        return -1;
    if (assignment == 0x12 || assignment == 0x13)
        return p12[c];
    if (assignment == 0x14)
        return p14[c];
    return c < count[assignment] ? (int)c : -1;
}

/* ------------------------------------------------------------- bit reader */
/* MSB-first reader over a byte range; the contract of reference
 * src/bitstream.c:1077-1111 (read), 1198-1206 (read_signed): read(0) == 0
 * without consuming; signed = sign bit then n-1 bits, two's complement. */
typedef struct {
    const uint8_t *p;
    size_t nbits;
    size_t pos;
    int eof;
} bits_t;

static void bits_init(bits_t *b, const uint8_t *p, size_t nbytes)
{
    b->p = p;
    b->nbits = nbytes * 8;
    b->pos = 0;
    b->eof = 0;
}

static uint32_t rd(bits_t *b, unsigned n)
{
    uint32_t v = 0;
    if (n == 0)
        return 0;
    if (b->pos + n > b->nbits) {
        b->eof = 1;
        b->pos = b->nbits;
        return 0;
    }
    while (n--) {
        v = (v << 1) | ((b->p[b->pos >> 3] >> (7 - (b->pos & 7))) & 1u);
        b->pos++;
    }
    return v;
}

static int rd_signed(bits_t *b, unsigned n)
{
    uint32_t v = rd(b, n);
    if (n == 0)
        return 0; /* ill-defined in the reference; never generated */
    if (n < 32 && (v & (1u << (n - 1))))
        return (int)(v | ~((1u << n) - 1u));
    return (int)v;
}

static void skip(bits_t *b, size_t n)
{
    if (b->pos + n > b->nbits) {
        b->eof = 1;
        b->pos = b->nbits;
    } else {
        b->pos += n;
    }
}

static size_t bytes_left(const bits_t *b) { return (b->nbits - b->pos) >> 3; }

/* ------------------------------------------------------------ decoder state */
typedef struct {
    unsigned order;
    unsigned shift;
    int coeff[HIST];
    int state[HIST]; /* state[0] = most recent */
    unsigned have;   /* valid history entries (<= HIST) */
} filter_t;

typedef struct {
    filter_t fir, iir;
    int huffman_offset;
    unsigned codebook;
    unsigned huffman_lsbs;
} chparams_t;

typedef struct {
    unsigned out_channel;
    unsigned lsb_bypass;
    int coeff[MAX_CHANNELS + 2];
    int *bypassed;     /* per-frame bypassed LSB bits */
    size_t bypassed_len, bypassed_cap;
} matrix_t;

typedef struct {
    /* substream info, mlp.c:48-53 */
    unsigned extraword, nonrestart, checkdata, end;
    /* restart header, mlp.c:55-63 */
    unsigned min_ch, max_ch, max_matrix_ch, noise_shift;
    uint32_t noise_seed;
    /* decoding parameters, mlp.c:88-101 */
    unsigned flags[8];
    unsigned block_size;
    unsigned matrix_len;
    matrix_t matrix[MAX_MATRICES];
    unsigned output_shift[MAX_CHANNELS];
    unsigned qss[MAX_CHANNELS];
    chparams_t ch[MAX_CHANNELS];
} substream_t;

typedef struct {
    int32_t *v;
    size_t len, cap;
} vec_t;

struct mlp_oracle {
    unsigned nch;
    unsigned status;
    /* byte queue (mlp.c:119, 349-351) */
    uint8_t *q;
    size_t qlen, qcap, qpos;
    /* latched major sync (mlp.c:121-122) */
    int sync_read;
    unsigned g0_bps, g1_bps, g0_rate, g1_rate, assignment, substream_count;
    substream_t ss[MAX_SUBSTREAMS];
    vec_t frame[MAX_CHANNELS]; /* "framelist", mlp.c:126 */
    vec_t out[MAX_CHANNELS];   /* the caller's "samples" */
    int32_t *residual;         /* scratch, block_size entries */
    size_t residual_cap;
};

static int vec_push_n(vec_t *v, const int32_t *src, size_t n)
{
    if (v->len + n > v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 1024;
        int32_t *nv;
        while (nc < v->len + n)
            nc *= 2;
        nv = (int32_t *)realloc(v->v, nc * sizeof(int32_t));
        if (!nv)
            return 0;
        v->v = nv;
        v->cap = nc;
    }
    if (src)
        memcpy(v->v + v->len, src, n * sizeof(int32_t));
    v->len += n;
    return 1;
}

/* mask(), mlp.c:246-253: clears the low q bits (arithmetic shifts there) */
static inline int32_t mask_q(int32_t x, unsigned q)
{
    return q ? (int32_t)((uint32_t)x & ~((1u << q) - 1u)) : x;
}

/* ------------------------------------------------------- parameter parsing */

/* mlp.c:809-854 */
static int parse_restart_header(bits_t *b, substream_t *s)
{
    unsigned sync, noise_type, c;
    sync = rd(b, 13);
    noise_type = rd(b, 1);
    (void)rd(b, 16); /* output_timestamp */
    s->min_ch = rd(b, 4);
    s->max_ch = rd(b, 4);
    s->max_matrix_ch = rd(b, 4);
    s->noise_shift = rd(b, 4);
    s->noise_seed = rd(b, 23);
    (void)rd(b, 19);
    (void)rd(b, 1);  /* check_data_present: ignored */
    (void)rd(b, 8);  /* lossless_check: ignored */
    (void)rd(b, 16);
    if (sync != 0x18F5 || noise_type != 0)
        return 0;
    if (s->max_ch < s->min_ch || s->max_matrix_ch < s->max_ch)
        return 0;
    for (c = 0; c <= s->max_matrix_ch; c++)
        if (rd(b, 6) > s->max_matrix_ch) /* channel_assignment: validated only */
            return 0;
    (void)rd(b, 8); /* checksum: ignored */
    return 1;
}

/* mlp.c:995-1027 */
static int parse_matrices(bits_t *b, substream_t *s, unsigned *envelope)
{
    unsigned m, c;
    s->matrix_len = rd(b, 4);
    if (s->matrix_len > MAX_MATRICES) {
        /* the reference does not check and overruns its array; the restatement reports it and
           keeps a state it can go on with (the decoder is fed further packets in the tests) */
        *envelope = 1;
        s->matrix_len = 0;
        return 0;
    }
    for (m = 0; m < s->matrix_len; m++) {
        matrix_t *mp = &s->matrix[m];
        unsigned frac;
        if ((mp->out_channel = rd(b, 4)) > s->max_matrix_ch)
            return 0;
        if ((frac = rd(b, 4)) > 14)
            return 0;
        mp->lsb_bypass = rd(b, 1);
        for (c = 0; c < s->max_matrix_ch + 3; c++) {
            int v = 0;
            if (rd(b, 1))
                v = (int)((uint32_t)rd_signed(b, frac + 2) << (14 - frac));
            if (c < MAX_CHANNELS + 2)
                mp->coeff[c] = v;
            else
                *envelope = 1; /* max_matrix_channel > 7 overruns coeff[8] there */
        }
    }
    return 1;
}

/* mlp.c:1029-1069 (FIR) and 1071-1120 (IIR) */
static int parse_filter(bits_t *b, filter_t *f, int is_iir, unsigned *envelope)
{
    unsigned order = rd(b, 4), i;
    if (order > 8)
        return 0;
    if (order == 0) {
        f->shift = 0;
        f->order = 0;
        if (is_iir)
            f->have = 0;
        return 1;
    }
    f->shift = rd(b, 4);
    {
        unsigned coeff_bits = rd(b, 5), coeff_shift;
        if (coeff_bits < 1 || coeff_bits > 16)
            return 0;
        coeff_shift = rd(b, 3);
        if (coeff_bits + coeff_shift > 16)
            return 0;
        f->order = order;
        for (i = 0; i < order; i++)
            f->coeff[i] = (int)((uint32_t)rd_signed(b, coeff_bits) << coeff_shift);
    }
    if (!is_iir) {
        if (rd(b, 1))
            return 0;
        return 1;
    }
    /* IIR: history is emptied, then optionally loaded; the first value read is
       the most recent one (mlp.c:1098-1108) */
    f->have = 0;
    if (rd(b, 1)) {
        unsigned state_bits = rd(b, 4), state_shift = rd(b, 4);
        if (state_bits == 0)
            *envelope = 1; /* read_signed(0) */
        for (i = 0; i < order; i++)
            f->state[i] = (int)((uint32_t)rd_signed(b, state_bits) << state_shift);
        f->have = order;
    }
    return 1;
}

/* mlp.c:856-993 */
static int parse_decoding_params(bits_t *b, substream_t *s, int header, unsigned *envelope)
{
    unsigned c;

    if (header) {
        if (rd(b, 1)) {
            for (c = 0; c < 8; c++)
                s->flags[c] = rd(b, 1);
        } else {
            for (c = 0; c < 8; c++)
                s->flags[c] = 1;
        }
    } else if (s->flags[0] && rd(b, 1)) {
        for (c = 0; c < 8; c++)
            s->flags[c] = rd(b, 1);
    }

    if (s->flags[7] && rd(b, 1)) {
        if ((s->block_size = rd(b, 9)) < 8)
            return 0;
    } else if (header) {
        s->block_size = 8;
    }

    if (s->flags[6] && rd(b, 1)) {
        if (!parse_matrices(b, s, envelope))
            return 0;
    } else if (header) {
        s->matrix_len = 0;
    }

    if (s->flags[5] && rd(b, 1)) {
        for (c = 0; c <= s->max_matrix_ch; c++) {
            int v = rd_signed(b, 4);
            if (c < MAX_CHANNELS)
                s->output_shift[c] = (unsigned)v;
            if (v < 0)
                *envelope = 1; /* becomes a huge unsigned shift there */
        }
    } else if (header) {
        for (c = 0; c < MAX_CHANNELS; c++)
            s->output_shift[c] = 0;
    }

    if (s->flags[4] && rd(b, 1)) {
        for (c = 0; c <= s->max_ch; c++) {
            unsigned v = rd(b, 4);
            if (c < MAX_CHANNELS)
                s->qss[c] = v;
        }
    } else if (header) {
        for (c = 0; c < MAX_CHANNELS; c++)
            s->qss[c] = 0;
    }

    for (c = s->min_ch; c <= s->max_ch; c++) {
        chparams_t *cp;
        if (c >= MAX_CHANNELS) {
            *envelope = 1;
            return 0;
        }
        cp = &s->ch[c];
        if (rd(b, 1)) {
            if (s->flags[3] && rd(b, 1)) {
                if (!parse_filter(b, &cp->fir, 0, envelope))
                    return 0;
            } else if (header) {
                cp->fir.shift = 0;
                cp->fir.order = 0;
            }
            if (s->flags[2] && rd(b, 1)) {
                if (!parse_filter(b, &cp->iir, 1, envelope))
                    return 0;
            } else if (header) {
                cp->iir.shift = 0;
                cp->iir.order = 0;
                cp->iir.have = 0;
            }
            if (s->flags[1] && rd(b, 1))
                cp->huffman_offset = rd_signed(b, 15);
            else if (header)
                cp->huffman_offset = 0;
            cp->codebook = rd(b, 2);
            {
                /* the reference stores the value and then fails; a later block would shift by it
                   (undefined there): the restatement fails without keeping it */
                const unsigned lsbs = rd(b, 5);
                if (lsbs > 24)
                    return 0;
                cp->huffman_lsbs = lsbs;
            }
        } else if (header) {
            cp->fir.shift = 0;
            cp->fir.order = 0;
            cp->iir.shift = 0;
            cp->iir.order = 0;
            cp->iir.have = 0;
            cp->huffman_offset = 0;
            cp->codebook = 0;
            cp->huffman_lsbs = 24;
        }
    }
    return 1;
}

/* ------------------------------------------------------------ block decode */

/* mlp.c:1243-1306.  `res` holds block_size residuals, results go to `dst`. */
static int filter_channel(mlp_oracle *d, const int32_t *res, unsigned n, filter_t *fir,
                          filter_t *iir, unsigned q, int32_t *dst)
{
    unsigned shift, i, j;
    if (fir->order + iir->order > 8)
        return 0;
    if (fir->shift > 0 && iir->shift > 0) {
        if (fir->shift != iir->shift)
            return 0;
        shift = fir->shift;
    } else if (fir->order > 0) {
        shift = fir->shift;
    } else {
        shift = iir->shift;
    }
    /* the reference indexes state[len - 1 - j] with no bounds check */
    if (fir->order > fir->have || iir->order > iir->have)
        d->status |= MLP_ORA_ERR_ENVELOPE;

    for (i = 0; i < n; i++) {
        int64_t sum = 0;
        int32_t ssum, value;
        for (j = 0; j < fir->order; j++)
            sum += (int64_t)fir->coeff[j] * (int64_t)fir->state[j];
        for (j = 0; j < iir->order; j++)
            sum += (int64_t)iir->coeff[j] * (int64_t)iir->state[j];
        ssum = (int32_t)(sum >> shift);
        value = mask_q((int32_t)((uint32_t)ssum + (uint32_t)res[i]), q);
        dst[i] = value;
        memmove(fir->state + 1, fir->state, (HIST - 1) * sizeof(int));
        fir->state[0] = value;
        if (fir->have < HIST)
            fir->have++;
        memmove(iir->state + 1, iir->state, (HIST - 1) * sizeof(int));
        iir->state[0] = (int32_t)((uint32_t)value - (uint32_t)ssum);
        if (iir->have < HIST)
            iir->have++;
    }
    return 1;
}

/* mlp.c:741-807 + 1122-1241; returns block size or 0 */
static unsigned decode_block(mlp_oracle *d, substream_t *s, bits_t *b)
{
    unsigned c, m, i, n;
    unsigned lsb_bits[MAX_CHANNELS];
    int sho[MAX_CHANNELS];
    unsigned envelope = 0;

    if (rd(b, 1)) {
        int restart = (int)rd(b, 1);
        if (restart && !parse_restart_header(b, s)) {
            d->status |= MLP_ORA_ERR_RESTART;
            return 0;
        }
        if (!parse_decoding_params(b, s, restart, &envelope)) {
            d->status |= envelope ? MLP_ORA_ERR_ENVELOPE : MLP_ORA_ERR_PARAMS;
            return 0;
        }
        if (envelope)
            d->status |= MLP_ORA_ERR_ENVELOPE;
    }
    if (b->eof)
        return 0;
    if (s->max_ch >= MAX_CHANNELS) {
        d->status |= MLP_ORA_ERR_ENVELOPE;
        return 0;
    }
    n = s->block_size;

    /* signed huffman offsets, mlp.c:1152-1176 */
    for (c = s->min_ch; c <= s->max_ch; c++) {
        const chparams_t *cp = &s->ch[c];
        int sign_shift;
        if (cp->huffman_lsbs < s->qss[c]) {
            d->status |= MLP_ORA_ERR_ENVELOPE; /* unsigned underflow there */
            return 0;
        }
        lsb_bits[c] = cp->huffman_lsbs - s->qss[c];
        if (cp->codebook) {
            sign_shift = (int)lsb_bits[c] + 2 - (int)cp->codebook;
            sho[c] = cp->huffman_offset - 7 * (1 << lsb_bits[c]) -
                     (sign_shift >= 0 ? (1 << sign_shift) : 0);
        } else {
            sign_shift = (int)lsb_bits[c] - 1;
            sho[c] = cp->huffman_offset - (sign_shift >= 0 ? (1 << sign_shift) : 0);
        }
    }

    if ((size_t)n * MAX_CHANNELS > d->residual_cap) {
        int32_t *nr = (int32_t *)realloc(d->residual, (size_t)n * MAX_CHANNELS * sizeof(int32_t));
        if (!nr)
            return 0;
        d->residual = nr;
        d->residual_cap = (size_t)n * MAX_CHANNELS;
    }
    for (m = 0; m < s->matrix_len; m++) {
        matrix_t *mp = &s->matrix[m];
        if (mp->bypassed_len + n > mp->bypassed_cap) {
            size_t nc = (mp->bypassed_len + n) * 2;
            int *nb = (int *)realloc(mp->bypassed, nc * sizeof(int));
            if (!nb)
                return 0;
            mp->bypassed = nb;
            mp->bypassed_cap = nc;
        }
    }

    /* residual rows, mlp.c:1194-1238 */
    for (i = 0; i < n; i++) {
        for (m = 0; m < s->matrix_len; m++) {
            matrix_t *mp = &s->matrix[m];
            mp->bypassed[mp->bypassed_len++] = mp->lsb_bypass ? (int)rd(b, 1) : 0;
        }
        for (c = s->min_ch; c <= s->max_ch; c++) {
            const chparams_t *cp = &s->ch[c];
            int msb = 0;
            uint32_t lsb;
            if (cp->codebook) {
                /* read_huffman_code == 9-bit peek LUT (bitstream.c:1806-1833) */
                size_t left = b->nbits - b->pos;
                unsigned peek, avail = left >= 9 ? 9 : (unsigned)left;
                const huff_entry *e;
                bits_t t = *b;
                peek = rd(&t, avail) << (9 - avail);
                e = &HUFF[cp->codebook][peek];
                if (e->len > avail) {
                    b->eof = 1;
                    return 0;
                }
                b->pos += e->len;
                msb = e->value;
                if (msb < 0) {
                    d->status |= MLP_ORA_ERR_HUFFMAN;
                    return 0;
                }
            }
            lsb = rd(b, lsb_bits[c]);
            d->residual[(size_t)c * n + i] = (int32_t)(
                ((uint32_t)(msb << lsb_bits[c]) + lsb + (uint32_t)sho[c]) << s->qss[c]);
        }
        if (b->eof)
            return 0;
    }

    /* filter each channel and append to the frame list, mlp.c:789-804 */
    for (c = s->min_ch; c <= s->max_ch; c++) {
        vec_t *fl = &d->frame[c];
        size_t at = fl->len;
        if (!vec_push_n(fl, NULL, n))
            return 0;
        if (!filter_channel(d, d->residual + (size_t)c * n, n, &s->ch[c].fir, &s->ch[c].iir,
                            s->qss[c], fl->v + at)) {
            d->status |= MLP_ORA_ERR_FILTER;
            return 0;
        }
    }
    return n;
}

/* mlp.c:714-739 */
static unsigned decode_substream(mlp_oracle *d, substream_t *s, bits_t *b)
{
    unsigned total = 0;
    do {
        unsigned n = decode_block(d, s, b);
        if (n == 0)
            return total;
        total += n;
    } while (rd(b, 1) == 0 && !b->eof);
    /* byte align, then an optional 32-bit end-of-stream word is consumed */
    b->pos = (b->pos + 7) & ~(size_t)7;
    if (bytes_left(b) >= 4)
        (void)rd(b, 32);
    return total;
}

/* mlp.c:1308-1358 */
static void rematrix(mlp_oracle *d, substream_t *s)
{
    const size_t rows = d->frame[0].len;
    size_t i;
    unsigned m, c;
    int32_t *n0, *n1;
    if (s->max_matrix_ch >= MAX_CHANNELS) {
        d->status |= MLP_ORA_ERR_ENVELOPE;
        return;
    }
    n0 = (int32_t *)malloc((rows ? rows : 1) * 2 * sizeof(int32_t));
    if (!n0)
        return;
    n1 = n0 + rows;
    for (i = 0; i < rows; i++) {
        const uint32_t seed = s->noise_seed;
        const uint32_t shifted = (seed >> 7) & 0xFFFF;
        n0[i] = (int32_t)((uint32_t)(int32_t)(int8_t)(seed >> 15) << s->noise_shift);
        n1[i] = (int32_t)((uint32_t)(int32_t)(int8_t)shifted << s->noise_shift);
        s->noise_seed = (seed << 16) ^ shifted ^ (shifted << 5);
    }
    for (m = 0; m < s->matrix_len; m++) {
        const matrix_t *mp = &s->matrix[m];
        for (c = 0; c <= s->max_matrix_ch; c++)
            if (d->frame[c].len < rows)
                d->status |= MLP_ORA_ERR_ENVELOPE; /* reads an empty channel there */
        if (mp->bypassed_len < rows)
            d->status |= MLP_ORA_ERR_ENVELOPE;     /* reads stale bypass bits there */
        if (mp->out_channel >= MAX_CHANNELS) {     /* matrix kept from a wider restart header: the
                                                      reference writes past framelist[8] there */
            d->status |= MLP_ORA_ERR_ENVELOPE;
            continue;
        }
        for (i = 0; i < rows; i++) {
            int64_t sum = 0;
            for (c = 0; c <= s->max_matrix_ch; c++)
                if (i < d->frame[c].len)
                    sum += (int64_t)d->frame[c].v[i] * (int64_t)mp->coeff[c];
            sum += (int64_t)n0[i] * (int64_t)mp->coeff[s->max_matrix_ch + 1];
            sum += (int64_t)n1[i] * (int64_t)mp->coeff[s->max_matrix_ch + 2];
            if (i < d->frame[mp->out_channel].len)
                d->frame[mp->out_channel].v[i] =
                    (int32_t)((uint32_t)mask_q((int32_t)(sum >> 14), s->qss[mp->out_channel]) +
                              (uint32_t)(i < mp->bypassed_len ? mp->bypassed[i] : 0));
        }
    }
    free(n0);
}

/* ------------------------------------------------------------ frame decode */

/* mlp.c:670-712: verifies the trailer of a substream of `len` bytes at p */
static int check_substream(mlp_oracle *d, const uint8_t *p, size_t len)
{
    uint8_t parity = 0, crc = 0x3C, final_crc = 0;
    size_t i;
    for (i = 0; i + 2 < len; i++) {
        parity ^= p[i];
        final_crc = crc ^ p[i];
        crc = CRC8[final_crc];
    }
    if ((uint8_t)(p[len - 2] ^ parity) != 0xA9) {
        d->status |= MLP_ORA_ERR_PARITY;
        return 0;
    }
    if (final_crc != p[len - 1]) {
        d->status |= MLP_ORA_ERR_CRC;
        return 0;
    }
    return 1;
}

/* mlp.c:407-612.  `p` points just past the 4-byte frame header. */
static unsigned decode_frame(mlp_oracle *d, const uint8_t *p, size_t len)
{
    bits_t b;
    unsigned s, c, m, pcm0 = 0;
    size_t ss_start, prev_end = 0;
    substream_t *last;

    bits_init(&b, p, len);

    /* major sync, mlp.c:614-654: 28 bytes, validated, else rewound */
    if (len >= 28) {
        bits_t t = b;
        unsigned sync = rd(&t, 24), type = rd(&t, 8);
        unsigned g0b = rd(&t, 4), g1b = rd(&t, 4), g0r = rd(&t, 4), g1r = rd(&t, 4);
        unsigned asg, count;
        skip(&t, 11);
        asg = rd(&t, 5);
        skip(&t, 48);
        (void)rd(&t, 1);
        (void)rd(&t, 15);
        count = rd(&t, 4);
        skip(&t, 92);
        if (sync == 0xF8726F && type == 0xBB && (count == 1 || count == 2)) {
            b = t;
            if (d->sync_read) {
                if (g0b != d->g0_bps || g1b != d->g1_bps || g0r != d->g0_rate ||
                    g1r != d->g1_rate || asg != d->assignment) {
                    d->status |= MLP_ORA_ERR_SYNC_CHANGE;
                    return 0;
                }
            } else {
                d->g0_bps = g0b;
                d->g1_bps = g1b;
                d->g0_rate = g0r;
                d->g1_rate = g1r;
                d->assignment = asg;
                d->substream_count = count;
                d->sync_read = 1;
            }
        }
    }
    if (!d->sync_read) {
        d->status |= MLP_ORA_ERR_NO_SYNC; /* substream_count uninitialised there */
        return 0;
    }

    /* substream info, mlp.c:462-468, 656-668 */
    for (s = 0; s < d->substream_count; s++) {
        substream_t *ss = &d->ss[s];
        ss->extraword = rd(&b, 1);
        ss->nonrestart = rd(&b, 1);
        ss->checkdata = rd(&b, 1);
        (void)rd(&b, 1);
        ss->end = rd(&b, 12) * 2;
        if (ss->extraword)
            skip(&b, 16);
    }
    if (b.eof) {
        d->status |= MLP_ORA_ERR_EOF;
        return 0;
    }
    ss_start = b.pos >> 3;
    last = &d->ss[d->substream_count - 1];

    for (s = 0; s < d->substream_count; s++) {
        substream_t *ss = &d->ss[s];
        /* substream 1 is checked with substream 0's flag (mlp.c:545) */
        const unsigned check = d->ss[0].checkdata;
        size_t sslen, datalen;
        bits_t sb;
        unsigned frames;

        if (ss->end < prev_end) {
            d->status |= MLP_ORA_ERR_EOF; /* negative length there */
            return 0;
        }
        sslen = ss->end - prev_end;
        if (ss_start + sslen > len || (check && sslen < 2)) {
            d->status |= MLP_ORA_ERR_EOF;
            return 0;
        }
        datalen = sslen;
        if (check) {
            if (!check_substream(d, p + ss_start, sslen))
                return 0;
            datalen = sslen - 2;
        }
        for (m = 0; m < MAX_MATRICES; m++)
            ss->matrix[m].bypassed_len = 0;

        bits_init(&sb, p + ss_start, datalen);
        frames = decode_substream(d, ss, &sb);
        if (sb.eof) {
            d->status |= MLP_ORA_ERR_EOF;
            return 0;
        }
        if (!frames)
            return 0;
        if (s == 0)
            pcm0 = frames;
        else if (frames != pcm0) {
            /* The two substreams of an access unit disagree on its length.  The reference rematrixes over
               channels[0]'s length and reads the other substream's channels past theirs (mlp.c:1308-1320,
               1342-1344), then extends every output channel by its own length (mlp.c:598-603): channels of
               different lengths, which dvda_read() interleaves against each other from there on
               (dvd-audio.c:781-792).  Outside what it defines: reported, not decoded. */
            d->status |= MLP_ORA_ERR_ENVELOPE;
            return 0;
        }
        ss_start += sslen;
        prev_end = ss->end;
    }

    /* rematrix / shift / output with the LAST substream's parameters as they
       stand after the frame's last block, mlp.c:504-538 / 575-609 */
    rematrix(d, last);
    for (c = 0; c <= last->max_matrix_ch && c < MAX_CHANNELS; c++) {
        const unsigned sh = last->output_shift[c];
        if (sh) {
            size_t i;
            for (i = 0; i < d->frame[c].len; i++)
                d->frame[c].v[i] = (int32_t)((uint32_t)d->frame[c].v[i] << (sh & 31));
        }
    }
    for (c = 0; c < d->nch; c++) {
        const int w = wave_channel(d->assignment, c);
        if (w < 0 || (unsigned)w >= d->nch) {
            d->status |= MLP_ORA_ERR_ENVELOPE;
            continue;
        }
        vec_push_n(&d->out[w], d->frame[c].v, d->frame[c].len);
    }
    for (c = 0; c < MAX_CHANNELS; c++)
        d->frame[c].len = 0;
    return pcm0;
}

/* ------------------------------------------------------------------ public */

mlp_oracle *mlp_oracle_open(unsigned nch)
{
    mlp_oracle *d;
    build_tables();
    if (nch == 0 || nch > 6)
        return NULL;
    d = (mlp_oracle *)calloc(1, sizeof(*d));
    if (d)
        d->nch = nch;
    return d;
}

void mlp_oracle_close(mlp_oracle *d)
{
    unsigned s, m, c;
    if (!d)
        return;
    for (s = 0; s < MAX_SUBSTREAMS; s++)
        for (m = 0; m < MAX_MATRICES; m++)
            free(d->ss[s].matrix[m].bypassed);
    for (c = 0; c < MAX_CHANNELS; c++) {
        free(d->frame[c].v);
        free(d->out[c].v);
    }
    free(d->residual);
    free(d->q);
    free(d);
}

/* mlp.c:344-405: enqueue everything, then decode while a whole frame is queued */
unsigned mlp_oracle_decode_packet(mlp_oracle *d, const uint8_t *data, size_t len)
{
    unsigned total = 0;
    if (d->qpos && d->qpos == d->qlen)
        d->qpos = d->qlen = 0;
    if (d->qlen + len > d->qcap) {
        size_t live = d->qlen - d->qpos;
        size_t nc = d->qcap ? d->qcap : 4096;
        uint8_t *nq;
        while (nc < live + len)
            nc *= 2;
        nq = (uint8_t *)malloc(nc);
        if (!nq)
            return 0;
        if (live)
            memcpy(nq, d->q + d->qpos, live);
        free(d->q);
        d->q = nq;
        d->qcap = nc;
        d->qlen = live;
        d->qpos = 0;
    }
    if (len)
        memcpy(d->q + d->qlen, data, len);
    d->qlen += len;

    for (;;) {
        const size_t avail = d->qlen - d->qpos;
        const uint8_t *h = d->q + d->qpos;
        size_t total_bytes;
        if (avail < 4)
            break;
        total_bytes = 2 * ((((size_t)h[0] & 0x0F) << 8) | h[1]);
        /* the reference asks for total-4 bytes as an unsigned count: a size
           field below 2 words wraps and is never satisfiable, stalling there */
        if (total_bytes < 4) {
            d->status |= MLP_ORA_ERR_EOF;
            break;
        }
        if (avail < total_bytes)
            break;
        total += decode_frame(d, h + 4, total_bytes - 4);
        d->qpos += total_bytes;
    }
    return total;
}

unsigned mlp_oracle_status(const mlp_oracle *d) { return d->status; }
size_t mlp_oracle_channel_len(const mlp_oracle *d, unsigned c) { return c < d->nch ? d->out[c].len : 0; }
const int32_t *mlp_oracle_channel(const mlp_oracle *d, unsigned c) { return c < d->nch ? d->out[c].v : NULL; }
size_t mlp_oracle_queued_bytes(const mlp_oracle *d) { return d->qlen - d->qpos; }

long mlp_oracle_decode(const uint8_t *data, size_t len, size_t chunk, unsigned nch, int32_t *out,
                       size_t cap, unsigned *status)
{
    mlp_oracle *d = mlp_oracle_open(nch);
    size_t off = 0;
    long total = 0;
    unsigned c;
    if (!d)
        return -1;
    if (chunk == 0)
        chunk = len ? len : 1;
    while (off < len) {
        size_t n = len - off < chunk ? len - off : chunk;
        total += mlp_oracle_decode_packet(d, data + off, n);
        off += n;
    }
    for (c = 0; c < nch; c++) {
        if (d->out[c].len > cap) {
            total = -1;
            break;
        }
        if (d->out[c].len)
            memcpy(out + (size_t)c * cap, d->out[c].v, d->out[c].len * sizeof(int32_t));
    }
    if (status)
        *status = d->status;
    mlp_oracle_close(d);
    return total;
}

/* ------------------------------------------------------------ test hooks */
/* reads n fields (width > 0: unsigned, width < 0: signed of -width bits) */
int mlp_oracle_test_read(const uint8_t *data, size_t len, const int *widths, int n, long *out)
{
    bits_t b;
    int i;
    bits_init(&b, data, len);
    for (i = 0; i < n; i++)
        out[i] = widths[i] >= 0 ? (long)rd(&b, (unsigned)widths[i]) : (long)rd_signed(&b, (unsigned)-widths[i]);
    return b.eof;
}

int mlp_oracle_test_crc8(unsigned i)
{
    build_tables();
    return CRC8[i & 0xFF];
}

/* (value & 0xFF) | length << 8 for a 9-bit peek of code book 1..3 */
int mlp_oracle_test_huff(unsigned book, unsigned peek9)
{
    build_tables();
    if (book < 1 || book > 3)
        return -1;
    return ((int)HUFF[book][peek9 & 511].value & 0xFF) | (HUFF[book][peek9 & 511].len << 8);
}
