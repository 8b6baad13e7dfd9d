/* mlp_synth.h -- synthetic MLP (Meridian Lossless Packing) stream generator.
 *
 * Tooling, not the hot path: the reference ships no MLP test vectors
 * (SURVEY.md section 4), so every parity test and the benchmark decode streams
 * produced here.  The generator does not compress audio; it draws decoding
 * parameters and residual symbols from a seeded LCG and writes them in the
 * exact syntax the reference parser accepts (reference src/mlp.c:384-405,
 * 614-668, 714-1241), staying inside the reference's well-definedness envelope
 *  so that the compiled reference can serve as oracle.
 */
#ifndef DVDA_MLP_SYNTH_H
#define DVDA_MLP_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* feature bits for mlp_synth_cfg.features (profile 1 = "fuzz") */
#define MLP_SF_IIR         (1u << 0)  /* feed-forward (IIR) taps with transmitted state   */
#define MLP_SF_QSS         (1u << 1)  /* non-zero quant_step_size                         */
#define MLP_SF_OUTSHIFT    (1u << 2)  /* non-zero output_shift                            */
#define MLP_SF_HUFFOFF     (1u << 3)  /* non-zero huffman_offset                          */
#define MLP_SF_VARBLOCK    (1u << 4)  /* random split of an AU into blocks                */
#define MLP_SF_VARROWS     (1u << 5)  /* PCM frames per AU vary (non-standard timing)     */
#define MLP_SF_MIDMATRIX   (1u << 6)  /* matrix/shift/qss updates in a non-first block    */
#define MLP_SF_CHAINED     (1u << 7)  /* no raw lead-in after segment 0: FIR state chains */
#define MLP_SF_EXTRAWORD   (1u << 8)  /* substream_info extraword present                 */
#define MLP_SF_MIDRESTART  (1u << 9)  /* restart headers in non-first blocks              */
#define MLP_SF_FLAGS       (1u << 10) /* random parameter-presence flag bytes             */
#define MLP_SF_MIXBOOKS    (1u << 11) /* per-channel random codebook / huffman_lsbs       */
#define MLP_SF_NOCHECK     (1u << 12) /* checkdata_present = 0                            */
#define MLP_SF_PARAMBLOCKS (1u << 13) /* parameter updates on arbitrary blocks            */
#define MLP_SF_MATRIXRAND  (1u << 14) /* random matrix count / fractional bits / sparsity */
#define MLP_SF_TERMINATOR  (1u << 15) /* 0xD234D234 marker after the last block           */
#define MLP_SF_FIRRAND     (1u << 16) /* random FIR order/shift/coefficients              */
#define MLP_SF_NOISE       (1u << 17) /* larger noise coefficients and noise_shift        */
/* (bits 18 and up are not part of the "all features" fuzz set: existing streams stay what they were) */
#define MLP_SF_CHECKQUIRK  (1u << 18) /* substream 1's checkdata_present flag says the OPPOSITE of substream 0's: the
                                         reference goes by substream 0's flag for both (src/mlp.c:545)               */
#define MLP_SF_DISC        (1u << 19) /* what an encoder writes: EVERY block carries parameters, most channels are
                                         sent and most of those re-send their FIR taps; fixed block positions       */
#define MLP_SF_SYNCONLY    (1u << 20) /* a major sync in front of some access units that carry NO restart header:
                                         the reference re-reads the sync's parameters and decodes on (src/mlp.c:449-460) */

typedef struct mlp_synth_cfg {
    uint32_t profile;          /* 0 = BASELINE.md recipe, 1 = fuzz (uses .features) */
    uint32_t features;
    uint32_t assignment;       /* 5-bit channel assignment, 0..20                   */
    uint32_t rate_code;        /* 0,1,2,8,9,10 -> 48,96,192,44.1,88.2,176.4 kHz     */
    uint32_t bps_code;         /* 0,1,2 -> 16,20,24 bits                            */
    uint32_t n_substreams;     /* 1 or 2                                            */
    uint32_t ss0_channels;     /* channels carried by substream 0 when 2 substreams */
    uint32_t n_aus;            /* access units (MLP frames) to emit                 */
    uint32_t restart_interval; /* AUs between major sync + restart header           */
    uint32_t blocks_per_au;    /* recipe: 2                                         */
    uint32_t fir_order;        /* recipe: 8 (taps of the BASELINE.md filter)        */
    uint32_t codebook;         /* recipe: 1                                         */
    uint32_t huffman_lsbs;     /* recipe: 12                                        */
    uint32_t n_matrices;       /* recipe: 2                                         */
    uint32_t reserved[2];
} mlp_synth_cfg;

/* fills *cfg with the BASELINE.md recipe for the given layout */
void mlp_synth_default(mlp_synth_cfg *cfg, uint32_t assignment,
                       uint32_t rate_code, uint32_t n_substreams,
                       uint32_t n_aus);

/* channel count of a channel assignment (reference src/dvd-audio.c:1459-1496) */
unsigned mlp_synth_channels(uint32_t assignment);

/* PCM frames per access unit at standard MLP timing for a rate code */
unsigned mlp_synth_rows_per_au(uint32_t rate_code);

/* safe upper bound for the byte size of the generated stream */
size_t mlp_synth_bound(const mlp_synth_cfg *cfg);

/* Generates one stream.  Returns bytes written (0 on error / cap too small).
 * *pcm_frames receives the number of PCM frames the stream decodes to. */
size_t mlp_synth_stream(const mlp_synth_cfg *cfg, uint64_t seed,
                        uint8_t *out, size_t cap, uint64_t *pcm_frames);

/* Generates `n` streams with seeds seed0 .. seed0+n-1 back to back into `out`
 * using up to `threads` worker threads.  offsets[n] receives byte offsets
 * (each stream start is 16-byte aligned, gaps are zero filled), sizes[n] the
 * exact byte length of each stream, frames[n] the PCM frame counts.
 * Returns total bytes including padding (0 on error). */
size_t mlp_synth_batch(const mlp_synth_cfg *cfg, uint64_t seed0, uint32_t n,
                       uint32_t threads, uint8_t *out, size_t cap,
                       uint64_t *offsets, uint64_t *sizes, uint64_t *frames);

#ifdef __cplusplus
}
#endif
#endif
