/* oracle/mlp_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's MLP decode path (tuffy/libdvd-audio
 * src/mlp.c) used as the parity checker for the HIP path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the
 * product (libdvd-audio_amd/) never does.
 *
 * Pinning: the reference ships no MLP vectors, so this oracle
 * is pinned against the reference itself compiled in the dev container
 * (oracle/_ref/libdvda_ref.so, built by oracle/Makefile) over seeded synthetic
 * streams, and against the committed fixtures in tests/golden/ that were
 * produced by that same reference build (tests/golden/make_golden.py).
 */
#ifndef MLP_ORACLE_H
#define MLP_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* error bits accumulated in mlp_oracle_status(); 0 = every frame decoded the
 * way the reference decodes a well-formed stream.  Where the reference would
 * assert()/abort or read out of bounds, the oracle sets a bit instead. */
#define MLP_ORA_ERR_NO_SYNC      (1u << 0)  /* frame before any major sync           */
#define MLP_ORA_ERR_SYNC_CHANGE  (1u << 1)  /* later major sync differs: frame dropped (mlp.c:450-455) */
#define MLP_ORA_ERR_PARITY       (1u << 2)  /* mlp.c:692-697                         */
#define MLP_ORA_ERR_CRC          (1u << 3)  /* mlp.c:701-706                         */
#define MLP_ORA_ERR_EOF          (1u << 4)  /* read past frame/substream end         */
#define MLP_ORA_ERR_RESTART      (1u << 5)  /* mlp.c:834-848                         */
#define MLP_ORA_ERR_PARAMS       (1u << 6)  /* mlp.c:902,975,1009-1014,1035-1062...  */
#define MLP_ORA_ERR_HUFFMAN      (1u << 7)  /* invalid code, mlp.c:1228              */
#define MLP_ORA_ERR_FILTER       (1u << 8)  /* mlp.c:1260-1264                       */
#define MLP_ORA_ERR_ENVELOPE     (1u << 9)  /* reference would invoke UB*/

typedef struct mlp_oracle mlp_oracle;

mlp_oracle *mlp_oracle_open(unsigned nch);
void mlp_oracle_close(mlp_oracle *d);

/* appends `len` bytes to the decoder's queue and decodes every complete frame
 * (reference mlp.c:344-382).  Returns PCM frames decoded by this call. */
unsigned mlp_oracle_decode_packet(mlp_oracle *d, const uint8_t *data, size_t len);

unsigned mlp_oracle_status(const mlp_oracle *d);
size_t mlp_oracle_channel_len(const mlp_oracle *d, unsigned c);
const int32_t *mlp_oracle_channel(const mlp_oracle *d, unsigned c);
size_t mlp_oracle_queued_bytes(const mlp_oracle *d);

/* one-shot helper with the same shape as oracle/ref_driver.c:ref_mlp_decode.
 * out is planar [nch][cap].  Returns PCM frames, or -1 when cap is too small;
 * *status receives the error bits. */
long mlp_oracle_decode(const uint8_t *data, size_t len, size_t chunk,
                       unsigned nch, int32_t *out, size_t cap,
                       unsigned *status);

/* test hooks (bit-reader contract, tables) */
int mlp_oracle_test_read(const uint8_t *data, size_t len, const int *widths, int n, long *out);
int mlp_oracle_test_crc8(unsigned i);
int mlp_oracle_test_huff(unsigned book, unsigned peek9);

#ifdef __cplusplus
}
#endif
#endif
