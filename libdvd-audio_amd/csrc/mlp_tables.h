// mlp_tables.h -- constant tables of the MLP decode path, built at compile time.
//
// Data sources (restated, not copied):
//   code books    reference src/mlp_codebook{1,2,3}.json      
//   CRC-8         reference src/mlp.c:1363-1395  (MSB-first, polynomial 0x63)
//   WAVE_CHANNEL  reference src/mlp.c:416-438
//   channel count reference src/dvd-audio.c:1459-1496
#pragma once
#include <stdint.h>

namespace mlp {

// ---------------------------------------------------------------- code books
// A 9-bit peek indexes (value | length << 8); value 0xFF marks the two invalid
// all-zero-tail codes of every book.  All three books share one structure:
//   "1" + (3-book) bits      -> 7 + bits          (length 4-book)
//   "0"^z "1", z = 2..8      -> 8 - z             (length z+1)
//   "01" "0"^k "1", k = 0..6 -> base(book) + k    (length k+3), base = 11, 9, 8
constexpr uint16_t huff_entry(int book, unsigned peek9)
{
    if (peek9 & 0x100) {
        const int sub = 3 - book;                       // extra bits after the leading 1
        const unsigned bits = (peek9 >> (8 - sub)) & ((1u << sub) - 1u);
        return (uint16_t)((7 + bits) | ((sub + 1) << 8));
    }
    if (peek9 & 0x080) {                                 // "01..."
        for (int k = 0; k <= 6; k++)
            if (peek9 & (0x040u >> k))
                return (uint16_t)(((book == 1 ? 11 : book == 2 ? 9 : 8) + k) | ((k + 3) << 8));
        return (uint16_t)(0xFF | (9 << 8));
    }
    for (int z = 2; z <= 8; z++)                         // "00..."
        if (peek9 & (0x100u >> z))
            return (uint16_t)((8 - z) | ((z + 1) << 8));
    return (uint16_t)(0xFF | (9 << 8));
}

struct HuffTable {
    uint16_t e[3 * 512];
};

constexpr HuffTable make_huff()
{
    HuffTable t{};
    for (int b = 1; b <= 3; b++)
        for (unsigned i = 0; i < 512; i++)
            t.e[(b - 1) * 512 + i] = huff_entry(b, i);
    return t;
}

// -------------------------------------------------------------------- CRC-8
// Slicing-by-4: T[k][x] = CRC state after byte x followed by k zero bytes.
struct CrcTable {
    uint8_t t[4 * 256];
};

constexpr CrcTable make_crc()
{
    CrcTable c{};
    for (unsigned i = 0; i < 256; i++) {
        unsigned v = i;
        for (int k = 0; k < 8; k++)
            v = (v & 0x80) ? (((v << 1) ^ 0x63) & 0xFF) : ((v << 1) & 0xFF);
        c.t[i] = (uint8_t)v;
    }
    for (int k = 1; k < 4; k++)
        for (unsigned i = 0; i < 256; i++)
            c.t[k * 256 + i] = c.t[c.t[(k - 1) * 256 + i]];
    return c;
}

// ---- CRC-8 as field arithmetic (mlp_check.h): 0x163 is primitive, so GF(2)[x] / 0x163 = GF(256) and x generates
//      its multiplicative group
struct CheckTables {
    uint8_t slice[16 * 256];        // slice[k * 256 + b] = b * x^(8 (k + 1))
    uint8_t log[256];               // log_x(v), v != 0
    uint8_t exp[512];               // x^i, i < 510 (period 255: no reduction of log + exponent needed)
};

constexpr CheckTables make_check()
{
    CheckTables t{};
    for (unsigned i = 0; i < 256; i++) {
        unsigned v = i;
        for (int k = 0; k < 8; k++)
            v = (v & 0x80) ? (((v << 1) ^ 0x63) & 0xFF) : ((v << 1) & 0xFF);
        t.slice[i] = (uint8_t)v;                                    // the reference's table, src/mlp.c:1363-1395
    }
    for (int k = 1; k < 16; k++)
        for (unsigned i = 0; i < 256; i++)
            t.slice[k * 256 + i] = t.slice[t.slice[(k - 1) * 256 + i]];
    unsigned v = 1;
    for (unsigned i = 0; i < 512; i++) {
        t.exp[i] = (uint8_t)v;
        if (i < 255)
            t.log[v] = (uint8_t)i;
        v = (v & 0x80) ? (((v << 1) ^ 0x63) & 0xFF) : ((v << 1) & 0xFF);
    }
    return t;
}

// ------------------------------------------------------------ channel tables
// wave_pack(assignment): nibble c = RIFF-WAVE index of MLP channel c, 0xF = none
constexpr uint32_t wave_pack(unsigned a)
{
    constexpr uint8_t count[21] = {1, 2, 3, 4, 3, 4, 5, 3, 4, 5, 4, 5, 6, 4, 5, 4, 5, 6, 5, 5, 6};
    if (a > 20)
        return 0xFFFFFFFFu;
    if (a == 0x12 || a == 0x13)
        return 0xFFF24310u;          // {0,1,3,4,2}
    if (a == 0x14)
        return 0xFF325410u;          // {0,1,4,5,2,3}
    uint32_t p = 0xFFFFFFFFu;
    for (unsigned c = 0; c < count[a]; c++)
        p = (p & ~(0xFu << (4 * c))) | (c << (4 * c));
    return p;
}

// wave_inv_pack(assignment): nibble w = MLP channel stored at RIFF-WAVE position w (inverse of wave_pack)
constexpr uint32_t wave_inv_pack(unsigned a)
{
    const uint32_t p = wave_pack(a);
    uint32_t inv = 0;
    for (unsigned c = 0; c < 6; c++) {
        const uint32_t w = (p >> (4 * c)) & 0xFu;
        if (w < 6)
            inv |= c << (4 * w);
    }
    return inv;
}

constexpr uint32_t channel_count(unsigned a)
{
    constexpr uint8_t count[21] = {1, 2, 3, 4, 3, 4, 5, 3, 4, 5, 4, 5, 6, 4, 5, 4, 5, 6, 5, 5, 6};
    return a <= 20 ? count[a] : 0;
}

// PCM frames per access unit at standard MLP timing for a major-sync rate code
constexpr uint32_t rows_per_au(unsigned rate_code)
{
    return (rate_code == 0 || rate_code == 8) ? 40u
         : (rate_code == 1 || rate_code == 9) ? 80u
         : (rate_code == 2 || rate_code == 10) ? 160u : 0u;
}

} // namespace mlp
