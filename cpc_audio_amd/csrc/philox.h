// Philox4x32-10 (the counter-based generator torch uses on GPUs) and the numbering of the dropout sites of the transformer
// layer (cpc/transformers.py:18,50 and :93,100): a keep decision is a pure function of (seed, site, element), so the backward
// regenerates it and a test can ask for exactly the mask a call used (cpc_dropout_keep_mask).  Shared by transformer.hip and the
// GEMM epilogue that applies the hidden layer's dropout (gemm.hip).
#pragma once
#include "cpc_common.h"

namespace cpc {

struct Philox4 { unsigned x, y, z, w; };
__device__ __forceinline__ unsigned mulhi32(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
__device__ __forceinline__ Philox4 philox4x32_10(unsigned long long seed, unsigned site, unsigned long long ctr) {
    Philox4 c{(unsigned)ctr, (unsigned)(ctr >> 32), site, 0u};
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned hi0 = mulhi32(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
        const unsigned hi1 = mulhi32(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
        c = Philox4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// an element is dropped iff its 32 random bits fall below p * 2^32
__device__ __forceinline__ unsigned drop_threshold(float p) { return (unsigned)((double)p * 4294967296.0); }
// Site 1 (feed-forward hidden layer, rows of kFfnWidth): a Philox block decides EIGHT elements -- four consecutive ROWS (word
// row & 3) of the two columns c and c + 32 (c & 32 == 0: the low / high 16 bits of the word), which is what a lane of a GEMM tile's
// accumulators holds in two neighbouring 32-column tiles.  An element is dropped iff its 16 bits fall below p * 2^16 (p = 0.1:
// 6553 / 65536 = 0.09999 -- a keep probability within 1.1e-5 of 1 - p; the survivors are scaled by the nominal 1 / (1 - p)).
// Round 6: lin1's fused epilogue (gemm_dma.hip) spent most of its time in one ten-round block per FOUR elements -- half of that now.
constexpr int kFfnWidth = 2048;
__device__ __forceinline__ unsigned long long ffn_drop_block(long row, int col) {
    return (unsigned long long)(row >> 2) * (kFfnWidth / 2) + (unsigned)(((col >> 6) << 5) | (col & 31));
}
__device__ __forceinline__ unsigned drop_threshold16(float p) { return (unsigned)((double)p * 65536.0); }
__device__ __forceinline__ unsigned philox_word(const Philox4& r, int word) {
    return word == 0 ? r.x : (word == 1 ? r.y : (word == 2 ? r.z : r.w));
}

// the 16 bits of element (row with row & 3 == word, col) in its block's draw
__device__ __forceinline__ unsigned ffn_drop_field(const Philox4& r, int word, int col) {
    const unsigned w = philox_word(r, word);
    return (col & 32) ? (w >> 16) : (w & 0xFFFFu);
}

}  // namespace cpc
