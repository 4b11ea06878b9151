// Dropout on the attention probabilities (transformer/MFT/multiTransformer.py:32-33) as a stored BIT MASK.
//
// Round 1 evaluated a counter hash per score inside every attention kernel, forward and backward: 13 cycles of SIMD issue per
// hash word plus 6 per decision for the compare-and-select, 40 % of the forward's and a third of the backward's instruction
// stream (tools/valu_micro.hip prices the instructions).  Now one generator launch per forward pass draws every decision of
// every layer once, 32 decisions per instruction, and stores them as lane masks; a consumer kernel loads the 16 masks of a
// 32x32 score tile with two scalar loads (s_load_dwordx16) and applies each with ONE v_cndmask_b32 whose select operand is
// the SGPR pair — no VGPR, no compare, no hash.
//
// Bernoulli words.  A 32-bit word whose bits are independent drops with probability thr16 / 65536 is built from up to 16
// uniform random words, least significant threshold bit first:  D = bit_b ? (D | R_b) : (D & R_b)   (each step halves the
// distance to the target probability).  Threshold bits below the lowest set one leave D = 0 and are skipped (p = 0.1:
// thr16 = 0x199A, 15 words; p = 0.25: 2 words).  R_b = drop_fin(drop_lin(...)) — the counter hash of common.h — of the word
// index ((q tile * nt + k tile) * 32 + key) * 16 + b inside the per-(batch, head) substream, so a mask is still a pure
// function of (seed, layer stream, batch*head, query, key) and mmt_debug_dropout_mask replays it.
//
// Layouts (uint32 words; nt = Tp / 32 tiles per axis).  One generator lane owns one 32x32 block and writes it twice:
//   MQ [bh][q tile][k tile][32]: word 2i + hh = keep bits of key (i&3) + 8(i>>2) + 4hh over the tile's 32 queries (bit = query):
//      the 64-bit lane mask of accumulator register i for kernels that keep the QUERY on the lane (forward, dQ);
//   MK [bh][k tile][q tile][32]: the transposed block in the same register order (bit = key): lane masks for kernels that
//      keep the KEY on the lane (dK/dV, the one-kernel backward).  The 32x32 bit transpose runs in registers (5 butterfly
//      stages), 64 blocks per wave at once.
#pragma once
#include "common.h"

#define MMT_MASK_BLOCK_WORDS 32

// keep bits of one 32x32 block: W[key] bit q = 1 iff (query q, key) of the block is kept
__device__ __forceinline__ void attn_keep_block(const DropCfg& dc, uint32_t blk, uint32_t (&W)[32]) {
    const uint32_t thr = dc.thr16;
    const int b0 = thr ? __builtin_ctz(thr) : 16;                 // wave-uniform
    uint32_t x = drop_lin(dc.s0, blk * 512u) + (uint32_t)b0 * MMT_DROP_C1;   // word index blk*512 + key*16 + b  (< 2^24 for Tp <= 4096)
#pragma unroll
    for (int key = 0; key < 32; ++key) W[key] = 0;                // the drop words D; threshold bits below b0 leave them 0
    // threshold bit outermost (a uniform, rolled loop), the 32 keys unrolled inside: 32 independent hash chains in flight
#pragma unroll 1
    for (int b = b0; b < 16; ++b) {
        if ((thr >> b) & 1u) {
#pragma unroll
            for (int key = 0; key < 32; ++key) W[key] |= drop_fin(x + (uint32_t)(16 * key) * MMT_DROP_C1, dc.s1);
        } else {
#pragma unroll
            for (int key = 0; key < 32; ++key) W[key] &= drop_fin(x + (uint32_t)(16 * key) * MMT_DROP_C1, dc.s1);
        }
        x += MMT_DROP_C1;
    }
#pragma unroll
    for (int key = 0; key < 32; ++key) W[key] = ~W[key];
}

// in-register transpose of a 32x32 bit matrix: on return A[i] bit j = old A[j] bit i
__device__ __forceinline__ void transpose32(uint32_t (&A)[32]) {
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const int j = 16 >> s;
        const uint32_t m = (s == 0) ? 0x0000FFFFu : (s == 1) ? 0x00FF00FFu : (s == 2) ? 0x0F0F0F0Fu : (s == 3) ? 0x33333333u : 0x55555555u;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (k & j) continue;
            // rows k and k+j exchange the off-diagonal j x j bit blocks
            const uint32_t t = ((A[k] >> j) ^ A[k + j]) & m;
            A[k + j] ^= t;
            A[k] ^= t << j;
        }
    }
}

// position of row `x` (a key for MQ, a query for MK) inside a stored block: the order of a 32x32 MFMA accumulator's registers
__device__ __forceinline__ int mask_word_pos(int x) { return 2 * ((x & 3) + 4 * (x >> 3)) + ((x >> 2) & 1); }

struct MaskGenParams {
    uint32_t* mq; uint32_t* mk;            // layer l at + l * layer_words
    size_t layer_words;
    int nbh, nt, nlayers;
    uint32_t thr16;
    uint32_t s0[16], s1[16];               // stream keys of the layers' attention dropout (make_drop(p, seed, 4l+0))
};

__device__ __forceinline__ void store_block(uint32_t* dst, const uint32_t (&W)[32]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {          // 16-byte pieces: words 4c .. 4c+3 = registers i = 2c, 2c+1, both halves
        u32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int p = 4 * c + e, i = p >> 1, hh = p & 1;
            v[e] = W[(i & 3) + 8 * (i >> 2) + 4 * hh];
        }
        *reinterpret_cast<u32x4_t*>(dst + 4 * c) = v;
    }
}

// one lane per 32x32 block; lanes of a wave walk the k tiles of one (bh, q tile) first: MQ blocks of a wave are contiguous
__global__ __launch_bounds__(256, 4) void attn_mask_gen_kernel(const MaskGenParams P) {
    const int layer = blockIdx.y;
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t per_bh = (size_t)P.nt * P.nt;
    const int bh = (int)(g / per_bh);
    if (bh >= P.nbh) return;
    const int rem = (int)(g - (size_t)bh * per_bh), qt = rem / P.nt, kt = rem - qt * P.nt;
    DropCfg base; base.thr16 = P.thr16; base.scale = 1.f; base.s0 = P.s0[layer]; base.s1 = P.s1[layer];
    const DropCfg dc = drop_substream(base, (uint32_t)bh);
    uint32_t W[32];
    attn_keep_block(dc, (uint32_t)(qt * P.nt + kt), W);
    store_block(P.mq + (size_t)layer * P.layer_words + ((size_t)bh * per_bh + (size_t)qt * P.nt + kt) * 32, W);
    transpose32(W);                        // W[query] bit key
    store_block(P.mk + (size_t)layer * P.layer_words + ((size_t)bh * per_bh + (size_t)kt * P.nt + qt) * 32, W);
}

__host__ inline size_t attn_mask_layer_words(int nbh, int nt) { return (size_t)nbh * nt * nt * MMT_MASK_BLOCK_WORDS; }

// ---- consumer side -------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) uint64_t* mask_cptr;       // constant address space: uniform loads become s_load
struct TileMask { uint64_t m[16]; };
// the 16 lane masks of block `blk` of a row of blocks starting at `row` (64-bit words, 16 per block); `row` must be wave-uniform
__device__ __forceinline__ TileMask load_tile_mask(const uint64_t* row, int blk) {
    TileMask t;
    mask_cptr p = (mask_cptr)(uintptr_t)(row + (size_t)blk * 16);
#pragma unroll
    for (int i = 0; i < 16; ++i) t.m[i] = p[i];
    return t;
}
// lane-masked select with the mask in an SGPR pair: mask bit ? v : 0
__device__ __forceinline__ float keep_sel(float v, uint64_t m) {
    float r;
    asm("v_cndmask_b32 %0, 0, %1, %2" : "=v"(r) : "v"(v), "s"(m));
    return r;
}

// The same select applied to accumulator registers [i0, i0+8) IN PLACE, for values fresh from v_exp_f32: gfx950 wants one wait
// state between a transcendental and a VALU instruction that reads its result, and hipcc's hazard recognizer does not look
// inside asm statements — the leading s_nop supplies it once per eight selects.
__device__ __forceinline__ void keep_sel8(f32x16& v, int i0, const TileMask& t) {
    float a0 = v[i0], a1 = v[i0 + 1], a2 = v[i0 + 2], a3 = v[i0 + 3], a4 = v[i0 + 4], a5 = v[i0 + 5], a6 = v[i0 + 6], a7 = v[i0 + 7];
    asm("s_nop 0\n\t"
        "v_cndmask_b32 %0, 0, %0, %8\n\tv_cndmask_b32 %1, 0, %1, %9\n\tv_cndmask_b32 %2, 0, %2, %10\n\tv_cndmask_b32 %3, 0, %3, %11\n\t"
        "v_cndmask_b32 %4, 0, %4, %12\n\tv_cndmask_b32 %5, 0, %5, %13\n\tv_cndmask_b32 %6, 0, %6, %14\n\tv_cndmask_b32 %7, 0, %7, %15"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
        : "s"(t.m[i0]), "s"(t.m[i0 + 1]), "s"(t.m[i0 + 2]), "s"(t.m[i0 + 3]), "s"(t.m[i0 + 4]), "s"(t.m[i0 + 5]), "s"(t.m[i0 + 6]), "s"(t.m[i0 + 7]));
    v[i0] = a0; v[i0 + 1] = a1; v[i0 + 2] = a2; v[i0 + 3] = a3; v[i0 + 4] = a4; v[i0 + 5] = a5; v[i0 + 6] = a6; v[i0 + 7] = a7;
}

// ---- test hook: expand the generator's decisions to one byte per (bh, query, key) ------------------------------------
__global__ void attn_mask_expand_kernel(DropCfg base, int nbh, int nt, uint8_t* __restrict__ keep) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_bh = (size_t)nt * nt;
    const int bh = (int)(g / per_bh);
    if (bh >= nbh) return;
    const int rem = (int)(g - (size_t)bh * per_bh), qt = rem / nt, kt = rem - qt * nt;
    const DropCfg dc = drop_substream(base, (uint32_t)bh);
    uint32_t W[32];
    attn_keep_block(dc, (uint32_t)(qt * nt + kt), W);
    const size_t Tp = (size_t)nt * 32;
#pragma unroll
    for (int key = 0; key < 32; ++key)
#pragma unroll 1
        for (int q = 0; q < 32; ++q)
            keep[((size_t)bh * Tp + qt * 32 + q) * Tp + kt * 32 + key] = (W[key] >> q) & 1u;
}
