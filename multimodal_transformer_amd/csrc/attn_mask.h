// Dropout on the attention probabilities (transformer/MFT/multiTransformer.py:32-33) as a stored BIT MASK.
//
// Round 1 evaluated a counter hash per score inside every attention kernel, forward and backward: 13 cycles of SIMD issue per
// hash word plus 6 per decision for the compare-and-select, 40 % of the forward's and a third of the backward's instruction
// stream (tools/valu_micro.hip prices the instructions).  Now one generator launch per forward pass draws every decision of
// every layer once, 32 decisions per instruction, and stores them in the order the consumers hold their scores: a consumer lane
// loads ONE 16-bit word per 32x32 score tile (a tile ahead, with the tile's operand prefetch) and turns bit i into an all-ones /
// zero word with v_bfe_i32, which it ANDs onto accumulator register i: two full-rate instructions per register, no compare, no
// hash, no scalar registers.  (A first version kept the masks as 64-bit lane masks in SGPRs, loaded with s_load_dwordx16 and
// applied with one v_cndmask_b32 per register: fewer vector instructions, but scalar loads share the LDS wait counter, so every
// tile's first LDS wait exposed the whole scalar-load latency: +5 us per forward launch over eval mode instead of +1.)
//
// Bernoulli words.  A 32-bit word whose bits are independent drops with probability thr16 / 65536 is built from up to 16
// uniform random words, least significant threshold bit first:  D = bit_b ? (D | R_b) : (D & R_b)   (each step halves the
// distance to the target probability).  Threshold bits below the lowest set one leave D = 0 and are skipped (p = 0.1:
// thr16 = 0x199A, 15 words; p = 0.25: 2 words).  R_b = drop_fin(drop_lin(...)) — the counter hash of common.h — of the word
// index ((q tile * nt + k tile) * 32 + key) * 16 + b inside the per-(batch, head) substream, so a mask is still a pure
// function of (seed, layer stream, batch*head, query, key) and mmt_debug_dropout_mask replays it.
//
// Layouts (uint16 words; nt = Tp / 32 tiles per axis).  One generator lane owns one 32x32 block and writes it twice, 128 bytes each:
//   LQ [bh][q tile][k tile][64 lanes]: for kernels that keep the QUERY on the lane (forward, dQ): lane l = 32 hh + q holds, in bit i,
//      the decision for (query q, key (i&3) + 8(i>>2) + 4hh) — accumulator register i of that lane;
//   LK [bh][k tile][q tile][64 lanes]: for kernels that keep the KEY on the lane (dK/dV, the one-kernel backward): lane 32 hh + key,
//      bit j = (query (j&3) + 8(j>>2) + 4hh, key).
// LQ needs the block transposed (bits over keys): a 32x32 bit transpose in registers (5 butterfly stages), 64 blocks per wave.
#pragma once
#include "common.h"

#define MMT_MASK_BLOCK_WORDS 32

// keep bits of one 32x32 block: W[key] bit q = 1 iff (query q, key) of the block is kept
__device__ __forceinline__ void attn_keep_block(const DropCfg& dc, uint32_t blk, uint32_t (&W)[32]) {
    const uint32_t thr = dc.thr16;
    const int b0 = thr ? __builtin_ctz(thr) : 16;                 // wave-uniform
#pragma unroll
    for (int key = 0; key < 32; ++key) W[key] = 0;                // the drop words D; threshold bits below b0 leave them 0
    // threshold bit outermost (a uniform, rolled loop), the 32 keys unrolled inside: 32 independent hash chains in flight (124 VGPRs,
    // 4 waves per SIMD).  16 or 8 chains at 6 or 8 waves per SIMD (all 6144 waves of configs[3] resident at once) measured the same
    // 58-60 us: the kernel is bound by the total integer instruction issue, not by latency, occupancy or its tail.
    uint32_t x = drop_lin(dc.s0, blk * 512u) + (uint32_t)b0 * MMT_DROP_C1;   // word index blk*512 + key*16 + b  (< 2^24 for Tp <= 4096)
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

// the 16 bits a consumer lane of half hh needs out of a 32-bit row: bit i of the result = bit (i&3) + 8(i>>2) + 4hh of x
// (the even / odd nibbles of x, packed)
__device__ __forceinline__ uint32_t lane_word(uint32_t x, int hh) {
    uint32_t y = (x >> (4 * hh)) & 0x0F0F0F0Fu;
    y = (y | (y >> 4)) & 0x00FF00FFu;
    return (y | (y >> 8)) & 0x0000FFFFu;
}

struct MaskGenParams {
    uint16_t* lq; uint16_t* lk;            // layer l at + l * layer_words
    size_t layer_words;
    int nbh, nt, nlayers;
    uint32_t thr16;
    uint32_t s0[16], s1[16];               // stream keys of the layers' attention dropout (make_drop(p, seed, 4l+0))
    const uint64_t* seedword;              // non-null: device-resident seed (common.h drop_resolve); s0[l] then holds the stream id 4l+0
};

// rows[x] = 32 bits over the lane index (x = the register-side index): the 64 lane words of one block, 8 x 16 bytes, into this lane's
// row of the wave's LDS patch
#define MMT_MASK_LDS_ROW 72            // uint16 per patch row: 64 + 8 pad (144-byte rows: conflict-free 16-byte writes)
__device__ __forceinline__ void park_lane_block(uint16_t* row, const uint32_t (&rows)[32]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {          // piece c: lanes 8c .. 8c+7, i.e. half hh = c >> 2, lane-side index 8(c&3) .. +7
        u32x4_t v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int l0 = 8 * (c & 3) + 2 * e, hh = c >> 2;
            v[e] = lane_word(rows[l0], hh) | (lane_word(rows[l0 + 1], hh) << 16);
        }
        *reinterpret_cast<u32x4_t*>(row + 8 * c) = v;
    }
}

// One lane per 32x32 block; the lanes of a wave own 64 consecutive blocks in (bh, q tile, k tile) order.  A block is 128 bytes per
// orientation, so lane-private stores would be 64 separate 16-byte writes per instruction (6.5 M partial-line transactions per launch
// at configs[3]).  The wave parks its 64 blocks in LDS and writes them out with the lanes laid along memory (no change in the
// kernel's own time — it is bound by integer issue — but whole lines reach the fabric): LQ blocks of a wave are contiguous (1 KB per store instruction), LK blocks are
// written as whole 128-byte lines by 8 lanes each.
__device__ __forceinline__ void attn_mask_gen_block(const MaskGenParams& P, int layer, int bx) {
    __shared__ __attribute__((aligned(16))) uint16_t patch[4][64 * MMT_MASK_LDS_ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t g0 = (size_t)bx * 256 + wave * 64;                         // first block of the wave
    const size_t per_bh = (size_t)P.nt * P.nt, total = (size_t)P.nbh * per_bh;
    if (g0 >= total) return;                                                // whole wave (no workgroup barrier below)
    const size_t g = g0 + lane;
    const bool live = g < total;
    const size_t gc = live ? g : total - 1;
    const int bh = (int)(gc / per_bh);
    const int rem = (int)(gc - (size_t)bh * per_bh), qt = rem / P.nt, kt = rem - qt * P.nt;
    DropCfg base; base.thr16 = P.thr16; base.scale = 1.f; base.s0 = P.s0[layer]; base.s1 = P.s1[layer];
    const DropCfg dc = drop_substream(drop_resolve(base, P.seedword), (uint32_t)bh);
    uint32_t W[32];
    attn_keep_block(dc, (uint32_t)(qt * P.nt + kt), W);                    // W[key] bit query
    uint16_t* const mine = patch[wave] + lane * MMT_MASK_LDS_ROW;
    // flush role: in store instruction i this lane moves piece (lane & 7) of the wave's block 8 i + (lane >> 3)
    const int fp = lane & 7;
    const uint16_t* const from = patch[wave] + (lane >> 3) * MMT_MASK_LDS_ROW + 8 * fp;
    // ---- LK: key on the lane: lane-side index = key, its row = W[key] (bits over the queries = the register side)
    park_lane_block(mine, W);
    {
        // (bh, q tile, k tile) of block lane >> 3, then + 8 blocks per instruction
        size_t gb = g0 + (lane >> 3);
        int bb = (int)((gb < total ? gb : total - 1) / per_bh);
        int rb = (int)((gb < total ? gb : total - 1) - (size_t)bb * per_bh), qb = rb / P.nt, kb = rb - qb * P.nt;
        uint16_t* const lk = P.lk + (size_t)layer * P.layer_words;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(from + 8 * i * MMT_MASK_LDS_ROW);
            if (gb < total) *reinterpret_cast<u32x4_t*>(lk + ((size_t)bb * per_bh + (size_t)kb * P.nt + qb) * 64 + 8 * fp) = v;
            gb += 8; kb += 8;
            while (kb >= P.nt) { kb -= P.nt; if (++qb == P.nt) { qb = 0; ++bb; } }
        }
    }
    transpose32(W);                                                        // W[query] bit key
    // ---- LQ: block index == g: the wave's 64 blocks are contiguous in memory
    park_lane_block(mine, W);              // (same wave wrote and read the patch: program order is enough, no barrier)
    {
        uint16_t* const lq = P.lq + (size_t)layer * P.layer_words + g0 * 64;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(from + 8 * i * MMT_MASK_LDS_ROW);
            if (g0 + 8 * i + (lane >> 3) < total) *reinterpret_cast<u32x4_t*>(lq + (size_t)i * 512 + lane * 8) = v;
        }
    }
}

__global__ __launch_bounds__(256, 4) void attn_mask_gen_kernel(const MaskGenParams P) { attn_mask_gen_block(P, blockIdx.y, blockIdx.x); }

__host__ inline size_t attn_mask_layer_words(int nbh, int nt) { return (size_t)nbh * nt * nt * 64; }      // uint16 words per layer and orientation

// ---- consumer side -------------------------------------------------------------------------------------------------
// all-ones where bit I of the lane's tile word is set (kept), zero where it is clear (dropped): ONE v_bfe_i32.  Written as asm with
// the bit index as an immediate: from `(int)w << (31 - i) >> 31` or __builtin_amdgcn_sbfe hipcc 7.2 builds v_and (1 << i) + v_cmp_ne +
// v_cndmask for the AND that follows — three instructions per score where two do (found in the ISA of both attention hot loops).
// (Its source is a word loaded long before and its consumer an ordinary v_and: no MFMA / transcendental hazard is involved.)
template <int I> struct IdxC { static constexpr int value = I; };
template <int B, int E, class F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (B < E) { f(IdxC<B>{}); static_for<B + 1, E>(f); }
}
template <int I> __device__ __forceinline__ uint32_t keep_bits(uint32_t w) {
    uint32_t m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(w), "n"(I));
    return m;
}
template <int I> __device__ __forceinline__ float keep_and(float v, uint32_t w) {
    return __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, v) & keep_bits<I>(w));
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
