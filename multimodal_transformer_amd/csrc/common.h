// Shared device-side definitions for the gfx950 (MI355X / CDNA4) kernels.
//
// Conventions used by every kernel in this directory
// ---------------------------------------------------
//  * wavefront = 64 lanes; workgroups are 256 threads (4 waves, one per SIMD) unless stated.
//  * matrix products run on bf16 MFMA with fp32 accumulation; the residual stream, LayerNorm
//    statistics, softmax and every reduction stay fp32.
//  * "row-major [M][ld]" tensors are indexed by the flattened window index m = b*T + t.
//  * every operand exists in memory in ONE layout, window-major: row-major [M][ld] for the GEMM / weight-gradient operands and, for
//    the attention operands, the per-(batch,head) *fragment* layout
//      R layout  [tile][e>>3][t&31][e&7]
//    in which an MFMA operand fragment of a product that contracts over the head feature e is one lane-linear 16-byte load (1 KiB per
//    wave instruction).  Products that contract over WINDOWS (PV, dV, dK, dQ, the weight gradients) read the same tiles out of LDS
//    with gfx950's transposing read (tr_frag2 below), in the order an MFMA 32x32 accumulator presents its rows:
//    t&31 = 16 s + 8 (j>>2) + 4 hh + (j&3)   (see cdna_hip_programming.md 3 "An accumulator tile as the next MFMA's operand").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MMT_WAVE 64
#define MMT_THREADS 256

// D(16x16) += A(16x32) * B(32x16): lane l holds A[l&15][8(l>>4)+j], B[8(l>>4)+j][l&15];
// D: col = l&15, row = 4(l>>4) + reg.
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// D(32x32) += A(32x16) * B(16x32): lane l (r=l&31, hh=l>>5) holds A[r][8hh+j], B[8hh+j][r];
// D: col = r, row = (reg&3) + 8(reg>>2) + 4hh.
__device__ __forceinline__ f32x16 mfma32(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() would also drain vmcnt(0), i.e. wait at every
// step for loads prefetched for later steps and for the step's output stores (cdna_hip_programming.md §5 "Pipelining across
// barriers").  Used where data reaches LDS through registers (the ds_write's own register dependency waits for its load).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// row index (0..31) of accumulator register `reg` of a 32x32 MFMA result for lane-half hh
__device__ __forceinline__ int acc32_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }

// pack 8 accumulator registers [8s .. 8s+7] into one bf16 operand fragment: four v_cvt_pk_bf16_f32 on the
// register pairs (0,1),(2,3),... (written pair-wise: the element-wise form makes hipcc pair (1,2),(3,4),...
// and repair the result with v_alignbit/v_perm)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 pack8(const f32x16& v, int s) {
    u32x4_t r;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x2 p = {v[8 * s + 2 * j], v[8 * s + 2 * j + 1]};
        r[j] = __builtin_bit_cast(unsigned, __builtin_convertvector(p, bf16x2));
    }
    return __builtin_bit_cast(bf16x8, r);
}

// ---- transposing LDS reads (gfx950 ds_read_b64_tr_b16) ------------------------------------------------------------
// Every lane supplies the address of an 8-byte chunk = 4 consecutive bf16 features of ONE row (window); inside a 16-lane group lane
// 4q + p addresses row q of a 4-row block, features 4p .. 4p+3, and lane j of the group RECEIVES feature j of the block's four rows.
// Two reads give a lane the 8 contraction elements of an MFMA operand fragment whose contraction index is the row — the operand
// of a product that contracts over windows, read from a tile that is stored window-major.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
__device__ __forceinline__ bf16x8 tr_frag2(const bf16* p0, const bf16* p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)p1);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// rows q and q + 4 of a tile with `row_stride` elements between rows
__device__ __forceinline__ bf16x8 tr_frag(const bf16* p, int row_stride) { return tr_frag2(p, p + 4 * row_stride); }

// ---- fragment layout index maps (element offsets inside one (batch,head) matrix) ----------------
// R layout: tile-major, then 8-feature chunk, then window-in-tile, then feature-in-chunk.
__device__ __forceinline__ size_t fragR_index(int t, int e, int DKP) {
    return ((size_t)(t >> 5) * (DKP >> 3) + (e >> 3)) * 256 + (size_t)(t & 31) * 8 + (e & 7);
}
__host__ __device__ __forceinline__ size_t fragR_elems(int Tp, int DKP) { return (size_t)Tp * DKP; }

__host__ __device__ __forceinline__ int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- dropout: counter-based generator.  One 32-bit hash word serves the index pair (2i, 2i+1); each
// element compares its 16-bit half with thr16 = round(p * 65536): P(drop) = thr16/65536 (exact to 1.5e-5),
// kept values are scaled by 65536/(65536 - thr16).  A mask is a pure function of (seed, stream, index), so the
// backward pass regenerates it instead of storing it.
//   * stream keys (s0, s1) are mixed on the host (splitmix64 of seed and stream id);
//   * the per-word hash uses only FULL-RATE integer ops (v_mad_u32_u24, shifts, xor/add: 7 instructions;
//     v_mul_lo_u32 is quarter rate on CDNA) — the attention kernels are VALU-issue bound and evaluate it per score.
//     It has two parts: drop_lin, LINEAR in the 24-bit pair index (mod 2^32: (a + b) C1 = a C1 + b C1 while a + b < 2^24),
//     so a kernel that walks indices in a fixed pattern replaces the multiply by one add of a precomputed step
//     (lane-constant part + wave-uniform part), and drop_fin, the avalanche.  Callers fold higher index bits into s0.
struct DropCfg { uint32_t thr16; float scale; uint32_t s0, s1; };

#define MMT_DROP_C1 0xD2B54Bu
__device__ __forceinline__ uint32_t drop_lin(uint32_t s0, uint32_t pair) { return __umul24(pair, MMT_DROP_C1) + s0; }
__device__ __forceinline__ uint32_t drop_fin(uint32_t x, uint32_t s1) {
    x = (x ^ (x >> 15)) + s1;                       // v_lshrrev, v_xad_u32
    x = __umul24(x, 0x9E3779u) + (x >> 24);         // v_lshrrev, v_mad_u32_u24
    x ^= x >> 14;                                   // v_lshrrev, v_xor
    return x;
}
__device__ __forceinline__ uint32_t drop_word(uint32_t s0, uint32_t s1, uint32_t pair) { return drop_fin(drop_lin(s0, pair), s1); }
// strong (3 full multiplies) mixer for once-per-wave key derivation, e.g. the per-(batch,head) attention streams
__device__ __forceinline__ uint32_t mix32(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t x = c * 0x9E3779B1u + a;
    x ^= x >> 15; x *= 0x85EBCA6Bu;
    x ^= x >> 13; x += b; x *= 0xC2B2AE35u;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ DropCfg drop_substream(const DropCfg& c, uint32_t id) {
    DropCfg d = c;
    d.s0 = mix32(c.s0, c.s1, id);
    d.s1 = mix32(c.s1, c.s0, id ^ 0x5bd1e995u);
    return d;
}
__device__ __forceinline__ bool drop_keep(const DropCfg& c, uint64_t idx) {
    const uint32_t w = drop_word(c.s0 + __umul24((uint32_t)(idx >> 25), 0x4A7C15u), c.s1, (uint32_t)(idx >> 1));
    return ((idx & 1) ? (w >> 16) : (w & 0xFFFFu)) >= c.thr16;
}
// pair form for an EVEN index: one hash word decides idx (low half) and idx+1 (high half)
__device__ __forceinline__ uint32_t drop_pair(const DropCfg& c, uint64_t even_idx) {
    return drop_word(c.s0 + __umul24((uint32_t)(even_idx >> 25), 0x4A7C15u), c.s1, (uint32_t)(even_idx >> 1));
}
__device__ __forceinline__ float drop_lo(const DropCfg& c, uint32_t w, float v) { return (w & 0xFFFFu) >= c.thr16 ? v * c.scale : 0.f; }
__device__ __forceinline__ float drop_hi(const DropCfg& c, uint32_t w, float v) { return (w >> 16) >= c.thr16 ? v * c.scale : 0.f; }
// 32-bit index form used by the attention kernels (index = q*Tp + key inside one (batch,head) stream)
// (host code rejects dropout for Tp > 4096, so the pair index q*Tp/2 + key/2 always fits the 24-bit multiply)
__device__ __forceinline__ uint32_t drop_word_idx32(const DropCfg& c, uint32_t idx) { return drop_word(c.s0, c.s1, idx >> 1); }
// stream = which dropout site of which layer
__host__ __device__ inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// Device-resident seeds.  A step that is replayed from a hipGraph cannot take its seed by value (it would be frozen at capture): the
// forward's first kernel (seed_advance_kernel) copies the caller's 64-bit seed state into the workspace's seed block, derives the keys
// of every dropout stream the call uses into the same block and advances the state; every kernel of that forward AND of its backward
// then fetches its stream's keys from the block (one 8-byte scalar load).  Such a kernel receives a DropCfg whose s0 holds the SLOT of
// its stream in the block (s1 = 0) next to the block's pointer; the keys are those make_drop() gives on the host for (seed, stream), so
// mmt_debug_dropout_mask(seed value, stream) replays a device-seeded mask too.
//   block[0] = the seed of this forward;  block[1 + i] = (s0 | s1 << 32) of stream first_stream + i
#define MMT_SEED_BLOCK_WORDS 80                 // uint64 words reserved in a workspace: 1 + up to 4 * 16 + 4 encoder streams
__device__ __forceinline__ DropCfg drop_resolve(DropCfg c, const uint64_t* __restrict__ seedblock) {
    if (seedblock && c.thr16) {
        const uint64_t k = seedblock[1 + c.s0];
        c.s0 = (uint32_t)k;
        c.s1 = (uint32_t)(k >> 32);
    }
    return c;
}
// state[0]: the seed the NEXT train-mode forward uses.  seed block <- state and the stream keys, state <- splitmix64(state).
__global__ void seed_advance_kernel(uint64_t* __restrict__ state, uint64_t* __restrict__ seedblock, uint32_t first_stream, int nstreams) {
    const uint64_t s = state[0];
    const int i = threadIdx.x;
    if (i < nstreams) seedblock[1 + i] = splitmix64(splitmix64(s) + 0x100000001B3ull * (uint64_t)(first_stream + i));
    __syncthreads();                            // every thread has read the state
    if (i == 0) { seedblock[0] = s; state[0] = splitmix64(s); }
}
// `bits`: resolution of the drop probability, P(drop) = round(p * 2^bits) / 2^bits.  16 for the streams that compare a 16-bit hash half
// with the threshold; 12 for the attention-probability stream, whose bit-parallel generator (attn_mask.h) spends one hash word per
// threshold bit above the lowest set one: p = 0.1 -> 410/4096 = 0.100098 with 11 words per 32 decisions instead of 15 (thr16 =
// 0x199A).  The scale of the kept values is always that of the probability actually used, so the estimator stays unbiased.
__host__ __device__ inline DropCfg make_drop(float p, uint64_t seed, uint32_t stream, int bits = 16) {
    DropCfg c;
    const double q = (double)(1u << bits);
    double t = (double)p * q + 0.5;
    c.thr16 = p <= 0.f ? 0u : (t >= q - 1.0 ? (uint32_t)(q - 1.0) : (uint32_t)t) << (16 - bits);
    if (p > 0.f && c.thr16 == 0u) c.thr16 = 1u << (16 - bits);          // a positive p never rounds to "off"
    c.scale = 65536.0f / (65536.0f - (float)c.thr16);
    const uint64_t k = splitmix64(splitmix64(seed) + 0x100000001B3ull * (uint64_t)stream);
    c.s0 = (uint32_t)k;
    c.s1 = (uint32_t)(k >> 32);
    return c;
}
