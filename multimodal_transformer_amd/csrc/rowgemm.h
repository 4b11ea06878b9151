// Row-local GEMM for gfx950:  C[M][N] = epilogue( prologue(A)[M][K] . W[N][K]^T ).
//
// Every window-local step of the encoder (QKV projection, output projection, the two FFN
// products, their backward input-gradients, embeds, read-out MLPs) is an instance of this kernel.
// One workgroup (8 waves) owns MMT_ROWS (32) consecutive windows — see "Tile geometry" below:
//   1. A tile -> LDS as bf16 [32][KP+8] (optionally through the reference's LayerNorm, computed
//      in fp32 from an fp32 LDS staging copy; optionally emitting the tile as a row-major bf16
//      array, an operand of the weight-gradient kernel);
//   2. per 128-column chunk, wave w multiplies the tile by W rows [n0+16w, n0+16w+16) with
//      mfma_f32_16x16x32_bf16 (W fragments straight from L2 through a small prefetch ring: each is
//      used by exactly one wave), and parks its fp32 accumulators in an LDS tile;
//   3. a row-wise epilogue reads that LDS tile with a thread->(row, 4 columns) mapping, so all
//      global traffic is 8/16-byte coalesced whatever the MFMA accumulator layout was.
// Epilogues: PLAIN (bias/ReLU/ReLU-mask/residual/row-scale; fp32 and bf16 outputs),
//            FRAG  (attention operand fragment layout (R) for Q/K/V or dO, plus delta = rowsum(dO.O)),
//            LNBWD (LayerNorm backward fused behind the input-gradient GEMM + residual gradient).
// Stages chain inside one kernel (the tile stays in LDS between them): the forward chains, the backward chain and the
// backward boundary chain at the end of this file.
#pragma once
#include "common.h"

// (`//@phase` comment lines mark the phase boundaries of a stage: tools/build_phase.sh builds a diagnostic library from a PATCHED COPY of this
// file in which they are cycle stamps — tools/make_phase.py; this file holds no diagnostic code)
enum { EPI_PLAIN = 0, EPI_FRAG = 1, EPI_LNBWD = 2 };

// Tile geometry.  These kernels are bound by INSTRUCTION ISSUE, not by MFMA or memory: a wave issues at most one vector
// instruction every ~6 cycles and a SIMD only reaches its issue rate with four or more waves (tools/valu_micro.hip), while a
// stage is a few thousand instructions of staging / LayerNorm / epilogue arithmetic around ~100 cycles of MFMA.  Round 1 ran a
// 32-window tile on 4 waves at 250+ VGPRs: two workgroups per CU = two waves per SIMD, half the issue slots idle.  Measured
// alternatives at configs[3]: 16-window tiles x 4 waves (twice the workgroups, four per CU) are SLOWER (+9 % step): the per-wave
// instruction count barely drops, so the total rises; 32-window tiles x 8 waves keep the total and halve it per wave
// (<= 128 VGPRs, two workgroups per CU = four waves per SIMD).  MMT_ROWS / MMT_RTHREADS select the geometry at build time.
#ifndef MMT_ROWS
#define MMT_ROWS 32
#endif
#ifndef MMT_RTHREADS
#define MMT_RTHREADS 512
#endif
static_assert(MMT_ROWS == 16 || MMT_ROWS == 32, "row tile must be 16 or 32 windows");
static_assert(MMT_RTHREADS == 256 || MMT_RTHREADS == 512, "row kernels run 4 or 8 waves");
static_assert(MMT_RTHREADS / MMT_ROWS == 8 || MMT_RTHREADS / MMT_ROWS == 16, "8 or 16 threads per row");
#define MMT_RNW (MMT_RTHREADS / 64)                // waves per workgroup
#define MMT_RTPR (MMT_RTHREADS / MMT_ROWS)         // threads per row in the row-wise passes (LayerNorm, its backward): 16 or 8
#define MMT_RSTEP (MMT_RTHREADS / 32)              // row step of the (row, 4 columns) epilogue tasks: rows rbase + RSTEP*it
#define MMT_RIT (MMT_ROWS / MMT_RSTEP)             // such tasks per thread
#define MMT_WCOLS (128 / MMT_RNW)                  // columns of a 128-column chunk per wave: 32 (two MFMA tiles) or 16 (one)
#define MMT_WNT (MMT_WCOLS / 16)
struct RowGemmParams {
    int M, K, KP, N, NP;
    // ---- A operand ----
    const void* A; int a_bf16; int lda;
    bf16* A_out; int lda_out;              // optional row-major copy of the bf16 A tile [M][lda_out] (an operand of the weight-gradient kernel)
    DropCfg a_drop;                        // thr16 != 0: fp32 A is multiplied by the dropout mask of index m*KP + k on load
    // LayerNorm prologue (A must be fp32, K = feature count)
    const float* ln_a; const float* ln_b; float eps; float* stats;   // stats: [M][2] = (mean, 1/(std+eps))
    // ---- W operand: bf16 [NP][KP], zero padded; bias fp32 [NP] zero padded ----
    const bf16* W; const float* bias;
    // ---- PLAIN ----
    int act;                               // 1 = ReLU, 2 = tanh, 3 = sigmoid
    const bf16* relu_mask; int ldm; float mask_scale;   // v = relu_mask[m][n] > 0 ? v * mask_scale : 0
    DropCfg drop;                          // thr16 != 0: dropout after the activation, index m*NP + n
    const float* residual; int ldr;
    const float* rowscale;                 // multiply row m by rowscale[m]
    float* out_f32; int ldo;
    bf16* out_bf16; int ldo16; int n_store16;   // columns [0, n_store16) are written (pads come out as exact zeros)
    // ---- FRAG ----
    bf16* fragR[3];
    int T, Tp, h, DKP, nwhich;             // N covers nwhich * h * DKP columns
    const float* rowmask; float qscale; int scale_first;   // first matrix: *qscale and zero where rowmask==0
    const bf16* ctx; int ldctx; float* delta;              // delta[bh][Tp] = -sum_e C*ctx (dO epilogue; stored negated)
    // ---- LNBWD ----
    const float* x; int ldx; const float* st; const float* dres; int lddres;
    float* colpart;                        // [gridDim.x][2][NP]: per-workgroup column sums (d ln_b, d ln_a)
    int d_real;
    int no_gs;                             // column sums of dy * x-hat recompute x-hat from x instead of reading a second fp32 tile (Gs)
    DropCfg next_drop; int next_lda;       // KEEP_AS: the next stage's A tile = bf16(drop(out)) is left in As with this row stride
    // ---- K-chunked A staging (bf16 A from global): only `kchunk` (a power of two >= 64) columns of the A tile are in LDS at a time
    int kchunk;                            // 0: the whole K
    // ---- device-resident dropout seed (common.h drop_resolve): non-null = the DropCfg fields above carry stream slots, not keys
    const uint64_t* seedword;
};

__host__ __device__ inline int rowgemm_fw(int EPI, bool lnpro, int KP, int NP) {
    int cw = (EPI == EPI_LNBWD) ? NP : 128;
    int fw = cw;
    if (lnpro && KP > fw) fw = KP;
    return fw;
}
inline size_t rowgemm_lds_bytes(int EPI, bool lnpro, int KP, int NP, int kchunk = 0, bool no_gs = false) {
    size_t a = (size_t)MMT_ROWS * ((kchunk > 0 && kchunk < KP ? kchunk : KP) + 8) * 2;
    size_t f = (size_t)MMT_ROWS * (rowgemm_fw(EPI, lnpro, KP, NP) + 4) * 4;
    size_t g = (EPI == EPI_LNBWD && !no_gs) ? (size_t)MMT_ROWS * (NP + 4) * 4 : 0;
    return a + f + g;
}

// A stage = one row-local GEMM with its prologue and epilogue.  Stages can be chained inside one kernel: the
// 32-window tile then stays in LDS between them (Xs: fp32 tile of the residual stream / a gradient; A2: bf16 tile
// ready to be the next A operand) instead of making a round trip through L2 and a kernel boundary.
enum { ASRC_GLOBAL = 0, ASRC_X = 1, ASRC_A2 = 2, ASRC_AS = 3 };   // where the A operand comes from (AS: already in As, left by KEEP_AS)
enum { KEEP_X = 1, KEEP_A2 = 2, RES_X = 4, KEEP_AS = 8 };     // epilogue: also write fp32 to Xs / bf16 to A2 / bf16 to As; residual from Xs
struct RowSmem { bf16* As; float* Fs; float* Gs; float* Xs; bf16* A2; int ldf; int ldx; int lda2; };

// (sequence, window) of row `row` of a tile whose first row is window t0 of sequence b0.  An integer division by a run-time
// divisor is ~40 vector instructions; these kernels are bound by instruction issue and used to pay one per epilogue task.  A tile
// is MMT_ROWS consecutive rows: with T >= MMT_ROWS it crosses at most one sequence boundary.
struct SeqPos { int b, t; };
__device__ __forceinline__ SeqPos seq_pos(int b0, int t0, int row, int T) {
    SeqPos sp;
    int t = t0 + row;
    if (T >= MMT_ROWS) { const bool over = t >= T; sp.b = b0 + (over ? 1 : 0); sp.t = over ? t - T : t; }
    else { const int q = t / T; sp.b = b0 + q; sp.t = t - q * T; }
    return sp;
}
// which of the (up to three) stacked matrices column n belongs to, its head and feature: HD = h * DKP, DKP a power of two
struct ColPos { int wi, rem, head, e; };
__device__ __forceinline__ ColPos col_pos(int n, int HD, int DKP) {
    ColPos cp;
    cp.wi = (n >= HD ? 1 : 0) + (n >= 2 * HD ? 1 : 0);
    cp.rem = n - cp.wi * HD;
    const int sh = __builtin_ctz((unsigned)DKP);
    cp.head = cp.rem >> sh;
    cp.e = cp.rem & (DKP - 1);
    return cp;
}

// WIDE: the d_model > 128 variant of a LayerNorm-backward stage — K-chunked A staging (RowGemmParams::kchunk) and column sums without the
// second fp32 tile (RowGemmParams::no_gs, which the host sets to match).  Its own instantiation, so that the d_model <= 128 kernels
// carry neither the extra barriers nor the registers of the batched re-reads.
// W-fragment ring depth of a stage: PFD k-blocks of 64 per wave.  The stand-alone LayerNorm-backward instance (bwd_qkv: K = 3 h d_k, six
// k-blocks at d_model = 128, 58 VGPRs) keeps six blocks in flight — with two, four L2 round trips per tile were exposed; the chained
// kernels sit at the 128-VGPR cap and keep two.
template <int EPI, int ASRC, int KEEP, bool WIDE> __host__ __device__ constexpr int rowgemm_pfd() {
    return (EPI == EPI_LNBWD && ASRC == ASRC_GLOBAL && KEEP == 0) ? (WIDE ? 4 : 6) : 2;
}
// DEVSEED: the stage's DropCfgs carry stream slots and the keys come from the workspace's seed block (common.h drop_resolve: one scalar
// load where a mask is first needed).  Its own instantiation: the by-value kernels read the keys straight from their arguments and carry
// none of it (these kernels run out of scalar registers as it is).
template <int EPI, bool LNPRO, int ASRC, int KEEP, bool DEVSEED, bool WIDE = false>
__device__ __forceinline__ void rowgemm_stage(const RowGemmParams& p, const RowSmem& sm) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KP = p.KP, NP = p.NP, K = p.K, M = p.M;
    // K-chunked staging: kc columns of the A tile in LDS at a time (kc == KP: all of it, the usual case)
    const int kc = (WIDE && p.kchunk > 0 && p.kchunk < KP) ? p.kchunk : KP;
    const int lda_s = (ASRC == ASRC_A2) ? sm.lda2 : kc + 8;    // A tile row stride (bf16 elements)
    const int ldf = sm.ldf;
    bf16* As = (ASRC == ASRC_A2) ? sm.A2 : sm.As;
    float* Fs = sm.Fs;
    float* Gs = sm.Gs;                                         // LNBWD only
    float* Xs = sm.Xs;
    constexpr int ROWS = MMT_ROWS, TPR = MMT_RTPR, MT = MMT_ROWS / 16;
    const int m0 = blockIdx.x * ROWS;
    const int l15 = lane & 15, lq = lane >> 4;
    //@phase decl

    // W fragments (straight from L2, ~1k cycles away) travel through a ring of PFD k-blocks per wave.  The first PFD blocks of the
    // first chunk go in flight NOW: their latency overlaps the A-tile staging.  (Requesting the NEXT stage's first blocks behind a stage's
    // last k-loop, so that they travel during its epilogue, was measured: the forward chain got 5 us per step slower — the ring's 16
    // registers stay live across the epilogue of kernels that sit at the 128-VGPR cap.)
    constexpr int PFD = rowgemm_pfd<EPI, ASRC, KEEP, WIDE>();
    bf16x8 wf[PFD][2][MMT_WNT];                                   // [slot][k half: +0 / +32][16-column tile of the wave]
    auto w_load = [&](int slot, int nb, int kb) {                 // slot and the guard are compile-time / wave-uniform
        const bf16* wr = p.W + (size_t)(nb + l15) * KP + 8 * lq + kb;
#pragma unroll
        for (int b = 0; b < MMT_WNT; ++b) {
            wf[slot][0][b] = *reinterpret_cast<const bf16x8*>(wr + (size_t)16 * b * KP);
            wf[slot][1][b] = *reinterpret_cast<const bf16x8*>(wr + (size_t)16 * b * KP + 32);
        }
    };
    auto w_prime = [&](int nb) {
#pragma unroll
        for (int j = 0; j < PFD; ++j) if (64 * j < KP) w_load(j, nb, 64 * j);
    };
    if (wave * MMT_WCOLS < NP) w_prime(wave * MMT_WCOLS);

    // ------------------------------------------------------------------ 1. A tile -> LDS (bf16)
    auto stage_a16 = [&](int k0) {                              // columns [k0, k0 + kc) of a bf16 A matrix in global memory
        const bf16* A = static_cast<const bf16*>(p.A);
        for (int row = tid >> 3; row < ROWS; row += MMT_RTHREADS / 8)               // 8 lanes x 16 bytes along a row
        for (int c = (tid & 7) * 8; c < kc; c += 64) {
            const int m = m0 + row;
            bf16x8 v;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (bf16)0.f;
            if (m < M && k0 + c < K) v = *reinterpret_cast<const bf16x8*>(A + (size_t)m * p.lda + k0 + c);
            *reinterpret_cast<bf16x8*>(As + row * lda_s + c) = v;
        }
    };
    if (ASRC == ASRC_A2 || ASRC == ASRC_AS) {
        // the previous stage left the bf16 tile in A2 / As (and ended on a barrier)
    } else if (LNPRO) {
        if (ASRC == ASRC_GLOBAL) {
            const float* A = static_cast<const float*>(p.A);
            for (int row = tid >> 4; row < ROWS; row += MMT_RTHREADS / 16) {          // 16 lanes x 16 bytes along a row (KP % 64 == 0)
                const int m = m0 + row;
                for (int c = (tid & 15) * 4; c < KP; c += 64) {
                    f32x4 v = {0.f, 0.f, 0.f, 0.f};
                    if (m < M && c < K) v = *reinterpret_cast<const f32x4*>(A + (size_t)m * p.lda + c);
                    *reinterpret_cast<f32x4*>(Fs + row * ldf + c) = v;
                }
            }
            __syncthreads();
        }
        // TPR threads per row; LayerNorm of the reference: unbiased std, eps added to std
        const int row = tid / TPR, j = tid % TPR, m = m0 + row;
        const float* xr = (ASRC == ASRC_X) ? (Xs + row * sm.ldx) : (Fs + row * ldf);
        float s = 0.f;
        for (int c = j * 4; c < K; c += 4 * TPR) { f32x4 v = *reinterpret_cast<const f32x4*>(xr + c); s += (v[0] + v[1]) + (v[2] + v[3]); }
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        if (TPR == 16) s += __shfl_xor(s, 8);
        const float mean = s / (float)K;
        float q = 0.f;
        for (int c = j * 4; c < K; c += 4 * TPR) {
            f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) { float dlt = v[i] - mean; q += dlt * dlt; }
        }
        q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);
        if (TPR == 16) q += __shfl_xor(q, 8);
        const float sigma = sqrtf(q / (float)(K - 1));
        const float rstd = 1.0f / (sigma + p.eps);
        if (j == 0 && m < M && p.stats) { p.stats[2 * (size_t)m] = mean; p.stats[2 * (size_t)m + 1] = rstd; }
        for (int c = j * 4; c < KP; c += 4 * TPR) {
            bf16x4 o = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            if (c < K && m < M) {
                f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
                f32x4 a = *reinterpret_cast<const f32x4*>(p.ln_a + c);
                f32x4 b = *reinterpret_cast<const f32x4*>(p.ln_b + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (bf16)(a[i] * ((v[i] - mean) * rstd) + b[i]);
            }
            *reinterpret_cast<bf16x4*>(As + row * lda_s + c) = o;
        }
    } else if (ASRC == ASRC_X) {
        // fp32 tile kept by the previous stage -> bf16 (optionally through the dropout mask of index m*KP + k)
        const DropCfg a_drop = DEVSEED ? drop_resolve(p.a_drop, p.seedword) : p.a_drop;
        for (int row = tid >> 4; row < ROWS; row += MMT_RTHREADS / 16)
        for (int c = (tid & 15) * 4; c < KP; c += 64) {
            const int m = m0 + row;
            bf16x4 o = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            if (m < M && c < K) {
                f32x4 v = *reinterpret_cast<const f32x4*>(Xs + row * sm.ldx + c);
                if (a_drop.thr16) {
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {
                        const uint32_t w = drop_pair(a_drop, (uint64_t)m * KP + c + i);
                        v[i] = drop_lo(a_drop, w, v[i]); v[i + 1] = drop_hi(a_drop, w, v[i + 1]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (bf16)v[i];
            }
            *reinterpret_cast<bf16x4*>(As + row * lda_s + c) = o;
        }
    } else if (p.a_bf16) {
        stage_a16(0);
    } else {
        const float* A = static_cast<const float*>(p.A);
        const DropCfg a_drop = DEVSEED ? drop_resolve(p.a_drop, p.seedword) : p.a_drop;
        for (int row = tid >> 4; row < ROWS; row += MMT_RTHREADS / 16)
        for (int c = (tid & 15) * 4; c < KP; c += 64) {
            const int m = m0 + row;
            bf16x4 o = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
            if (m < M && c < K) {
                f32x4 v = *reinterpret_cast<const f32x4*>(A + (size_t)m * p.lda + c);
                if (a_drop.thr16) {
#pragma unroll
                    for (int i = 0; i < 4; i += 2) {
                        const uint32_t w = drop_pair(a_drop, (uint64_t)m * KP + c + i);
                        v[i] = drop_lo(a_drop, w, v[i]); v[i + 1] = drop_hi(a_drop, w, v[i + 1]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = (bf16)v[i];
            }
            *reinterpret_cast<bf16x4*>(As + row * lda_s + c) = o;
        }
    }
    __syncthreads();
    //@phase 0: A tile staged (LayerNorm prologue included)

    if (p.A_out) {    // row-major copy of the bf16 tile (16-byte pieces, whole rows contiguous)
        for (int row = tid >> 3; row < ROWS; row += MMT_RTHREADS / 8)
        for (int c = (tid & 7) * 8; c < KP; c += 64) {
            const int m = m0 + row;
            if (m < M) *reinterpret_cast<bf16x8*>(p.A_out + (size_t)m * p.lda_out + c) = *reinterpret_cast<const bf16x8*>(As + row * lda_s + c);
        }
    }

    //@phase 1: row-major copy of the A tile
    // ------------------------------------------------------------------ 2. chunks of 128 columns
    for (int n0 = 0; n0 < NP; n0 += 128) {
        const int nb = n0 + wave * MMT_WCOLS;
        const bool active = nb < NP;                       // wave-uniform; with K-chunked staging idle waves still take part in the barriers
        if (WIDE || active) {
            f32x4 acc[MT][MMT_WNT];
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < MMT_WNT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
            const bf16* arow0 = As + l15 * lda_s + 8 * lq;
            const bf16* arow1 = arow0 + 16 * lda_s;
            if (WIDE && kc < KP && n0 > 0) { __syncthreads(); stage_a16(0); __syncthreads(); }      // K-chunked: back to the first chunk
            for (int kb0 = 0; kb0 < KP; kb0 += 64 * PFD) {
#pragma unroll
                for (int j = 0; j < PFD; ++j) {
                    const int kb = kb0 + 64 * j;
                    if (kb >= KP) break;                          // wave-uniform
                    if (WIDE && kc < KP && kb > 0 && (kb & (kc - 1)) == 0) { __syncthreads(); stage_a16(kb); __syncthreads(); }
                    if (WIDE && !active) continue;
                    const int ka = (WIDE && kc < KP) ? (kb & (kc - 1)) : kb;
                    const bf16x8 a00 = *reinterpret_cast<const bf16x8*>(arow0 + ka);
                    const bf16x8 a10 = *reinterpret_cast<const bf16x8*>(arow0 + ka + 32);
#pragma unroll
                    for (int b = 0; b < MMT_WNT; ++b) acc[0][b] = mfma16(a00, wf[j][0][b], acc[0][b]);
                    if (MT == 2) {
                        const bf16x8 a01 = *reinterpret_cast<const bf16x8*>(arow1 + ka);
#pragma unroll
                        for (int b = 0; b < MMT_WNT; ++b) acc[MT - 1][b] = mfma16(a01, wf[j][0][b], acc[MT - 1][b]);
                    }
#pragma unroll
                    for (int b = 0; b < MMT_WNT; ++b) acc[0][b] = mfma16(a10, wf[j][1][b], acc[0][b]);
                    if (MT == 2) {
                        const bf16x8 a11 = *reinterpret_cast<const bf16x8*>(arow1 + ka + 32);
#pragma unroll
                        for (int b = 0; b < MMT_WNT; ++b) acc[MT - 1][b] = mfma16(a11, wf[j][1][b], acc[MT - 1][b]);
                    }
                    if (kb + 64 * PFD < KP) w_load(j, nb, kb + 64 * PFD);        // the slot just consumed takes the block PFD ahead
                }
            }
            if (active) {
                if (nb + 128 < NP) w_prime(nb + 128);             // next chunk's first blocks travel behind this chunk's epilogue
                const int cbase = ((EPI == EPI_LNBWD) ? nb : wave * MMT_WCOLS) + l15;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < MMT_WNT; ++nt)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            Fs[(mt * 16 + 4 * lq + r) * ldf + cbase + nt * 16] = acc[mt][nt][r];
            }
        }
        if (EPI == EPI_LNBWD) { /*@phase 2*/ continue; }
        __syncthreads();
        //@phase 2: k-loop + accumulators parked

        // -------------------------------------------------------------- 3. row-wise epilogue (chunk)
        if (EPI == EPI_PLAIN) {
            // thread -> column group cg (4 columns, the same for its 4 tasks) and rows (tid>>5) + 8*it.
            // Phase 1 issues every global load of the 4 tasks, phase 2 computes, phase 3 stores: stores through the
            // (non-restrict) output pointers would otherwise fence the later tasks' loads and serialise 4 L2 round trips.
            const int cg = tid & 31, n = n0 + cg * 4, rbase = tid >> 5;
            const bool col_ok = n < NP;
            const DropCfg e_drop = DEVSEED ? drop_resolve(p.drop, p.seedword) : p.drop;
            f32x4 v[MMT_RIT], res[MMT_RIT];
            bf16x4 mk[MMT_RIT];
            float rs[MMT_RIT];
            f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
            if (col_ok && p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
            const bool res_vec = p.residual && (n + 4 <= p.N) && ((p.ldr & 3) == 0);
#pragma unroll
            for (int it = 0; it < MMT_RIT; ++it) {
                const int row = rbase + MMT_RSTEP * it, m = m0 + row;
                res[it] = f32x4{0.f, 0.f, 0.f, 0.f};
                rs[it] = 1.f;
                if (col_ok && m < M) {
                    if (KEEP & RES_X) {
                        res[it] = *reinterpret_cast<const f32x4*>(Xs + row * sm.ldx + n);
                    } else if (p.residual) {
                        if (res_vec) res[it] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.ldr + n);
                        else for (int i = 0; i < 4; ++i) if (n + i < p.N) res[it][i] = p.residual[(size_t)m * p.ldr + n + i];
                    }
                    if (p.relu_mask) mk[it] = *reinterpret_cast<const bf16x4*>(p.relu_mask + (size_t)m * p.ldm + n);
                    if (p.rowscale) rs[it] = p.rowscale[m];
                }
                v[it] = *reinterpret_cast<const f32x4*>(Fs + row * ldf + cg * 4);
            }
#pragma unroll
            for (int it = 0; it < MMT_RIT; ++it) {
                const int row = rbase + MMT_RSTEP * it, m = m0 + row;
                if (!col_ok) continue;
                f32x4 x = v[it];
                if (m < M) {
                    x += bias4;
                    if (p.act == 1) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[i] = fmaxf(x[i], 0.f);
                    } else if (p.act == 2) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[i] = 2.0f * __builtin_amdgcn_rcpf(1.0f + fast_exp2(-2.8853900817779268f * x[i])) - 1.0f;
                    } else if (p.act == 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[i] = __builtin_amdgcn_rcpf(1.0f + fast_exp2(-1.4426950408889634f * x[i]));
                    }
                    if (p.relu_mask) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) x[i] = ((float)mk[it][i] > 0.f) ? x[i] * p.mask_scale : 0.f;
                    }
                    if (e_drop.thr16) {
#pragma unroll
                        for (int i = 0; i < 4; i += 2) {
                            const uint32_t w = drop_pair(e_drop, (uint64_t)m * NP + n + i);
                            x[i] = drop_lo(e_drop, w, x[i]); x[i + 1] = drop_hi(e_drop, w, x[i + 1]);
                        }
                    }
                    x += res[it];
                    x *= rs[it];
                } else {
                    x = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                v[it] = x;
            }
#pragma unroll
            for (int it = 0; it < MMT_RIT; ++it) {
                const int row = rbase + MMT_RSTEP * it, m = m0 + row;
                if (!col_ok) continue;
                if (m < M) {
                    if (p.out_f32) {
                        float* dst = p.out_f32 + (size_t)m * p.ldo + n;
                        if (n + 4 <= p.N && (p.ldo & 3) == 0) *reinterpret_cast<f32x4*>(dst) = v[it];
                        else for (int i = 0; i < 4 && n + i < p.N; ++i) dst[i] = v[it][i];
                    }
                    if (p.out_bf16 && n < p.n_store16) {
                        bf16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) o[i] = (bf16)v[it][i];
                        bf16* dst = p.out_bf16 + (size_t)m * p.ldo16 + n;
                        if (n + 4 <= p.n_store16 && (p.ldo16 & 3) == 0) *reinterpret_cast<bf16x4*>(dst) = o;
                        else for (int i = 0; i < 4 && n + i < p.n_store16; ++i) dst[i] = o[i];
                    }
                }
                if (KEEP & KEEP_X) *reinterpret_cast<f32x4*>(Xs + row * sm.ldx + n) = v[it];
                if (KEEP & KEEP_A2) {
                    bf16x4 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = (bf16)v[it][i];
                    *reinterpret_cast<bf16x4*>(sm.A2 + row * sm.lda2 + n) = o;
                }
            }
        } else if (EPI == EPI_FRAG) {
            // Output: the R fragment layout [batch*head][tile][e>>3][t&31][e&7] — the only layout the attention kernels read.
            // The store pass lays the lanes along the tile's windows: 16 bytes per lane, 512-byte runs per 8-feature group
            // (8-byte stores with the lanes across columns cost one write transaction per lane).
            const int HD = p.h * p.DKP;
            const size_t szR = fragR_elems(p.Tp, p.DKP);
            // first row of the tile in (sequence, window) form: one wave-uniform division per stage
            const int tb0 = __builtin_amdgcn_readfirstlane(m0 / p.T), tt0 = m0 - tb0 * p.T;
            if (p.delta) {
                // dO epilogue: delta = rowsum(dO . O) per (window, head) needs the row-wise thread mapping first: round to bf16, take the
                // partial sums over the head's lanes, and leave the rounded values in the tile for the store pass
                const int nred = p.DKP >> 2;             // lanes per head (4, 8 or 16), aligned groups
                const int cg = tid & 31, n = n0 + cg * 4, rbase = tid >> 5;
                const bool col_ok = n < p.nwhich * HD;
                const ColPos cp = col_pos(col_ok ? n : 0, HD, p.DKP);
                f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
                if (col_ok && p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
                bf16x4 c4[MMT_RIT];
#pragma unroll
                for (int it = 0; it < MMT_RIT; ++it) {
                    const int m = m0 + rbase + MMT_RSTEP * it;
                    if (col_ok && m < M) c4[it] = *reinterpret_cast<const bf16x4*>(p.ctx + (size_t)m * p.ldctx + cp.rem);
                }
#pragma unroll
                for (int it = 0; it < MMT_RIT; ++it) {
                    const int row = rbase + MMT_RSTEP * it, m = m0 + row;
                    const bool ok = col_ok && (m < M);
                    float part = 0.f;
                    f32x4 w = {0.f, 0.f, 0.f, 0.f};
                    if (ok) {
                        f32x4 v = *reinterpret_cast<const f32x4*>(Fs + row * ldf + cg * 4);
                        v += bias4;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { const bf16 o = (bf16)v[i]; w[i] = (float)o; part += w[i] * (float)c4[it][i]; }
                    }
                    *reinterpret_cast<f32x4*>(Fs + row * ldf + cg * 4) = w;
                    part += __shfl_xor(part, 1); part += __shfl_xor(part, 2);
                    if (nred >= 8) part += __shfl_xor(part, 4);
                    if (nred == 16) part += __shfl_xor(part, 8);
                    if (ok && cp.e == 0) {
                        const SeqPos sp = seq_pos(tb0, tt0, row, p.T);
                        p.delta[(size_t)(sp.b * p.h + cp.head) * p.Tp + sp.t] = -part;      // stored negated (accumulator init of the backward)
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int it = 0; it < (MMT_ROWS * 16 + MMT_RTHREADS - 1) / MMT_RTHREADS; ++it) {
                const int q = tid + it * MMT_RTHREADS, rr = q % MMT_ROWS, cc = q / MMT_ROWS, nn = n0 + 8 * cc, mm = m0 + rr;
                if (cc < 16 && nn < p.nwhich * HD && mm < M) {
                    const ColPos cj = col_pos(nn, HD, p.DKP);
                    const SeqPos sp = seq_pos(tb0, tt0, rr, p.T);
                    f32x4 lo = *reinterpret_cast<const f32x4*>(Fs + rr * ldf + 8 * cc);
                    f32x4 hi = *reinterpret_cast<const f32x4*>(Fs + rr * ldf + 8 * cc + 4);
                    if (!p.delta) {                      // QKV epilogue: bias, and for Q the softmax scale and the query-row mask
                        if (p.bias) { lo += *reinterpret_cast<const f32x4*>(p.bias + nn); hi += *reinterpret_cast<const f32x4*>(p.bias + nn + 4); }
                        if (p.scale_first && cj.wi == 0) {
                            const float sc = (p.rowmask[mm] == 0.0f) ? 0.f : p.qscale;       // mask == 0 -> blank query row
                            lo *= sc; hi *= sc;
                        }
                    }
                    bf16x8 o;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { o[i] = (bf16)lo[i]; o[4 + i] = (bf16)hi[i]; }
                    bf16* const fr = cj.wi == 0 ? p.fragR[0] : (cj.wi == 1 ? p.fragR[1] : p.fragR[2]);      // (selects, not a dynamic index: a local copy of the parameters stays in registers)
                    *reinterpret_cast<bf16x8*>(fr + (size_t)(sp.b * p.h + cj.head) * szR + fragR_index(sp.t, cj.e, p.DKP)) = o;
                }
            }
        }
        __syncthreads();
        //@phase 3: chunk epilogue (PLAIN / FRAG)
    }

    // ------------------------------------------------------------------ LayerNorm backward epilogue
    // Its operands (x, LayerNorm gain, residual gradient) are loaded where they are used, one dependent round trip after the other.
    // Fetching them early (inside the epilogue in round 1, before the k-loop in round 2: -1.5 % of the configs[3] step) is the
    // variant that was NOT bit-reproducible while a second process shared the GPU (reproducibility); cause unknown, so
    // the plain form stays.
    if (EPI == EPI_LNBWD) {
        __syncthreads();
        const int d = p.d_real;
        const int row = tid / TPR, j = tid % TPR, m = m0 + row;
        float* cr = Fs + row * ldf;          // dxn = grad wrt LayerNorm output (fp32)
        float* gr = Gs + row * (NP + 4);
        float mean = 0.f, rstd = 0.f;
        if (m < M) { mean = p.st[2 * (size_t)m]; rstd = p.st[2 * (size_t)m + 1]; }
        if (WIDE && j == 0) { cr[NP] = mean; cr[NP + 1] = rstd; }        // pad columns of the tile row (ldf = NP + 4): for the column pass
        constexpr int LNB_ITS = WIDE ? 4 : 2;         // 4-column pieces per thread that the one-pass form keeps in registers
        if (NP <= 4 * TPR * LNB_ITS) {
        // Row tile of at most two (d_model <= 128) or four (WIDE: <= 256; 16 threads per row) 4-column pieces per thread: ONE pass over memory.  The first
        // pass's operands — x-hat and dy a, and the residual gradient, fetched beside them — stay in registers across the two row
        // reductions, so the second pass reads nothing (round 4 read x, a and dy a second time: two dependent round trips to L2 per
        // stage, the largest phase of the backward chains: DESIGN.md 4.1b).  Same expressions in the same order: bit-identical results.
        float s1 = 0.f, s2 = 0.f;
        f32x4 gk[LNB_ITS], xk[LNB_ITS], rk[WIDE ? 1 : LNB_ITS];     // (WIDE fetches the residual gradient in the second pass: registers)
#pragma unroll
        for (int it = 0; it < LNB_ITS; ++it) {
            const int c = j * 4 + it * 4 * TPR;
            gk[it] = f32x4{0.f, 0.f, 0.f, 0.f}; xk[it] = gk[it]; if (!WIDE) rk[it] = gk[it];
            if (c >= NP) continue;
            f32x4 g = {0.f, 0.f, 0.f, 0.f}, xh = {0.f, 0.f, 0.f, 0.f};
            if (m < M && c < d) {
                f32x4 dy = *reinterpret_cast<const f32x4*>(cr + c);
                f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.ldx + c);
                f32x4 a = *reinterpret_cast<const f32x4*>(p.ln_a + c);
                if (!WIDE && p.dres) rk[it] = *reinterpret_cast<const f32x4*>(p.dres + (size_t)m * p.lddres + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xh[i] = (xv[i] - mean) * rstd;
                    g[i] = dy[i] * a[i];
                    s1 += g[i];
                    s2 += g[i] * xh[i];
                }
                if (!WIDE) {
                    f32x4 gx;
#pragma unroll
                    for (int i = 0; i < 4; ++i) gx[i] = dy[i] * xh[i];
                    *reinterpret_cast<f32x4*>(gr + c) = gx;
                }
                gk[it] = g; xk[it] = xh;
            } else {
                *reinterpret_cast<f32x4*>(cr + c) = g;
                if (!WIDE) *reinterpret_cast<f32x4*>(gr + c) = g;
            }
        }
        s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
        s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
        if (TPR == 16) { s1 += __shfl_xor(s1, 8); s2 += __shfl_xor(s2, 8); }
        {
            const bool live = m < M;
            const DropCfg next_drop = DEVSEED ? drop_resolve(p.next_drop, p.seedword) : p.next_drop;
            const float sigma = live ? 1.0f / rstd - p.eps : 1.f;
            const float k1 = s1 / (float)d, k2 = s2 / ((float)(d - 1) * sigma);
#pragma unroll
            for (int it = 0; it < LNB_ITS; ++it) {
                const int c = j * 4 + it * 4 * TPR;
                if (c >= ((KEEP & KEEP_AS) ? NP : d)) continue;
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (live && c < d) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) o[i] = rstd * (gk[it][i] - k1) - k2 * xk[it][i];
                    if (p.dres) { if (WIDE) o += *reinterpret_cast<const f32x4*>(p.dres + (size_t)m * p.lddres + c); else o += rk[it]; }
                    *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)m * p.ldo + c) = o;
                    if (KEEP & KEEP_X) *reinterpret_cast<f32x4*>(Xs + row * sm.ldx + c) = o;
                }
                if (KEEP & KEEP_AS) {        // the next stage's A tile: bf16 of the (dropped) gradient, zero in the pads and in rows >= M
                    if (next_drop.thr16 && live && c < d) {
#pragma unroll
                        for (int i = 0; i < 4; i += 2) {
                            const uint32_t w = drop_pair(next_drop, (uint64_t)m * NP + c + i);
                            o[i] = drop_lo(next_drop, w, o[i]); o[i + 1] = drop_hi(next_drop, w, o[i + 1]);
                        }
                    }
                    bf16x4 o16;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o16[i] = (bf16)o[i];
                    *reinterpret_cast<bf16x4*>(sm.As + row * p.next_lda + c) = o16;
                }
            }
        }
        } else
        {
        float s1 = 0.f, s2 = 0.f;
        for (int c = j * 4; c < NP; c += 4 * TPR) {
            f32x4 g = {0.f, 0.f, 0.f, 0.f}, xh = {0.f, 0.f, 0.f, 0.f};
            if (m < M && c < d) {
                f32x4 dy = *reinterpret_cast<const f32x4*>(cr + c);
                f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.ldx + c);
                f32x4 a = *reinterpret_cast<const f32x4*>(p.ln_a + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xh[i] = (xv[i] - mean) * rstd;
                    g[i] = dy[i] * a[i];
                    s1 += g[i];
                    s2 += g[i] * xh[i];
                }
                if (!WIDE) {
                    f32x4 gx;
#pragma unroll
                    for (int i = 0; i < 4; ++i) gx[i] = dy[i] * xh[i];
                    *reinterpret_cast<f32x4*>(gr + c) = gx;
                }
            } else {
                *reinterpret_cast<f32x4*>(cr + c) = g;
                if (!WIDE) *reinterpret_cast<f32x4*>(gr + c) = g;
            }
        }
        s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
        s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
        if (TPR == 16) { s1 += __shfl_xor(s1, 8); s2 += __shfl_xor(s2, 8); }
        {
            const bool live = m < M;
            const DropCfg next_drop = DEVSEED ? drop_resolve(p.next_drop, p.seedword) : p.next_drop;
            const float sigma = live ? 1.0f / rstd - p.eps : 1.f;
            const float k1 = s1 / (float)d, k2 = s2 / ((float)(d - 1) * sigma);
            for (int c = j * 4; c < ((KEEP & KEEP_AS) ? NP : d); c += 4 * TPR) {
                f32x4 o = {0.f, 0.f, 0.f, 0.f};
                if (live && c < d) {
                    f32x4 dy = *reinterpret_cast<const f32x4*>(cr + c);
                    f32x4 xv = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.ldx + c);
                    f32x4 a = *reinterpret_cast<const f32x4*>(p.ln_a + c);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float xh = (xv[i] - mean) * rstd;
                        o[i] = rstd * (dy[i] * a[i] - k1) - k2 * xh;
                    }
                    if (p.dres) { f32x4 r = *reinterpret_cast<const f32x4*>(p.dres + (size_t)m * p.lddres + c); o += r; }
                    *reinterpret_cast<f32x4*>(p.out_f32 + (size_t)m * p.ldo + c) = o;
                    if (KEEP & KEEP_X) *reinterpret_cast<f32x4*>(Xs + row * sm.ldx + c) = o;
                }
                if (KEEP & KEEP_AS) {        // the next stage's A tile: bf16 of the (dropped) gradient, zero in the pads and in rows >= M
                    if (next_drop.thr16 && live && c < d) {
#pragma unroll
                        for (int i = 0; i < 4; i += 2) {
                            const uint32_t w = drop_pair(next_drop, (uint64_t)m * NP + c + i);
                            o[i] = drop_lo(next_drop, w, o[i]); o[i + 1] = drop_hi(next_drop, w, o[i + 1]);
                        }
                    }
                    bf16x4 o16;
#pragma unroll
                    for (int i = 0; i < 4; ++i) o16[i] = (bf16)o[i];
                    *reinterpret_cast<bf16x4*>(sm.As + row * p.next_lda + c) = o16;
                }
            }
        }
        }
        __syncthreads();
        if (p.colpart && !WIDE && MMT_ROWS == 32 && MMT_RTHREADS == 512 && NP == 128) {
            // column sums of the 32-row tile on all 512 threads: wave w owns columns 16 w .. + 15, the lane's quarter rq the rows
            // 4 rq + i and 16 + 4 rq + i (i < 4): with 132-float rows the two quarters of a 32-lane half sit 16 banks apart, so every
            // read is conflict-free; the four quarters meet through two lane exchanges, in a fixed order
            const int c = 16 * (tid >> 6) + (tid & 15), rq = (tid >> 4) & 3;
            float sb = 0.f, sa = 0.f;
#pragma unroll
            for (int hh2 = 0; hh2 < 2; ++hh2)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = 16 * hh2 + 4 * rq + i;
                    sb += Fs[r * ldf + c]; sa += Gs[r * (NP + 4) + c];
                }
            sb += __shfl_xor(sb, 16); sa += __shfl_xor(sa, 16);
            sb += __shfl_xor(sb, 32); sa += __shfl_xor(sa, 32);
            if (rq == 0) {
                float* dst = p.colpart + (size_t)blockIdx.x * 2 * NP;
                dst[c] = sb; dst[NP + c] = sa;
            }
        } else if (p.colpart) {
            for (int c = tid; c < NP; c += MMT_RTHREADS) {
                float sb = 0.f, sa = 0.f;
                if (!WIDE) {
#pragma unroll 8
                    for (int r = 0; r < MMT_ROWS; ++r) { sb += Fs[r * ldf + c]; sa += Gs[r * (NP + 4) + c]; }
                } else {
                    // no second fp32 tile (it halves the workgroups per CU at d_model = 256): x-hat again from x (an L2 hit, just read
                    // by the row pass) with the same expression, so the sums are bit-identical to the stored form.  Four rows' loads
                    // in flight at a time; the rows' (mean, 1/std) were parked in the tile's pad columns by the row pass.
                    // (rows >= M: dy = 0 and the parked statistics are 0, so any finite x will do: the last valid row's)
                    const float* xc = p.x + (c < d ? c : 0);
                    for (int r0 = 0; r0 < MMT_ROWS; r0 += 4) {
                        float xv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) { const int mr = m0 + r0 + i; xv[i] = xc[(size_t)(mr < M ? mr : M - 1) * p.ldx]; }
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float* fr = Fs + (r0 + i) * ldf;
                            const float dyv = fr[c];
                            sb += dyv;
                            sa += dyv * ((xv[i] - fr[NP]) * fr[NP + 1]);
                        }
                    }
                }
                float* dst = p.colpart + (size_t)blockIdx.x * 2 * NP;
                dst[c] = sb; dst[NP + c] = sa;
            }
        }
        __syncthreads();
        //@phase 4: LayerNorm-backward epilogue + column partials
    }
    //@phase flush
}

// (kernel definitions that use the chained stages follow the stage template below)
// Touch every 128-byte line of a later stage's weight matrix at kernel entry (one dword per line, a quarter of the workgroups):
// the lines come up from MALL into this XCD's L2 while the first stage stages its A tile, so the later stages' W-fragment
// loads are L2 hits instead of first-touch misses.
__device__ __forceinline__ void warm_weights(const bf16* W, int NP, int KP) {
    if ((blockIdx.x & (MMT_ROWS == 16 ? 7 : 3)) != 0) return;
    const int lines = (NP * KP) >> 6;                           // 64 bf16 per 128-byte line
    unsigned acc = 0;
    for (int l = threadIdx.x; l < lines; l += MMT_RTHREADS) acc |= *reinterpret_cast<const unsigned*>(W + (size_t)l * 64);
    asm volatile("" :: "v"(acc));
}

// ---- single stage ------------------------------------------------------------------------------
template <int EPI, bool LNPRO, bool WIDE = false, bool DEVSEED = false>
__global__ __launch_bounds__(MMT_RTHREADS, (MMT_ROWS == 16 || MMT_RTHREADS == 512) ? 4 : 2) void rowgemm_kernel(const RowGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RowSmem sm;
    sm.As = reinterpret_cast<bf16*>(smem);
    sm.Fs = reinterpret_cast<float*>(smem + (size_t)MMT_ROWS * ((p.kchunk > 0 && p.kchunk < p.KP ? p.kchunk : p.KP) + 8) * 2);
    sm.ldf = rowgemm_fw(EPI, LNPRO, p.KP, p.NP) + 4;
    sm.Gs = sm.Fs + (size_t)MMT_ROWS * sm.ldf;
    sm.Xs = nullptr; sm.A2 = nullptr; sm.ldx = 0; sm.lda2 = 0;
    rowgemm_stage<EPI, LNPRO, ASRC_GLOBAL, 0, DEVSEED, WIDE>(p, sm);
}

// ---- chained stages ------------------------------------------------------------------------------
// final LayerNorm of the stack applied to the last stage's output tile while it is still in LDS (y == nullptr: none)
struct LnOut { const float* a; const float* b; float eps; float* y; float* stats; int d; };
struct RowChain3 { RowGemmParams a, b, c; int lda_max, ldf, ldx, lda2; LnOut ln; };   // LDS geometry decided by the host
struct RowChain4 { RowGemmParams a, b, c, d; int lda_max, ldf, ldx, lda2; };

template <typename CH>
__host__ __device__ inline size_t rowchain_lds_bytes(const CH& ch, bool with_g) {
    return (size_t)MMT_ROWS * ch.lda_max * 2 + (size_t)MMT_ROWS * ch.ldf * 4 * (with_g ? 2 : 1) + (size_t)MMT_ROWS * ch.ldx * 4 + (size_t)MMT_ROWS * ch.lda2 * 2;
}
template <typename CH>
__device__ __forceinline__ RowSmem rowchain_carve(char* smem, const CH& ch, bool with_g) {
    RowSmem sm;
    sm.As = reinterpret_cast<bf16*>(smem);
    sm.Fs = reinterpret_cast<float*>(smem + (size_t)MMT_ROWS * ch.lda_max * 2);
    sm.ldf = ch.ldf;
    sm.Gs = sm.Fs + (size_t)MMT_ROWS * ch.ldf;
    sm.Xs = sm.Fs + (size_t)MMT_ROWS * ch.ldf * (with_g ? 2 : 1);
    sm.ldx = ch.ldx;
    sm.A2 = reinterpret_cast<bf16*>(sm.Xs + (size_t)MMT_ROWS * ch.ldx);
    sm.lda2 = ch.lda2;
    return sm;
}

// y = LayerNorm(tile rows) -> global, with the reference's variant (unbiased std, eps added to std) and the (mean, 1/(std+eps)) pairs
// its backward needs; the tile is the fp32 output of the stage that just ended on a barrier.  MMT_RTPR threads per row.
__device__ __forceinline__ void ln_tile_out(const float* Xs, int ldx, const LnOut& lo, int M) {
    constexpr int TPR = MMT_RTPR;
    const int row = threadIdx.x / TPR, j = threadIdx.x % TPR, m = blockIdx.x * MMT_ROWS + row, K = lo.d;
    const float* xr = Xs + row * ldx;
    float s = 0.f;
    for (int c = j * 4; c < K; c += 4 * TPR) { f32x4 v = *reinterpret_cast<const f32x4*>(xr + c); s += (v[0] + v[1]) + (v[2] + v[3]); }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    if (TPR == 16) s += __shfl_xor(s, 8);
    const float mean = s / (float)K;
    float q = 0.f;
    for (int c = j * 4; c < K; c += 4 * TPR) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float t = v[i] - mean; q += t * t; }
    }
    q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);
    if (TPR == 16) q += __shfl_xor(q, 8);
    const float rstd = 1.0f / (sqrtf(q / (float)(K - 1)) + lo.eps);
    if (m >= M) return;
    if (j == 0 && lo.stats) { lo.stats[2 * (size_t)m] = mean; lo.stats[2 * (size_t)m + 1] = rstd; }
    for (int c = j * 4; c < K; c += 4 * TPR) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        const f32x4 av = *reinterpret_cast<const f32x4*>(lo.a + c), bv = *reinterpret_cast<const f32x4*>(lo.b + c);
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = av[i] * ((v[i] - mean) * rstd) + bv[i];
        *reinterpret_cast<f32x4*>(lo.y + (size_t)m * K + c) = o;
    }
}

// Stage roles of the chained kernels.  rowgemm_stage() serves every affine map of the library and decides most of its options at run
// time; in a chain each stage's role is fixed, and what mmt_encoder_forward / _backward (api.hip) leave unset for that role is pinned
// here in a local copy of the stage's parameters: the copies are constants to the compiler, so the options' code — the tanh / sigmoid
// epilogues, the ReLU-mask, row-scale and second-output paths, K-chunked staging — and their scalar branch / exec-mask bookkeeping
// (40 % of the static instruction stream of these kernels was scalar) drop out of the kernel.
__device__ __forceinline__ RowGemmParams role_plain(RowGemmParams p) {          // bias + dropout (+ residual), fp32 out: out-proj, FFN2
    p.act = 0; p.relu_mask = nullptr; p.rowscale = nullptr; p.out_bf16 = nullptr; p.A_out = nullptr; p.a_drop.thr16 = 0;
    p.next_drop.thr16 = 0; p.kchunk = 0; p.ln_a = nullptr; p.stats = nullptr;
    return p;
}
__device__ __forceinline__ RowGemmParams role_ffn1(RowGemmParams p) {           // LayerNorm + bias + ReLU + dropout, bf16 out and copy of the A tile
    p.act = 1; p.relu_mask = nullptr; p.rowscale = nullptr; p.out_f32 = nullptr; p.residual = nullptr; p.a_drop.thr16 = 0;
    p.next_drop.thr16 = 0; p.kchunk = 0;
    return p;
}
__device__ __forceinline__ RowGemmParams role_qkv(RowGemmParams p) {            // LayerNorm + Q/K/V projection -> attention operand fragments
    p.delta = nullptr; p.ctx = nullptr; p.scale_first = 1; p.nwhich = 3; p.a_drop.thr16 = 0; p.drop.thr16 = 0; p.next_drop.thr16 = 0; p.kchunk = 0;
    return p;
}
__device__ __forceinline__ RowGemmParams role_dO(RowGemmParams p) {             // dO = drop'(dx1) Wo -> fragments + delta, A tile already in LDS
    p.bias = nullptr; p.scale_first = 0; p.rowmask = nullptr; p.nwhich = 1; p.a_drop.thr16 = 0; p.drop.thr16 = 0; p.next_drop.thr16 = 0; p.kchunk = 0;
    p.ln_a = nullptr; p.stats = nullptr;
    return p;
}
__device__ __forceinline__ RowGemmParams role_lnbwd(RowGemmParams p) {          // input gradient + LayerNorm backward + residual gradient (bf16 A when it comes from memory)
    p.a_bf16 = 1; p.A_out = nullptr; p.a_drop.thr16 = 0; p.drop.thr16 = 0; p.bias = nullptr; p.ln_b = nullptr; p.stats = nullptr;
    return p;
}
__device__ __forceinline__ RowGemmParams role_bwd_relu(RowGemmParams p) {       // dh = (drop'(g) W) * relu'(hid): bf16 out, copy of the A tile
    p.act = 0; p.bias = nullptr; p.rowscale = nullptr; p.out_f32 = nullptr; p.residual = nullptr; p.drop.thr16 = 0;
    p.next_drop.thr16 = 0; p.kchunk = 0; p.ln_a = nullptr; p.stats = nullptr;
    return p;
}

// Forward, after the attention core of a layer:   x1 = x + drop(ctx Wo^T + bo)        (out-proj + residual; x1 kept in LDS)
//                                                 hid = drop(relu(LN2(x1) W1^T + b1))  (hid kept in LDS as the next A tile)
//                                                 x2 = x1 + drop(hid W2^T + b2)        (residual read from LDS)
template <int K_, int N_> __device__ __forceinline__ RowGemmParams shape_pin(RowGemmParams p);
template <int LDA_MAX, int LDF, bool WITH_G, int LDX, int LDA2> __device__ __forceinline__ RowSmem carve_fixed(char* smem);
template <bool DEVSEED, int SHAPE = 0>
__global__ __launch_bounds__(MMT_RTHREADS, (MMT_ROWS == 16 || MMT_RTHREADS == 512) ? 4 : 2) void encoder_post_attn_fwd_kernel(const RowChain3 ch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    warm_weights(ch.b.W, ch.b.NP, ch.b.KP); warm_weights(ch.c.W, ch.c.NP, ch.c.KP);
    if (SHAPE == 128) {                                         // (fixed-shape instances: see below)
        const RowSmem sm = carve_fixed<136, 132, false, 132, 136>(smem);
        rowgemm_stage<EPI_PLAIN, false, ASRC_GLOBAL, KEEP_X, DEVSEED>(shape_pin<128, 128>(role_plain(ch.a)), sm);
        rowgemm_stage<EPI_PLAIN, true, ASRC_X, KEEP_A2, DEVSEED>(shape_pin<128, 128>(role_ffn1(ch.b)), sm);
        rowgemm_stage<EPI_PLAIN, false, ASRC_A2, RES_X | KEEP_X, DEVSEED>(shape_pin<128, 128>(role_plain(ch.c)), sm);
        if (ch.ln.y) ln_tile_out(sm.Xs, sm.ldx, ch.ln, ch.c.M);
        return;
    }
    if (SHAPE == 256) {
        const RowSmem sm = carve_fixed<264, 132, false, 260, 136>(smem);
        rowgemm_stage<EPI_PLAIN, false, ASRC_GLOBAL, KEEP_X, DEVSEED>(shape_pin<256, 256>(role_plain(ch.a)), sm);
        rowgemm_stage<EPI_PLAIN, true, ASRC_X, KEEP_A2, DEVSEED>(shape_pin<256, 128>(role_ffn1(ch.b)), sm);
        rowgemm_stage<EPI_PLAIN, false, ASRC_A2, RES_X | KEEP_X, DEVSEED>(shape_pin<128, 256>(role_plain(ch.c)), sm);
        if (ch.ln.y) ln_tile_out(sm.Xs, sm.ldx, ch.ln, ch.c.M);
        return;
    }
    const RowSmem sm = rowchain_carve(smem, ch, false);
    rowgemm_stage<EPI_PLAIN, false, ASRC_GLOBAL, KEEP_X, DEVSEED>(role_plain(ch.a), sm);
    rowgemm_stage<EPI_PLAIN, true, ASRC_X, KEEP_A2, DEVSEED>(role_ffn1(ch.b), sm);
    rowgemm_stage<EPI_PLAIN, false, ASRC_A2, RES_X | KEEP_X, DEVSEED>(role_plain(ch.c), sm);
    if (ch.ln.y) ln_tile_out(sm.Xs, sm.ldx, ch.ln, ch.c.M);          // last layer: the stack's final LayerNorm, from the tile in LDS
}

// The same chain followed by the NEXT layer's LayerNorm-1 + Q/K/V projection (its input x2 is already in LDS): every
// layer but the last.  One launch and one round trip of the residual stream less per layer.
// ---- fixed-shape instances.  The chains above take every width, stride and the LDS geometry at run time; 70 % of their static
// instruction stream was integer / scalar bookkeeping (address arithmetic, bounds compares, branches) around 12 % floating point.
// For the widths of the BASELINE configs the host selects an instance in which the stages' shapes, leading dimensions and the LDS
// carve are pinned like the roles (shape_pin, carve_fixed): the arithmetic folds into immediates (configs[3]: 2 879 -> 1 989
// static instructions in the 4-stage forward chain, -11 % time).  SHAPE: 0 = generic; 128 = d_model = d_ff = h d_k = 128, 8 heads of 16.
template <int K_, int N_> __device__ __forceinline__ RowGemmParams shape_pin(RowGemmParams p) {
    p.K = K_; p.KP = K_; p.N = N_; p.NP = N_;
    p.lda = K_; p.lda_out = K_;                                 // A row-major [M][K] (and its bf16 copy)
    p.ldr = N_; p.ldo = N_; p.ldo16 = N_; p.n_store16 = N_; p.ldm = N_;      // outputs, residual, ReLU mask: [M][N]
    p.ldx = N_; p.lddres = N_; p.d_real = N_; p.next_lda = N_ + 8; p.ldctx = N_;   // LayerNorm backward (N = d_model), next A tile, dO's ctx
    return p;
}
template <int LDA_MAX, int LDF, bool WITH_G, int LDX, int LDA2>
__device__ __forceinline__ RowSmem carve_fixed(char* smem) {
    RowSmem sm;
    sm.As = reinterpret_cast<bf16*>(smem);
    sm.Fs = reinterpret_cast<float*>(smem + (size_t)MMT_ROWS * LDA_MAX * 2);
    sm.ldf = LDF;
    sm.Gs = sm.Fs + (size_t)MMT_ROWS * LDF;
    sm.Xs = sm.Fs + (size_t)MMT_ROWS * LDF * (WITH_G ? 2 : 1);
    sm.ldx = LDX;
    sm.A2 = reinterpret_cast<bf16*>(sm.Xs + (size_t)MMT_ROWS * LDX);
    sm.lda2 = LDA2;
    return sm;
}
// the geometry launch_rowchain (api.hip) computes for SHAPE 128; it refuses the fixed instance when its own numbers differ
#define MMT_FIX128_LDF 132
#define MMT_FIX128_LDA2 136
#define MMT_FIX128_LDA_FWD 136
#define MMT_FIX128_LDA_BND 392
__device__ __forceinline__ RowGemmParams qkv128(RowGemmParams p) { p = shape_pin<128, 384>(p); p.lda_out = 128; p.h = 8; p.DKP = 16; return p; }
__device__ __forceinline__ RowGemmParams dO128(RowGemmParams p) { p = shape_pin<128, 128>(p); p.h = 8; p.DKP = 16; return p; }
// SHAPE 256 = the MFT's per-modality stacks (configs[2], configs[4]): d_model = h d_k = 256 (8 heads of 32), d_ff = 128; the backward
// chains are the WIDE instances (K-chunked dQKV tile, x-hat recomputed)
#define MMT_FIX256_LDA 264
#define MMT_FIX256_LDA_BND 520
#define MMT_FIX256_LDF_FWD 132
#define MMT_FIX256_LDF_BWD 260
#define MMT_FIX256_LDX 260
#define MMT_FIX256_LDA2 136
__device__ __forceinline__ RowGemmParams qkv256(RowGemmParams p) { p = shape_pin<256, 768>(p); p.lda_out = 256; p.h = 8; p.DKP = 32; return p; }
__device__ __forceinline__ RowGemmParams dO256(RowGemmParams p) { p = shape_pin<256, 256>(p); p.h = 8; p.DKP = 32; return p; }
__device__ __forceinline__ RowGemmParams lnbwd256(RowGemmParams p) { p.no_gs = 1; return p; }

template <bool DEVSEED, int SHAPE = 0>
__global__ __launch_bounds__(MMT_RTHREADS, (MMT_ROWS == 16 || MMT_RTHREADS == 512) ? 4 : 2) void encoder_post_attn_fwd4_kernel(const RowChain4 ch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    warm_weights(ch.b.W, ch.b.NP, ch.b.KP); warm_weights(ch.c.W, ch.c.NP, ch.c.KP);
    warm_weights(ch.d.W, ch.d.NP, ch.d.KP);
    if (SHAPE == 128) {
        const RowSmem sm = carve_fixed<MMT_FIX128_LDA_FWD, MMT_FIX128_LDF, false, 132, MMT_FIX128_LDA2>(smem);
        rowgemm_stage<EPI_PLAIN, false, ASRC_GLOBAL, KEEP_X, DEVSEED>(shape_pin<128, 128>(role_plain(ch.a)), sm);
        rowgemm_stage<EPI_PLAIN, true, ASRC_X, KEEP_A2, DEVSEED>(shape_pin<128, 128>(role_ffn1(ch.b)), sm);
        rowgemm_stage<EPI_PLAIN, false, ASRC_A2, RES_X | KEEP_X, DEVSEED>(shape_pin<128, 128>(role_plain(ch.c)), sm);
        rowgemm_stage<EPI_FRAG, true, ASRC_X, 0, false>(qkv128(role_qkv(ch.d)), sm);
        return;
    }
    if (SHAPE == 256) {
        const RowSmem sm = carve_fixed<MMT_FIX256_LDA, MMT_FIX256_LDF_FWD, false, MMT_FIX256_LDX, MMT_FIX256_LDA2>(smem);
        rowgemm_stage<EPI_PLAIN, false, ASRC_GLOBAL, KEEP_X, DEVSEED>(shape_pin<256, 256>(role_plain(ch.a)), sm);
        rowgemm_stage<EPI_PLAIN, true, ASRC_X, KEEP_A2, DEVSEED>(shape_pin<256, 128>(role_ffn1(ch.b)), sm);
        rowgemm_stage<EPI_PLAIN, false, ASRC_A2, RES_X | KEEP_X, DEVSEED>(shape_pin<128, 256>(role_plain(ch.c)), sm);
        rowgemm_stage<EPI_FRAG, true, ASRC_X, 0, false>(qkv256(role_qkv(ch.d)), sm);
        return;
    }
    const RowSmem sm = rowchain_carve(smem, ch, false);
    rowgemm_stage<EPI_PLAIN, false, ASRC_GLOBAL, KEEP_X, DEVSEED>(role_plain(ch.a), sm);
    rowgemm_stage<EPI_PLAIN, true, ASRC_X, KEEP_A2, DEVSEED>(role_ffn1(ch.b), sm);
    rowgemm_stage<EPI_PLAIN, false, ASRC_A2, RES_X | KEEP_X, DEVSEED>(role_plain(ch.c), sm);
    rowgemm_stage<EPI_FRAG, true, ASRC_X, 0, false>(role_qkv(ch.d), sm);
}

// layer 0's two single-stage launches as fixed-shape instances (SHAPE 128)
__global__ __launch_bounds__(MMT_RTHREADS, (MMT_ROWS == 16 || MMT_RTHREADS == 512) ? 4 : 2) void encoder_ln1_qkv128_kernel(const RowGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RowSmem sm = carve_fixed<MMT_FIX128_LDA_FWD, MMT_FIX128_LDF, false, 0, 0>(smem);
    RowGemmParams q = qkv128(role_qkv(p));
    q.lda = 128;
    rowgemm_stage<EPI_FRAG, true, ASRC_GLOBAL, 0, false>(q, sm);
}
template <bool DEVSEED>
__global__ __launch_bounds__(MMT_RTHREADS, (MMT_ROWS == 16 || MMT_RTHREADS == 512) ? 4 : 2) void encoder_bwd_qkv_ln1_128_kernel(const RowGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const RowSmem sm = carve_fixed<MMT_FIX128_LDA_BND, MMT_FIX128_LDF, true, 0, 0>(smem);
    RowGemmParams q = shape_pin<384, 128>(role_lnbwd(p));
    q.next_drop.thr16 = 0; q.kchunk = 0; q.no_gs = 0;
    rowgemm_stage<EPI_LNBWD, false, ASRC_GLOBAL, 0, DEVSEED, false>(q, sm);
}

// Backward, from the layer-output gradient dx2 down to the attention core's operands:
//     dh  = (drop'(dx2) W2) * relu'(hid) * drop'          (dh kept in LDS as the next A tile; dx2^T, dh^T emitted for dW)
//     dx1 = dx2 + LN2bwd(dh W1)                            (drop'(dx1) kept in LDS as the next A tile)
//     dO  = drop'(dx1) Wo  -> fragment layouts + delta     (dx1^T emitted for dW)
template <bool WIDE, bool DEVSEED, int SHAPE = 0>
__global__ __launch_bounds__(MMT_RTHREADS, (MMT_ROWS == 16 || MMT_RTHREADS == 512) ? 4 : 2) void encoder_pre_attn_bwd_kernel(const RowChain3 ch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    warm_weights(ch.b.W, ch.b.NP, ch.b.KP); warm_weights(ch.c.W, ch.c.NP, ch.c.KP);
    if (SHAPE == 128 && !WIDE) {
        const RowSmem sm = carve_fixed<MMT_FIX128_LDA_FWD, MMT_FIX128_LDF, true, 0, MMT_FIX128_LDA2>(smem);
        rowgemm_stage<EPI_PLAIN, false, ASRC_GLOBAL, KEEP_A2, DEVSEED>(shape_pin<128, 128>(role_bwd_relu(ch.a)), sm);
        rowgemm_stage<EPI_LNBWD, false, ASRC_A2, KEEP_AS, DEVSEED, WIDE>(shape_pin<128, 128>(role_lnbwd(ch.b)), sm);
        rowgemm_stage<EPI_FRAG, false, ASRC_AS, 0, false>(dO128(role_dO(ch.c)), sm);
        return;
    }
    if (SHAPE == 256 && WIDE) {
        const RowSmem sm = carve_fixed<MMT_FIX256_LDA, MMT_FIX256_LDF_BWD, false, 0, MMT_FIX256_LDA2>(smem);
        rowgemm_stage<EPI_PLAIN, false, ASRC_GLOBAL, KEEP_A2, DEVSEED>(shape_pin<256, 128>(role_bwd_relu(ch.a)), sm);
        rowgemm_stage<EPI_LNBWD, false, ASRC_A2, KEEP_AS, DEVSEED, WIDE>(lnbwd256(shape_pin<128, 256>(role_lnbwd(ch.b))), sm);
        rowgemm_stage<EPI_FRAG, false, ASRC_AS, 0, false>(dO256(role_dO(ch.c)), sm);
        return;
    }
    const RowSmem sm = rowchain_carve(smem, ch, !WIDE);
    rowgemm_stage<EPI_PLAIN, false, ASRC_GLOBAL, KEEP_A2, DEVSEED>(role_bwd_relu(ch.a), sm);
    rowgemm_stage<EPI_LNBWD, false, ASRC_A2, KEEP_AS, DEVSEED, WIDE>(role_lnbwd(ch.b), sm);      // dx1 -> global (fp32) and, as the next A tile, LDS (bf16)
    rowgemm_stage<EPI_FRAG, false, ASRC_AS, 0, false>(role_dO(ch.c), sm);
}


// Backward across a layer boundary: the attention block of layer l closes (dx = dx1 + LN1bwd(dQKV Wqkv), the layer's input gradient)
// and the feed-forward block of layer l-1 opens on the same 32 windows (chain above), in ONE kernel: dx goes to global memory (fp32:
// the residual gradient of the LayerNorm-2 backward two stages later reads it back — written and read by this workgroup only, with
// workgroup barriers, each draining the stores, in between) and, dropped and rounded to bf16, straight into the next A tile.  One
// launch and one staging pass less per layer than `bwd_qkv+ln1` followed by the chain.  (WIDE: 83 KB of LDS at d_model = 256, one
// workgroup per CU — at that width two workgroups sharing a CU take twice as long each anyway)
template <bool WIDE, bool DEVSEED, int SHAPE = 0>
__global__ __launch_bounds__(MMT_RTHREADS, (MMT_ROWS == 16 || MMT_RTHREADS == 512) ? 4 : 2) void encoder_bwd_boundary_kernel(const RowChain4 ch) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    warm_weights(ch.b.W, ch.b.NP, ch.b.KP); warm_weights(ch.c.W, ch.c.NP, ch.c.KP); warm_weights(ch.d.W, ch.d.NP, ch.d.KP);
    if (SHAPE == 128 && !WIDE) {
        const RowSmem sm = carve_fixed<MMT_FIX128_LDA_BND, MMT_FIX128_LDF, true, 0, MMT_FIX128_LDA2>(smem);
        rowgemm_stage<EPI_LNBWD, false, ASRC_GLOBAL, KEEP_AS, DEVSEED, WIDE>(shape_pin<384, 128>(role_lnbwd(ch.a)), sm);
        rowgemm_stage<EPI_PLAIN, false, ASRC_AS, KEEP_A2, DEVSEED>(shape_pin<128, 128>(role_bwd_relu(ch.b)), sm);
        rowgemm_stage<EPI_LNBWD, false, ASRC_A2, KEEP_AS, DEVSEED, WIDE>(shape_pin<128, 128>(role_lnbwd(ch.c)), sm);
        rowgemm_stage<EPI_FRAG, false, ASRC_AS, 0, false>(dO128(role_dO(ch.d)), sm);
        return;
    }
    if (SHAPE == 256 && WIDE) {
        const RowSmem sm = carve_fixed<MMT_FIX256_LDA_BND, MMT_FIX256_LDF_BWD, false, 0, MMT_FIX256_LDA2>(smem);
        RowGemmParams a = lnbwd256(shape_pin<768, 256>(role_lnbwd(ch.a)));
        a.kchunk = 512;
        rowgemm_stage<EPI_LNBWD, false, ASRC_GLOBAL, KEEP_AS, DEVSEED, WIDE>(a, sm);
        rowgemm_stage<EPI_PLAIN, false, ASRC_AS, KEEP_A2, DEVSEED>(shape_pin<256, 128>(role_bwd_relu(ch.b)), sm);
        rowgemm_stage<EPI_LNBWD, false, ASRC_A2, KEEP_AS, DEVSEED, WIDE>(lnbwd256(shape_pin<128, 256>(role_lnbwd(ch.c))), sm);
        rowgemm_stage<EPI_FRAG, false, ASRC_AS, 0, false>(dO256(role_dO(ch.d)), sm);
        return;
    }
    const RowSmem sm = rowchain_carve(smem, ch, !WIDE);
    rowgemm_stage<EPI_LNBWD, false, ASRC_GLOBAL, KEEP_AS, DEVSEED, WIDE>(role_lnbwd(ch.a), sm);        // layer l:   dx -> global (fp32) + next A tile (bf16, dropped)
    rowgemm_stage<EPI_PLAIN, false, ASRC_AS, KEEP_A2, DEVSEED>(role_bwd_relu(ch.b), sm);                     // layer l-1: dh
    rowgemm_stage<EPI_LNBWD, false, ASRC_A2, KEEP_AS, DEVSEED, WIDE>(role_lnbwd(ch.c), sm);                              //            dx1
    rowgemm_stage<EPI_FRAG, false, ASRC_AS, 0, false>(role_dO(ch.d), sm);                                  //            dO fragments + delta
}
