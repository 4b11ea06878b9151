// Support kernels around the row-local GEMM and the attention core:
//   * encoder_prep_kernel  — fp32 master weights (flat, reference layout) -> zero-padded bf16 operand
//                            copies (forward and transposed forms, head-padded where heads are < 16 wide)
//   * layernorm_fwd/bwd    — the reference's LayerNorm alone (final norm of a stack; standalone module)
//   * wgrad_kernel         — dW[n][k] = sum_m dY[m][n] X[m][k] from row-major operands (transposing LDS reads), split over windows
//   * finalize kernels     — deterministic sums of the split slabs into the flat fp32 gradient
#pragma once
#include "common.h"

// ---- flat parameter layout of one encoder layer (fp32 elements), mirroring the reference's
//      registration order (transformer/MFT/multiTransformer.py:43,15-16,100,84-85)
struct LayerLayout {
    int d, f, h, dk, DKP, HD, HDP, DP, FP, NQ;
    __host__ __device__ size_t oW(int i) const { return (size_t)i * ((size_t)d * d + d); }   // linears[i].weight
    __host__ __device__ size_t ob(int i) const { return oW(i) + (size_t)d * d; }             // linears[i].bias
    __host__ __device__ size_t oW1() const { return oW(4); }
    __host__ __device__ size_t ob1() const { return oW1() + (size_t)f * d; }
    __host__ __device__ size_t oW2() const { return ob1() + f; }
    __host__ __device__ size_t ob2() const { return oW2() + (size_t)d * f; }
    __host__ __device__ size_t oln(int i) const { return ob2() + d + (size_t)i * d; }        // ln1.a, ln1.b, ln2.a, ln2.b
    __host__ __device__ size_t stride() const { return oln(4); }
    // prepared (bf16) block of one layer
    __host__ __device__ size_t pWqkv() const { return 0; }
    __host__ __device__ size_t pWqkvT() const { return pWqkv() + (size_t)NQ * DP; }
    __host__ __device__ size_t pWo() const { return pWqkvT() + (size_t)DP * NQ; }
    __host__ __device__ size_t pWoT() const { return pWo() + (size_t)DP * HDP; }
    __host__ __device__ size_t pW1() const { return pWoT() + (size_t)HDP * DP; }
    __host__ __device__ size_t pW1T() const { return pW1() + (size_t)FP * DP; }
    __host__ __device__ size_t pW2() const { return pW1T() + (size_t)DP * FP; }
    __host__ __device__ size_t pW2T() const { return pW2() + (size_t)DP * FP; }
    __host__ __device__ size_t pstride() const { return pW2T() + (size_t)FP * DP; }
    // prepared (fp32) padded biases of one layer
    __host__ __device__ size_t qbqkv() const { return 0; }
    __host__ __device__ size_t qbo() const { return NQ; }
    __host__ __device__ size_t qb1() const { return (size_t)NQ + DP; }
    __host__ __device__ size_t qb2() const { return (size_t)NQ + DP + FP; }
    __host__ __device__ size_t qstride() const { return (size_t)NQ + 2 * DP + FP; }
};

inline LayerLayout make_layout(int d, int f, int h) {
    LayerLayout L;
    L.d = d; L.f = f; L.h = h; L.dk = d / h;
    L.DKP = (L.dk <= 16) ? 16 : (L.dk <= 32 ? 32 : 64);
    L.HD = h * L.DKP; L.HDP = round_up(L.HD, 64);
    L.DP = round_up(d, 64); L.FP = round_up(f, 64);
    L.NQ = round_up(3 * L.HD, 64);
    return L;
}

// source feature index of head-padded column c (= head*DKP + e), or -1 for a pad column
__device__ __forceinline__ int unpad_head(int c, const LayerLayout& L) {
    if (c >= L.HD) return -1;
    const int head = c / L.DKP, e = c - head * L.DKP;
    return (e < L.dk) ? head * L.dk + e : -1;
}

// Prepared (bf16, padded, transposed) weights and padded biases of one layer: block bx of nbx; every element is computed from its coordinates
__device__ __forceinline__ void encoder_prep_block(const float* __restrict__ params, bf16* __restrict__ wprep, float* __restrict__ bprep,
                                                   const LayerLayout& L, int layer, int bx, int nbx) {
    const float* P = params + (size_t)layer * L.stride();
    bf16* W = wprep + (size_t)layer * L.pstride();
    float* Bp = bprep + (size_t)layer * L.qstride();
    const size_t total = L.pstride();
    const int d = L.d, f = L.f;
    for (size_t idx = (size_t)bx * blockDim.x + threadIdx.x; idx < total + L.qstride();
         idx += (size_t)nbx * blockDim.x) {
        if (idx >= total) {                 // padded fp32 biases
            const int i = (int)(idx - total);
            float v = 0.f;
            if (i < L.NQ) {
                const int wi = i / L.HD;
                if (wi < 3) { const int s = unpad_head(i - wi * L.HD, L); if (s >= 0) v = P[L.ob(wi) + s]; }
            } else if (i < L.NQ + L.DP) { const int n = i - L.NQ; if (n < d) v = P[L.ob(3) + n]; }
            else if (i < L.NQ + L.DP + L.FP) { const int n = i - L.NQ - L.DP; if (n < f) v = P[L.ob1() + n]; }
            else { const int n = i - L.NQ - L.DP - L.FP; if (n < d) v = P[L.ob2() + n]; }
            Bp[i] = v;
            continue;
        }
        float v = 0.f;
        if (idx < L.pWqkvT()) {             // Wqkv [NQ][DP]: row = (wi*h+head)*DKP+e, col = input feature
            const int n = (int)(idx / L.DP), k = (int)(idx % L.DP), wi = n / L.HD;
            if (wi < 3 && k < d) { const int s = unpad_head(n - wi * L.HD, L); if (s >= 0) v = P[L.oW(wi) + (size_t)s * d + k]; }
        } else if (idx < L.pWo()) {         // Wqkv^T [DP][NQ]
            const size_t i = idx - L.pWqkvT();
            const int k = (int)(i / L.NQ), n = (int)(i % L.NQ), wi = n / L.HD;
            if (wi < 3 && k < d) { const int s = unpad_head(n - wi * L.HD, L); if (s >= 0) v = P[L.oW(wi) + (size_t)s * d + k]; }
        } else if (idx < L.pWoT()) {        // Wo [DP][HDP]: col = head-padded context feature
            const size_t i = idx - L.pWo();
            const int n = (int)(i / L.HDP), c = (int)(i % L.HDP);
            if (n < d) { const int s = unpad_head(c, L); if (s >= 0) v = P[L.oW(3) + (size_t)n * d + s]; }
        } else if (idx < L.pW1()) {         // Wo^T [HDP][DP]
            const size_t i = idx - L.pWoT();
            const int c = (int)(i / L.DP), n = (int)(i % L.DP);
            if (n < d) { const int s = unpad_head(c, L); if (s >= 0) v = P[L.oW(3) + (size_t)n * d + s]; }
        } else if (idx < L.pW1T()) {        // W1 [FP][DP]
            const size_t i = idx - L.pW1();
            const int n = (int)(i / L.DP), k = (int)(i % L.DP);
            if (n < f && k < d) v = P[L.oW1() + (size_t)n * d + k];
        } else if (idx < L.pW2()) {         // W1^T [DP][FP]
            const size_t i = idx - L.pW1T();
            const int k = (int)(i / L.FP), n = (int)(i % L.FP);
            if (n < f && k < d) v = P[L.oW1() + (size_t)n * d + k];
        } else if (idx < L.pW2T()) {        // W2 [DP][FP]
            const size_t i = idx - L.pW2();
            const int n = (int)(i / L.FP), k = (int)(i % L.FP);
            if (n < d && k < f) v = P[L.oW2() + (size_t)n * f + k];
        } else {                            // W2^T [FP][DP]
            const size_t i = idx - L.pW2T();
            const int k = (int)(i / L.DP), n = (int)(i % L.DP);
            if (n < d && k < f) v = P[L.oW2() + (size_t)n * f + k];
        }
        W[idx] = (bf16)v;
    }
}
// grid = (blocks, n_layers)
__global__ void encoder_prep_kernel(const float* __restrict__ params, bf16* __restrict__ wprep,
                                    float* __restrict__ bprep, LayerLayout L) {
    encoder_prep_block(params, wprep, bprep, L, blockIdx.y, blockIdx.x, gridDim.x);
}
// Train mode: weight preparation and the dropout-bit generator are independent, so they share ONE launch (grid.y = layer): the first
// `gen_blocks` workgroups of a layer draw its attention-dropout decisions, the last `gridDim.x - gen_blocks` prepare its weights — they
// are dispatched last and run in the slots the generator's incomplete last round leaves free.  (Placed first, 900 small preparation
// workgroups per layer at the generator's occupancy — its registers and 36 KB of LDS — made the launch 8 us LONGER than the two apart.)
__global__ __launch_bounds__(256, 4) void encoder_prep_maskgen_kernel(const float* __restrict__ params, bf16* __restrict__ wprep,
                                                                      float* __restrict__ bprep, LayerLayout L, int gen_blocks,
                                                                      const MaskGenParams P) {
    if ((int)blockIdx.x >= gen_blocks) { encoder_prep_block(params, wprep, bprep, L, blockIdx.y, blockIdx.x - gen_blocks, gridDim.x - gen_blocks); return; }
    attn_mask_gen_block(P, blockIdx.y, blockIdx.x);
}

// generic: fp32 [N][K] (ld = K) -> bf16 [NP][KP] zero padded, optionally transposed source
__global__ void pad_cast_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int N, int K, int NP, int KP,
                                int transpose_src) {
    const size_t total = (size_t)NP * KP;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / KP), k = (int)(idx % KP);
        float v = 0.f;
        if (n < N && k < K) v = transpose_src ? src[(size_t)k * N + n] : src[(size_t)n * K + k];
        dst[idx] = (bf16)v;
    }
}

// ---- LayerNorm alone (reference variant).  8 threads per row, 32 rows per workgroup. -------------
__global__ __launch_bounds__(MMT_THREADS) void layernorm_fwd_kernel(
        const float* __restrict__ x, const float* __restrict__ a, const float* __restrict__ b, float eps,
        float* __restrict__ y, float* __restrict__ stats, int M, int d) {
    const int row = threadIdx.x >> 3, j = threadIdx.x & 7, m = blockIdx.x * 32 + row;
    const bool ok = m < M;
    const float* xr = x + (size_t)(ok ? m : 0) * d;
    float s = 0.f;
    for (int c = j * 4; c < d; c += 32) { f32x4 v = *reinterpret_cast<const f32x4*>(xr + c); s += (v[0] + v[1]) + (v[2] + v[3]); }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
    const float mean = s / (float)d;
    float q = 0.f;
    for (int c = j * 4; c < d; c += 32) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const float t = v[i] - mean; q += t * t; }
    }
    q += __shfl_xor(q, 1); q += __shfl_xor(q, 2); q += __shfl_xor(q, 4);
    const float rstd = 1.0f / (sqrtf(q / (float)(d - 1)) + eps);
    if (!ok) return;
    if (j == 0 && stats) { stats[2 * (size_t)m] = mean; stats[2 * (size_t)m + 1] = rstd; }
    for (int c = j * 4; c < d; c += 32) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        f32x4 av = *reinterpret_cast<const f32x4*>(a + c), bv = *reinterpret_cast<const f32x4*>(b + c), o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = av[i] * ((v[i] - mean) * rstd) + bv[i];
        *reinterpret_cast<f32x4*>(y + (size_t)m * d + c) = o;
    }
}

// dx = LNbwd(dy); per-workgroup column sums for d ln_b (sum dy) and d ln_a (sum dy*xhat) -> colpart[g][2][DP]
__global__ __launch_bounds__(MMT_THREADS) void layernorm_bwd_kernel(
        const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ a,
        const float* __restrict__ stats, float eps, float* __restrict__ dx, float* __restrict__ colpart,
        int M, int d, int DP) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Fs = reinterpret_cast<float*>(smem);            // [32][DP+4]  dy
    float* Gs = Fs + 32 * (DP + 4);                        // [32][DP+4]  dy * xhat
    const int ld = DP + 4;
    const int row = threadIdx.x >> 3, j = threadIdx.x & 7, m = blockIdx.x * 32 + row;
    const bool ok = m < M;
    float mean = 0.f, rstd = 1.f;
    if (ok) { mean = stats[2 * (size_t)m]; rstd = stats[2 * (size_t)m + 1]; }
    float s1 = 0.f, s2 = 0.f;
    for (int c = j * 4; c < DP; c += 32) {
        f32x4 g = {0.f, 0.f, 0.f, 0.f}, gx = g;
        if (ok && c < d) {
            g = *reinterpret_cast<const f32x4*>(dy + (size_t)m * d + c);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)m * d + c);
            const f32x4 av = *reinterpret_cast<const f32x4*>(a + c);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float xh = (xv[i] - mean) * rstd;
                gx[i] = g[i] * xh;
                s1 += g[i] * av[i];
                s2 += g[i] * av[i] * xh;
            }
        }
        *reinterpret_cast<f32x4*>(Fs + row * ld + c) = g;
        *reinterpret_cast<f32x4*>(Gs + row * ld + c) = gx;
    }
    s1 += __shfl_xor(s1, 1); s1 += __shfl_xor(s1, 2); s1 += __shfl_xor(s1, 4);
    s2 += __shfl_xor(s2, 1); s2 += __shfl_xor(s2, 2); s2 += __shfl_xor(s2, 4);
    if (ok) {
        const float sigma = 1.0f / rstd - eps;
        const float k1 = s1 / (float)d, k2 = s2 / ((float)(d - 1) * sigma);
        for (int c = j * 4; c < d; c += 32) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(Fs + row * ld + c);
            const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (size_t)m * d + c);
            const f32x4 av = *reinterpret_cast<const f32x4*>(a + c);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = rstd * (g[i] * av[i] - k1) - k2 * ((xv[i] - mean) * rstd);
            *reinterpret_cast<f32x4*>(dx + (size_t)m * d + c) = o;
        }
    }
    __syncthreads();
    if (colpart) {
        for (int c = threadIdx.x; c < DP; c += MMT_THREADS) {
            float sb = 0.f, sa = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) { sb += Fs[r * ld + c]; sa += Gs[r * ld + c]; }
            colpart[(size_t)blockIdx.x * 2 * DP + c] = sb;
            colpart[(size_t)blockIdx.x * 2 * DP + DP + c] = sa;
        }
    }
}

// ---- weight gradients ---------------------------------------------------------------------------
// dW[n][k] = sum_m A[m][n] * B[m][k]  with A = dY and B = X both ROW-MAJOR bf16 [MP rows][ld] (rows >= M are zero and never
// written): the contraction runs over the windows m, the slow index of both operands.  Round 1 had every producer also write a
// transposed ("T layout") copy of these eight operands so that an MFMA fragment was a plain 16-byte LDS read; those copies were
// 45 % of the row kernels' HBM writes.  Now a [64 windows][64 features] tile of each operand is staged as it lies in memory
// (full 128-byte lines) and the fragments come out of LDS already transposed by ds_read_b64_tr_b16 (gfx950's transposing read:
// each 16-lane group gets a 4-window x 16-feature block column-major), two reads per 16-window k-step and operand.  LDS rows
// are 192 bytes apart: the four rows a 32-lane half reads then fall into the four quarters of the 64 banks (conflict-free).
// One workgroup = one 64x64 output tile (4 waves as 2x2 of 32x32) over one slice of the windows; slices go to separate fp32
// slabs [nsplit][NP][KP] (+ [nsplit][NP] for the bias gradient = column sums of A, taken from the A fragments on VALU), summed
// later by the finalize kernels (deterministic).
struct WgradJob {
    const bf16* A; const bf16* B; float* out; float* bias_out;
    int lda, ldb;
    int NPj, KPj, tile0, tiles_k;
};
#define MMT_MAX_WGRAD_JOBS 64
struct WgradJobs { WgradJob j[MMT_MAX_WGRAD_JOBS]; int njobs; int MP; int M16; int mchunk;
                   int tiles_per_layer, nlayers, nsplit; };    // > 0: 1-D XCD-aware grid (see wgrad_kernel); 0: grid = (tiles, splits)

__global__ __launch_bounds__(MMT_THREADS, 2) void wgrad_kernel(const WgradJobs jobs) {
    constexpr int LDR = 96;                                    // bf16 elements per LDS row (192 bytes)
    __shared__ __attribute__((aligned(16))) bf16 As[2][64 * LDR];
    __shared__ __attribute__((aligned(16))) bf16 Bs[2][64 * LDR];
    // Workgroup -> (tile, window split).  The tiles of one layer and one window split read the same eight operands (every
    // operand column block is shared by 2..6 tiles).  Consecutive workgroup ids go round-robin to the 8 XCDs, each with a private
    // L2, so with a plain (tile, split) grid no two sharers met in an L2 and FETCH_SIZE was 2.7x the operand bytes.  The
    // encoder launch uses a 1-D grid in which ids congruent mod 8 — one XCD — carry whole (layer, split) units one after the
    // other.
    int tile_g = blockIdx.x, split = blockIdx.y;
    if (jobs.tiles_per_layer > 0) {
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        const int uix = slot / jobs.tiles_per_layer, tix = slot - uix * jobs.tiles_per_layer;
        const int unit = uix * 8 + xcd;
        if (unit >= jobs.nlayers * jobs.nsplit) return;         // whole workgroup, before any barrier
        split = unit / jobs.nlayers;
        tile_g = (unit - split * jobs.nlayers) * jobs.tiles_per_layer + tix;
    }
    int ji = 0;
    for (int i = 1; i < jobs.njobs; ++i) if (tile_g >= jobs.j[i].tile0) ji = i;
    const WgradJob& J = jobs.j[ji];
    const int tile = tile_g - J.tile0, tn = tile / J.tiles_k, tk = tile - tn * J.tiles_k;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int wn = wave >> 1, wk = wave & 1;
    const int mbeg = split * jobs.mchunk, mend = min(jobs.MP, mbeg + jobs.mchunk);
    // staging role of this thread: windows row0 and row0+32 of each 64-window chunk, 16-byte segment seg of the 64 features
    const int row0 = tid >> 3, seg = tid & 7;
    const bf16* ag = J.A + (size_t)row0 * J.lda + tn * 64 + seg * 8;
    const bf16* bg = J.B + (size_t)row0 * J.ldb + tk * 64 + seg * 8;
    const size_t ahalf = (size_t)32 * J.lda, bhalf = (size_t)32 * J.ldb;
    const bool want_bias = (J.bias_out != nullptr) && tk == 0 && wk == 0;
    // transposing-read role: 16-lane group g = lane >> 4 covers features 16(g&1) .. +15 of the wave's 32, windows 8(g>>1) .. +7 of a
    // k-step; inside the group lane 4q + p addresses window q of a 4-window block, features 4p .. 4p+3
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int aoff = (8 * (g >> 1) + q) * LDR + wn * 32 + 16 * (g & 1) + 4 * pp;
    const int boff = (8 * (g >> 1) + q) * LDR + wk * 32 + 16 * (g & 1) + 4 * pp;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    if (mbeg < mend) {
        bf16x8 ra0 = *reinterpret_cast<const bf16x8*>(ag + (size_t)mbeg * J.lda), ra1 = *reinterpret_cast<const bf16x8*>(ag + (size_t)mbeg * J.lda + ahalf);
        bf16x8 rb0 = *reinterpret_cast<const bf16x8*>(bg + (size_t)mbeg * J.ldb), rb1 = *reinterpret_cast<const bf16x8*>(bg + (size_t)mbeg * J.ldb + bhalf);
        int buf = 0;
        for (int m = mbeg; m < mend; m += 64) {
            *reinterpret_cast<bf16x8*>(&As[buf][row0 * LDR + seg * 8]) = ra0;
            *reinterpret_cast<bf16x8*>(&As[buf][(row0 + 32) * LDR + seg * 8]) = ra1;
            *reinterpret_cast<bf16x8*>(&Bs[buf][row0 * LDR + seg * 8]) = rb0;
            *reinterpret_cast<bf16x8*>(&Bs[buf][(row0 + 32) * LDR + seg * 8]) = rb1;
            __syncthreads();
            if (m + 64 < mend) {                                // next chunk in flight behind this chunk's MFMAs
                const size_t ao = (size_t)(m + 64) * J.lda, bo = (size_t)(m + 64) * J.ldb;
                ra0 = *reinterpret_cast<const bf16x8*>(ag + ao); ra1 = *reinterpret_cast<const bf16x8*>(ag + ao + ahalf);
                rb0 = *reinterpret_cast<const bf16x8*>(bg + bo); rb1 = *reinterpret_cast<const bf16x8*>(bg + bo + bhalf);
            }
            const bf16* ap = &As[buf][aoff];
            const bf16* bp = &Bs[buf][boff];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8 a = tr_frag(ap + ks * 16 * LDR, LDR);
                const bf16x8 b = tr_frag(bp + ks * 16 * LDR, LDR);
                acc = mfma32(a, b, acc);
                if (want_bias) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) bsum += (float)a[i];
                }
            }
            buf ^= 1;       // the buffer written at chunk c is next written at chunk c+2, after the barrier of chunk c+1
        }
    }
    const int n0 = tn * 64 + wn * 32, k0 = tk * 64 + wk * 32;
    float* out = J.out + (size_t)split * J.NPj * J.KPj;
#pragma unroll
    for (int i = 0; i < 16; ++i) out[(size_t)(n0 + acc32_row(i, hh)) * J.KPj + k0 + r] = acc[i];
    if (want_bias) {
        bsum += __shfl_xor(bsum, 32);
        if (hh == 0) J.bias_out[(size_t)split * J.NPj + n0 + r] = bsum;
    }
}

// generic slab sum with crop: dst[n][k] (ld = K) = sum_s slab[s][n][k] for n<N, k<K
__global__ void slab_sum_kernel(const float* __restrict__ slab, int nsplit, int NP, int KP,
                                float* __restrict__ dst, int N, int K) {
    const size_t total = (size_t)N * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(idx / K), k = (int)(idx % K);
        float s = 0.f;
        for (int sp = 0; sp < nsplit; ++sp) s += slab[((size_t)sp * NP + n) * KP + k];
        dst[idx] = s;
    }
}

// Encoder layer gradients -> flat fp32 gradient block of the layer (same layout as the parameters).
struct LayerSlabs {
    const float *dWqkv, *dbqkv, *dWo, *dbo, *dW1, *db1, *dW2, *db2;   // layer 0: [nsplit][..]; layer l at + l * slab_stride
    const float *ln1part, *ln2part;                                    // [G][2][DP]
    int nsplit, G;
    size_t slab_stride;                                                // floats between consecutive layers' slab sets
};

// ONE launch turns every partial result of the backward pass into the flat gradient (round 2 ran four: ~1.6 us of idle GPU between
// any two kernels of a hipGraph plus a 6-9 us latency floor per tiny kernel).  1-D grid of 1024-thread workgroups:
//   workgroups [0, nln * lnblocks): LayerNorm parameter gradients of job j = id / lnblocks: out_a[c] = sum_g part[g][1][c],
//       out_b[c] = sum_g part[g][0][c] over the G row-tile partials, 32 columns x 32 row groups per workgroup, fixed order;
//   the rest, wblocks per layer: the layer's weight / bias slabs summed over the window splits in a fixed order.
#define MMT_MAX_LN_JOBS 40
struct LnJobs { const float* part[MMT_MAX_LN_JOBS]; float* out_a[MMT_MAX_LN_JOBS]; int G[MMT_MAX_LN_JOBS]; int n, lnblocks, wblocks, DP, d; };

__global__ __launch_bounds__(1024) void encoder_finalize_kernel(LayerSlabs S0, LayerLayout L, float* __restrict__ grad0, LnJobs J) {
    __shared__ float red[32][33];
    if ((int)blockIdx.x < J.n * J.lnblocks) {
        const int job = blockIdx.x / J.lnblocks, r = blockIdx.x - job * J.lnblocks, which = r & 1, cb = r >> 1;
        const float* part = J.part[job];
        const int G = J.G[job];
        const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5, c = cb * 32 + cx;
        float s = 0.f;
        if (c < J.DP)
            for (int g = rg; g < G; g += 32) s += part[(size_t)g * 2 * J.DP + (size_t)which * J.DP + c];
        red[rg][cx] = s;
        __syncthreads();
        if (rg == 0 && c < J.d) {
            float t = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) t += red[i][cx];
            (J.out_a[job] + (which ? 0 : J.d))[c] = t;          // (a_2, b_2) are adjacent in the parameter order
        }
        return;
    }
    const int wid = blockIdx.x - J.n * J.lnblocks, layer = wid / J.wblocks, wb = wid - layer * J.wblocks;
    LayerSlabs S = S0;
    const size_t so = (size_t)layer * S0.slab_stride;
    S.dWqkv += so; S.dbqkv += so; S.dWo += so; S.dbo += so; S.dW1 += so; S.db1 += so; S.dW2 += so; S.db2 += so;
    float* grad = grad0 + (size_t)layer * L.stride();
    const int d = L.d, f = L.f;
    const size_t n_w = L.oln(0);                   // weights and biases of the layer
    for (size_t idx = (size_t)wb * blockDim.x + threadIdx.x; idx < n_w; idx += (size_t)J.wblocks * blockDim.x) {
        const float* src; size_t off, sstride;
        if (idx < L.oW1()) {                        // the four attention linears
            const int wi = (int)(idx / ((size_t)d * d + d));
            const size_t r = idx - L.oW(wi);
            if (r < (size_t)d * d) {
                const int n = (int)(r / d), k = (int)(r % d);
                if (wi < 3) {                       // row n = head*dk+e -> head-padded row of dWqkv
                    const int head = n / L.dk, e = n - head * L.dk;
                    src = S.dWqkv; off = ((size_t)wi * L.HD + head * L.DKP + e) * L.DP + k; sstride = (size_t)L.NQ * L.DP;
                } else {                            // Wo column k = head*dk+e -> head-padded column
                    const int head = k / L.dk, e = k - head * L.dk;
                    src = S.dWo; off = (size_t)n * L.HDP + head * L.DKP + e; sstride = (size_t)L.DP * L.HDP;
                }
            } else {
                const int n = (int)(r - (size_t)d * d);
                if (wi < 3) { const int head = n / L.dk, e = n - head * L.dk; src = S.dbqkv; off = (size_t)wi * L.HD + head * L.DKP + e; sstride = L.NQ; }
                else { src = S.dbo; off = n; sstride = L.DP; }
            }
        } else if (idx < L.ob1()) { const size_t r = idx - L.oW1(); src = S.dW1; off = (r / d) * L.DP + (r % d); sstride = (size_t)L.FP * L.DP; }
        else if (idx < L.oW2()) { src = S.db1; off = idx - L.ob1(); sstride = L.FP; }
        else if (idx < L.ob2()) { const size_t r = idx - L.oW2(); src = S.dW2; off = (r / f) * L.FP + (r % f); sstride = (size_t)L.DP * L.FP; }
        else { src = S.db2; off = idx - L.ob2(); sstride = L.DP; }
        float s = 0.f;
        for (int sp = 0; sp < S.nsplit; ++sp) s += src[(size_t)sp * sstride + off];
        grad[idx] = s;
    }
}

// LayerNorm parameter gradients: out_a[c] = sum_g part[g][1][c], out_b[c] = sum_g part[g][0][c].
// grid = (DP/32, 2 {b,a}); 1024 threads = 32 columns x 32 row groups, coalesced 128-byte row reads,
// fixed summation order (deterministic).
// blockIdx.z batches several LayerNorms: item z reads part + z*part_stride and writes out_a/out_b + z*out_stride
__global__ __launch_bounds__(1024) void ln_param_finalize_kernel(const float* __restrict__ part0, int G, int DP, int d,
                                                                 float* __restrict__ out_a0, float* __restrict__ out_b0,
                                                                 size_t part_stride, size_t out_stride) {
    __shared__ float red[32][33];
    const float* part = part0 + (size_t)blockIdx.z * part_stride;
    float* out_a = out_a0 + (size_t)blockIdx.z * out_stride;
    float* out_b = out_b0 + (size_t)blockIdx.z * out_stride;
    const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cx, which = blockIdx.y;
    float s = 0.f;
    if (c < DP)
        for (int g = rg; g < G; g += 32) s += part[(size_t)g * 2 * DP + (size_t)which * DP + c];
    red[rg][cx] = s;
    __syncthreads();
    if (rg == 0 && c < d) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[i][cx];
        (which ? out_a : out_b)[c] = t;
    }
}


// Concordance correlation coefficient of one sequence per workgroup (transformer/SFT/train.py:42-50): population moments over the
// first lengths[b] windows, fp64 accumulation, fixed-order tree reduction (deterministic).
__global__ __launch_bounds__(256) void ccc_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                  const int* __restrict__ lengths, double* __restrict__ out, int T) {
    __shared__ double red[5][256];
    const int b = blockIdx.x, n = lengths[b];
    const float* p = pred + (size_t)b * T;
    const float* t = target + (size_t)b * T;
    double sp = 0, st = 0, spp = 0, stt = 0, spt = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double a = p[i], c = t[i];
        sp += a; st += c; spp += a * a; stt += c * c; spt += a * c;
    }
    red[0][threadIdx.x] = sp; red[1][threadIdx.x] = st; red[2][threadIdx.x] = spp; red[3][threadIdx.x] = stt; red[4][threadIdx.x] = spt;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
#pragma unroll
            for (int k = 0; k < 5; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double N = (double)n, mp = red[0][0] / N, mt = red[1][0] / N;
        const double vp = red[2][0] / N - mp * mp, vt = red[3][0] / N - mt * mt, cov = red[4][0] / N - mp * mt;
        out[b] = (n < 2) ? __builtin_nan("") : 2.0 * cov / (vt + vp + (mp - mt) * (mp - mt));
    }
}


// Training loss of the reference (transformer/SFT/train.py:133-137, criterion :538): MSELoss(reduction='sum')(out, target) divided by
// the number of valid windows, and its gradient 2 (out - target) / denom, in one pass (torch would run sub, pow, sum, div and four
// backward kernels).  Partial sums in fp64 per workgroup, summed in a fixed order by a second tiny launch: deterministic.
__global__ __launch_bounds__(256) void mse_sum_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target, float inv_denom,
                                                              float* __restrict__ dpred, double* __restrict__ part, size_t n) {
    __shared__ double red[256];
    double acc = 0;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const f32x4 p = reinterpret_cast<const f32x4*>(pred)[i], t = reinterpret_cast<const f32x4*>(target)[i];
        const f32x4 d = p - t;
        acc += (double)d[0] * d[0] + (double)d[1] * d[1] + (double)d[2] * d[2] + (double)d[3] * d[3];
        reinterpret_cast<f32x4*>(dpred)[i] = d * (2.0f * inv_denom);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        const float d = pred[i] - target[i];
        acc += (double)d * d;
        dpred[i] = d * (2.0f * inv_denom);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void mse_sum_final_kernel(const double* __restrict__ part, int nparts, float inv_denom, float* __restrict__ loss) {
    __shared__ double red[256];
    double acc = 0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += part[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] * (double)inv_denom);
}


// Adam step of the reference's optimiser (torch.optim.Adam with L2 weight decay: transformer/SFT/train.py:621) on up to
// MMT_ADAM_MAX_CHUNKS contiguous parameter ranges in ONE launch: the encoder's parameters, gradients and moments are flat buffers
// (multiTransformer.Encoder._flat_storage, functional._EncoderStackParamsFn), so the whole stack is one chunk; torch's foreach
// implementation runs ~10 multi-tensor kernels over the 98 tensors (90 us at configs[3]).  grid = (blocks, chunks).
#define MMT_ADAM_MAX_CHUNKS 48
struct AdamChunks { float* p[MMT_ADAM_MAX_CHUNKS]; const float* g[MMT_ADAM_MAX_CHUNKS]; float* m[MMT_ADAM_MAX_CHUNKS]; float* v[MMT_ADAM_MAX_CHUNKS];
                    unsigned long long n[MMT_ADAM_MAX_CHUNKS]; };
__global__ __launch_bounds__(256) void adam_step_kernel(const AdamChunks C, float step, float beta1, float beta2, float eps, float wd,
                                                        float bc2_sqrt) {       // step = lr / (1 - beta1^t), bc2_sqrt = sqrt(1 - beta2^t): taken in double on the host, like torch
    const int c = blockIdx.y;
    float* __restrict__ p = C.p[c]; const float* __restrict__ g = C.g[c]; float* __restrict__ m = C.m[c]; float* __restrict__ v = C.v[c];
    const size_t n = C.n[c];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float pi = p[i];
        const float gi = g[i] + wd * pi;                          // grad = grad.add(param, alpha=weight_decay)
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;       // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
        m[i] = mi; v[i] = vi;
        p[i] = pi - step * (mi / (sqrtf(vi) / bc2_sqrt + eps));   // param.addcdiv_(exp_avg, exp_avg_sq.sqrt() / bias_correction2_sqrt + eps, value=-step_size)
    }
}
