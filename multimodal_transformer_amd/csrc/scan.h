// T-sequential recurrences of the path, as persistent-state scan kernels for gfx950.
//
// LSTM scan (MFN's per-modality nn.LSTMCell, transformer/MFT/multiTransformer.py:152,208, and the SFT
// decoder's nn.LSTM step, transformer/SFT/multiTransformer.py:471-476).  Only gates += W_rec . h_{t-1} is
// recurrent; the input projection gx[t] = x_t W_ih^T + b_ih + b_hh is batched over all T by the row GEMM.
// One workgroup owns 16 sequences (the MFMA N dimension) for the whole scan; wave w owns hidden units
// [16w, 16w+16) and keeps its slice of W_rec as MFMA A fragments in registers for all T steps, the cell
// state in registers (fp32), and h crosses waves through a double-buffered bf16 LDS tile: one barrier
// per step.  Gate order i, f, g, o (torch).
#pragma once
#include "common.h"

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp2(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanh_f(float x) { return 2.0f * sigmoid_f(2.0f * x) - 1.0f; }

// W_rec (4H,H) fp32 -> forward operand Wf bf16 [4][HP16][KP] (Wf[q][j][k] = W[q*H+j][k]) and
// backward operand Wb bf16 [HP16][KP4] (Wb[j][q*HP16+j'] = W[q*H+j'][j]); zero padded.
__global__ void lstm_prep_kernel(const float* __restrict__ W, bf16* __restrict__ Wf, bf16* __restrict__ Wb,
                                 int H, int HP16, int KP, int KP4) {
    const size_t nf = (size_t)4 * HP16 * KP, nb = (size_t)HP16 * KP4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nf + nb; idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < nf) {
            const int k = (int)(idx % KP), j = (int)((idx / KP) % HP16), q = (int)(idx / ((size_t)KP * HP16));
            Wf[idx] = (bf16)((j < H && k < H) ? W[((size_t)q * H + j) * H + k] : 0.f);
        } else {
            const size_t i = idx - nf;
            const int c = (int)(i % KP4), j = (int)(i / KP4);
            float v = 0.f;
            if (c < 4 * HP16) { const int q = c / HP16, jp = c - q * HP16; if (j < H && jp < H) v = W[((size_t)q * H + jp) * H + j]; }
            Wb[i] = (bf16)v;
        }
    }
}

// grid = ceil(B/16); block = 64 * (HP16/16) <= NT.  MAXKS >= KP/32.  WREG: W_rec fragments stay in registers
// for the whole scan (HP16 <= 128); otherwise (HP16 = 256: 512 KB of bf16 weights exceed one CU's register
// file) they are re-streamed from L2 every step.
template <int MAXKS, int NT, bool WREG>
__global__ __launch_bounds__(NT) void lstm_scan_fwd_kernel(const float* __restrict__ gx, const bf16* __restrict__ Wf,
                                     const float* __restrict__ h0, const float* __restrict__ c0,
                                     float* __restrict__ h_all, float* __restrict__ c_all, float* __restrict__ acts,
                                     int T, int B, int H, int HP16, int KP) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ldh = KP + 8;
    bf16* hbuf = reinterpret_cast<bf16*>(smem);                 // [2][16][ldh]
    const int lane = threadIdx.x & 63, jt = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int nks = KP >> 5;
    const int b = blockIdx.x * 16 + l15, j0 = jt * 16 + 4 * lq;
    const bool live = (b < B) && (j0 < H);                      // H % 4 == 0: the 4 rows of a lane are all in or all out

    const bf16* wrow = Wf + (size_t)(jt * 16 + l15) * KP + 8 * lq;      // + q*HP16*KP + ks*32
    const size_t wq = (size_t)HP16 * KP;
    bf16x8 a[WREG ? 4 : 1][WREG ? MAXKS : 1];
    if (WREG) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ks = 0; ks < MAXKS; ++ks)
                if (ks < nks) a[q][ks] = *reinterpret_cast<const bf16x8*>(wrow + q * wq + ks * 32);
    }

    for (int i = threadIdx.x; i < 2 * 16 * ldh; i += blockDim.x) hbuf[i] = (bf16)0.f;
    __syncthreads();
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        if (c0) c = *reinterpret_cast<const f32x4*>(c0 + (size_t)b * H + j0);
        if (h0) {
            const f32x4 hv = *reinterpret_cast<const f32x4*>(h0 + (size_t)b * H + j0);
#pragma unroll
            for (int r = 0; r < 4; ++r) hbuf[l15 * ldh + j0 + r] = (bf16)hv[r];
        }
    }
    __syncthreads();

    f32x4 nxt[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        nxt[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live) nxt[q] = *reinterpret_cast<const f32x4*>(gx + ((size_t)0 * B + b) * 4 * H + (size_t)q * H + j0);
    }
    int cur = 0;
    for (int t = 0; t < T; ++t) {
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = nxt[q];
        if (t + 1 < T && live) {                                // next step's input projection in flight behind this step
#pragma unroll
            for (int q = 0; q < 4; ++q)
                nxt[q] = *reinterpret_cast<const f32x4*>(gx + ((size_t)(t + 1) * B + b) * 4 * H + (size_t)q * H + j0);
        }
        const bf16* hb = hbuf + cur * 16 * ldh + l15 * ldh + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks) {
            if (ks < nks) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(hb + ks * 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf16x8 af = WREG ? a[WREG ? q : 0][WREG ? ks : 0] : *reinterpret_cast<const bf16x8*>(wrow + q * wq + ks * 32);
                    acc[q] = mfma16(af, bf, acc[q]);
                }
            }
        }
        f32x4 ig, fg, gg, og, hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ig[r] = sigmoid_f(acc[0][r]); fg[r] = sigmoid_f(acc[1][r]); gg[r] = tanh_f(acc[2][r]); og[r] = sigmoid_f(acc[3][r]);
            c[r] = fg[r] * c[r] + ig[r] * gg[r];
            hn[r] = og[r] * tanh_f(c[r]);
        }
        bf16* hw = hbuf + (cur ^ 1) * 16 * ldh + l15 * ldh + j0;
        if (live) {
            const size_t o = ((size_t)t * B + b) * H + j0;
            *reinterpret_cast<f32x4*>(h_all + o) = hn;
            *reinterpret_cast<f32x4*>(c_all + o) = c;
            float* ap = acts + ((size_t)t * B + b) * 4 * H + j0;
            *reinterpret_cast<f32x4*>(ap) = ig; *reinterpret_cast<f32x4*>(ap + H) = fg;
            *reinterpret_cast<f32x4*>(ap + 2 * H) = gg; *reinterpret_cast<f32x4*>(ap + 3 * H) = og;
#pragma unroll
            for (int r = 0; r < 4; ++r) hw[r] = (bf16)hn[r];
        }
        __syncthreads();
        cur ^= 1;
    }
}

// Backward through time.  dG[t] (gate pre-activation gradients, fp32 (T,B,4H)) is also what the batched
// input-projection / weight gradients consume afterwards.  MAXKS >= KP4/32.
template <int MAXKS, int NT, bool WREG>
__global__ __launch_bounds__(NT) void lstm_scan_bwd_kernel(const float* __restrict__ dh_ext, const float* __restrict__ dc_ext,
                                     const bf16* __restrict__ Wb, const float* __restrict__ c0,
                                     const float* __restrict__ c_all, const float* __restrict__ acts,
                                     float* __restrict__ dG, float* __restrict__ dh0, float* __restrict__ dc0,
                                     int T, int B, int H, int HP16, int KP4) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ldg = KP4 + 8;
    bf16* gbuf = reinterpret_cast<bf16*>(smem);                 // [2][16][ldg]
    const int lane = threadIdx.x & 63, jt = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int nks = KP4 >> 5;
    const int b = blockIdx.x * 16 + l15, j0 = jt * 16 + 4 * lq;
    const bool live = (b < B) && (j0 < H);

    const bf16* wrow = Wb + (size_t)(jt * 16 + l15) * KP4 + 8 * lq;
    bf16x8 a[WREG ? MAXKS : 1];
    if (WREG) {
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks)
            if (ks < nks) a[ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
    }
    for (int i = threadIdx.x; i < 2 * 16 * ldg; i += blockDim.x) gbuf[i] = (bf16)0.f;
    __syncthreads();

    f32x4 dh_rec = {0.f, 0.f, 0.f, 0.f}, dc = {0.f, 0.f, 0.f, 0.f};
    int cur = 0;
    for (int t = T - 1; t >= 0; --t) {
        f32x4 dgi = {0.f, 0.f, 0.f, 0.f}, dgf = dgi, dgg = dgi, dgo = dgi;
        if (live) {
            const size_t o = ((size_t)t * B + b) * H + j0;
            const float* ap = acts + ((size_t)t * B + b) * 4 * H + j0;
            const f32x4 ig = *reinterpret_cast<const f32x4*>(ap), fg = *reinterpret_cast<const f32x4*>(ap + H);
            const f32x4 gg = *reinterpret_cast<const f32x4*>(ap + 2 * H), og = *reinterpret_cast<const f32x4*>(ap + 3 * H);
            const f32x4 ct = *reinterpret_cast<const f32x4*>(c_all + o);
            f32x4 cp = {0.f, 0.f, 0.f, 0.f};
            if (t > 0) cp = *reinterpret_cast<const f32x4*>(c_all + o - (size_t)B * H);
            else if (c0) cp = *reinterpret_cast<const f32x4*>(c0 + (size_t)b * H + j0);
            f32x4 dh = dh_rec;
            if (dh_ext) dh += *reinterpret_cast<const f32x4*>(dh_ext + o);
            if (dc_ext) dc += *reinterpret_cast<const f32x4*>(dc_ext + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float th = tanh_f(ct[r]);
                const float dct = dc[r] + dh[r] * og[r] * (1.f - th * th);
                dgo[r] = dh[r] * th * og[r] * (1.f - og[r]);
                dgi[r] = dct * gg[r] * ig[r] * (1.f - ig[r]);
                dgf[r] = dct * cp[r] * fg[r] * (1.f - fg[r]);
                dgg[r] = dct * ig[r] * (1.f - gg[r] * gg[r]);
                dc[r] = dct * fg[r];
            }
            float* gp = dG + ((size_t)t * B + b) * 4 * H + j0;
            *reinterpret_cast<f32x4*>(gp) = dgi; *reinterpret_cast<f32x4*>(gp + H) = dgf;
            *reinterpret_cast<f32x4*>(gp + 2 * H) = dgg; *reinterpret_cast<f32x4*>(gp + 3 * H) = dgo;
        }
        bf16* gw = gbuf + cur * 16 * ldg + l15 * ldg + j0;
        if (j0 < HP16) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                gw[r] = (bf16)dgi[r]; gw[HP16 + r] = (bf16)dgf[r]; gw[2 * HP16 + r] = (bf16)dgg[r]; gw[3 * HP16 + r] = (bf16)dgo[r];
            }
        }
        __syncthreads();
        const bf16* gb = gbuf + cur * 16 * ldg + l15 * ldg + 8 * lq;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < MAXKS; ++ks)
            if (ks < nks) {
                const bf16x8 af = WREG ? a[WREG ? ks : 0] : *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
                acc = mfma16(af, *reinterpret_cast<const bf16x8*>(gb + ks * 32), acc);
            }
        dh_rec = acc;
        cur ^= 1;
    }
    if (live) {
        if (dh0) *reinterpret_cast<f32x4*>(dh0 + (size_t)b * H + j0) = dh_rec;
        if (dc0) *reinterpret_cast<f32x4*>(dc0 + (size_t)b * H + j0) = dc;
    }
}


// ------------------------------------------------------------------------------------------------
// MFN delta-memory recurrence (transformer/MFT/multiTransformer.py:221-224), mem_dim = 128 (:133),
// two gate MLPs with 64 hidden units each (:140-141).  Per step, with apre[t] = fc1(attended part) + b
// batched beforehand for both gates (U = 128 rows: gamma1's 64 then gamma2's 64):
//     u    = relu(apre[t] + Wm mem)                Wm  (128 x 128) = memory columns of gamma{1,2}_fc1.weight
//     g1,2 = sigmoid(W2_{1,2} u_{1,2} + b2_{1,2})   W2  (2 x 128 x 64)
//     mem  = g1 * mem + g2 * chat[t]
// One workgroup = 16 sequences, 8 waves; wave w owns u rows / memory units [16w, 16w+16); weights are
// MFMA A fragments in registers for the whole scan; mem (bf16) and u (bf16) cross waves through LDS:
// two barriers per step.
#define MFN_MD 128
#define MFN_U 128
#define MFN_HG 64

// Wm (U,MD), W2 (2,MD,HG) fp32 -> bf16 operands: forward WmF [U][MD], W2F [2*MD][HG];
// backward WmB = Wm^T [MD][U], W2B [U][MD] with W2B[g*HG + k][j] = W2[g][j][k]
__global__ void mfn_prep_kernel(const float* __restrict__ Wm, const float* __restrict__ W2,
                                bf16* __restrict__ WmF, bf16* __restrict__ W2F, bf16* __restrict__ WmB, bf16* __restrict__ W2B) {
    const int n1 = MFN_U * MFN_MD, n2 = 2 * MFN_MD * MFN_HG;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < 2 * n1 + 2 * n2; idx += gridDim.x * blockDim.x) {
        if (idx < n1) WmF[idx] = (bf16)Wm[idx];
        else if (idx < n1 + n2) W2F[idx - n1] = (bf16)W2[idx - n1];
        else if (idx < 2 * n1 + n2) { const int i = idx - n1 - n2, j = i / MFN_U, u = i % MFN_U; WmB[i] = (bf16)Wm[u * MFN_MD + j]; }
        else { const int i = idx - 2 * n1 - n2, u = i / MFN_MD, j = i % MFN_MD, g = u / MFN_HG, k = u % MFN_HG;
               W2B[i] = (bf16)W2[((size_t)g * MFN_MD + j) * MFN_HG + k]; }
    }
}

__global__ __launch_bounds__(512) void mfn_mem_scan_fwd_kernel(
        const float* __restrict__ apre, const float* __restrict__ chat, const bf16* __restrict__ WmF,
        const bf16* __restrict__ W2F, const float* __restrict__ b2,
        float* __restrict__ mem_all, float* __restrict__ u_all, float* __restrict__ g_all, int T, int B, DropCfg drop) {
    __shared__ __attribute__((aligned(16))) bf16 membuf[16 * (MFN_MD + 8)];
    __shared__ __attribute__((aligned(16))) bf16 ubuf[16 * (MFN_U + 8)];
    constexpr int LDM = MFN_MD + 8, LDU = MFN_U + 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x * 16 + l15, j0 = w * 16 + 4 * lq;
    const bool live = b < B;
    bf16x8 am[4], a1[2], a2[2];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) am[ks] = *reinterpret_cast<const bf16x8*>(WmF + (size_t)(w * 16 + l15) * MFN_MD + ks * 32 + 8 * lq);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a1[ks] = *reinterpret_cast<const bf16x8*>(W2F + (size_t)(w * 16 + l15) * MFN_HG + ks * 32 + 8 * lq);
        a2[ks] = *reinterpret_cast<const bf16x8*>(W2F + (size_t)(MFN_MD + w * 16 + l15) * MFN_HG + ks * 32 + 8 * lq);
    }
    const f32x4 bias1 = *reinterpret_cast<const f32x4*>(b2 + j0), bias2 = *reinterpret_cast<const f32x4*>(b2 + MFN_MD + j0);
    for (int i = threadIdx.x; i < 16 * LDM; i += blockDim.x) membuf[i] = (bf16)0.f;
    for (int i = threadIdx.x; i < 16 * LDU; i += blockDim.x) ubuf[i] = (bf16)0.f;
    __syncthreads();
    f32x4 mem = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < T; ++t) {
        const size_t row = (size_t)t * B + b;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, ch = acc;
        if (live) { acc = *reinterpret_cast<const f32x4*>(apre + row * MFN_U + j0); ch = *reinterpret_cast<const f32x4*>(chat + row * MFN_MD + j0); }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            acc = mfma16(am[ks], *reinterpret_cast<const bf16x8*>(membuf + l15 * LDM + ks * 32 + 8 * lq), acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
        if (drop.thr16) {          // gamma{1,2}_dropout on relu(fc1) (reference :222-223); u_all keeps the dropped values
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const uint32_t wd = drop_pair(drop, (uint64_t)row * MFN_U + j0 + r);
                acc[r] = drop_lo(drop, wd, acc[r]); acc[r + 1] = drop_hi(drop, wd, acc[r + 1]);
            }
        }
        if (live) {
            *reinterpret_cast<f32x4*>(u_all + row * MFN_U + j0) = acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) ubuf[l15 * LDU + j0 + r] = (bf16)acc[r];
        }
        __syncthreads();
        f32x4 z1 = bias1, z2 = bias2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            z1 = mfma16(a1[ks], *reinterpret_cast<const bf16x8*>(ubuf + l15 * LDU + ks * 32 + 8 * lq), z1);
            z2 = mfma16(a2[ks], *reinterpret_cast<const bf16x8*>(ubuf + l15 * LDU + MFN_HG + ks * 32 + 8 * lq), z2);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) { z1[r] = sigmoid_f(z1[r]); z2[r] = sigmoid_f(z2[r]); mem[r] = z1[r] * mem[r] + z2[r] * ch[r]; }
        if (live) {
            *reinterpret_cast<f32x4*>(mem_all + row * MFN_MD + j0) = mem;
            *reinterpret_cast<f32x4*>(g_all + row * 2 * MFN_MD + j0) = z1;
            *reinterpret_cast<f32x4*>(g_all + row * 2 * MFN_MD + MFN_MD + j0) = z2;
#pragma unroll
            for (int r = 0; r < 4; ++r) membuf[l15 * LDM + j0 + r] = (bf16)mem[r];
        }
        __syncthreads();
    }
}

// Backward through time: emits dchat (T,B,MD), dapre (T,B,U) and dz (T,B,2MD) (pre-sigmoid gate gradients);
// the batched weight gradients (dWm = dapre^T mem_prev, dW2 = dz^T u, db2 = sum dz) are formed afterwards.
__global__ __launch_bounds__(512) void mfn_mem_scan_bwd_kernel(
        const float* __restrict__ dmem_ext, const float* __restrict__ chat, const float* __restrict__ mem_all,
        const float* __restrict__ u_all, const float* __restrict__ g_all, const bf16* __restrict__ WmB, const bf16* __restrict__ W2B,
        float* __restrict__ dchat, float* __restrict__ dapre, float* __restrict__ dz_all, int T, int B, float drop_scale) {
    __shared__ __attribute__((aligned(16))) bf16 zbuf[16 * (2 * MFN_MD + 8)];
    __shared__ __attribute__((aligned(16))) bf16 pbuf[16 * (MFN_U + 8)];
    constexpr int LDZ = 2 * MFN_MD + 8, LDP = MFN_U + 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x * 16 + l15, j0 = w * 16 + 4 * lq;
    const bool live = b < B;
    const int gsel = (w * 16) / MFN_HG;                         // which gate MLP this wave's u rows belong to
    bf16x8 a2[4], am[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a2[ks] = *reinterpret_cast<const bf16x8*>(W2B + (size_t)(w * 16 + l15) * MFN_MD + ks * 32 + 8 * lq);
        am[ks] = *reinterpret_cast<const bf16x8*>(WmB + (size_t)(w * 16 + l15) * MFN_U + ks * 32 + 8 * lq);
    }
    for (int i = threadIdx.x; i < 16 * LDZ; i += blockDim.x) zbuf[i] = (bf16)0.f;
    for (int i = threadIdx.x; i < 16 * LDP; i += blockDim.x) pbuf[i] = (bf16)0.f;
    __syncthreads();
    f32x4 dcarry = {0.f, 0.f, 0.f, 0.f};
    for (int t = T - 1; t >= 0; --t) {
        const size_t row = (size_t)t * B + b;
        f32x4 dz1 = {0.f, 0.f, 0.f, 0.f}, dz2 = dz1, uu = dz1, dmg = dz1;
        if (live) {
            f32x4 dm = dcarry;
            if (dmem_ext) dm += *reinterpret_cast<const f32x4*>(dmem_ext + row * MFN_MD + j0);
            const f32x4 g1 = *reinterpret_cast<const f32x4*>(g_all + row * 2 * MFN_MD + j0);
            const f32x4 g2 = *reinterpret_cast<const f32x4*>(g_all + row * 2 * MFN_MD + MFN_MD + j0);
            const f32x4 ch = *reinterpret_cast<const f32x4*>(chat + row * MFN_MD + j0);
            f32x4 mp = {0.f, 0.f, 0.f, 0.f};
            if (t > 0) mp = *reinterpret_cast<const f32x4*>(mem_all + (row - B) * MFN_MD + j0);
            f32x4 dch;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dz1[r] = dm[r] * mp[r] * g1[r] * (1.f - g1[r]);
                dz2[r] = dm[r] * ch[r] * g2[r] * (1.f - g2[r]);
                dch[r] = dm[r] * g2[r];
                dmg[r] = dm[r] * g1[r];
            }
            *reinterpret_cast<f32x4*>(dchat + row * MFN_MD + j0) = dch;
            *reinterpret_cast<f32x4*>(dz_all + row * 2 * MFN_MD + j0) = dz1;
            *reinterpret_cast<f32x4*>(dz_all + row * 2 * MFN_MD + MFN_MD + j0) = dz2;
            uu = *reinterpret_cast<const f32x4*>(u_all + row * MFN_U + j0);
#pragma unroll
            for (int r = 0; r < 4; ++r) { zbuf[l15 * LDZ + j0 + r] = (bf16)dz1[r]; zbuf[l15 * LDZ + MFN_MD + j0 + r] = (bf16)dz2[r]; }
        }
        __syncthreads();
        // du rows [16w,16w+16) = W2_g^T dz_g ;  dpre = du * relu'(u)
        f32x4 du = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            du = mfma16(a2[ks], *reinterpret_cast<const bf16x8*>(zbuf + l15 * LDZ + gsel * MFN_MD + ks * 32 + 8 * lq), du);
#pragma unroll
        for (int r = 0; r < 4; ++r) du[r] = (uu[r] > 0.f) ? du[r] * drop_scale : 0.f;   // u_all > 0 <=> relu passed AND kept
        if (live) {
            *reinterpret_cast<f32x4*>(dapre + row * MFN_U + j0) = du;
#pragma unroll
            for (int r = 0; r < 4; ++r) pbuf[l15 * LDP + j0 + r] = (bf16)du[r];
        }
        __syncthreads();
        f32x4 rec = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            rec = mfma16(am[ks], *reinterpret_cast<const bf16x8*>(pbuf + l15 * LDP + ks * 32 + 8 * lq), rec);
        dcarry = dmg + rec;
    }
}
