// T-sequential recurrences of the path, as persistent-state scan kernels for gfx950.
//
// LSTM scan (MFN's per-modality nn.LSTMCell, transformer/MFT/multiTransformer.py:152,208, and the SFT
// decoder's nn.LSTM step, transformer/SFT/multiTransformer.py:471-476).  Only gates += W_rec . h_{t-1} is
// recurrent; the input projection gx[t] = x_t W_ih^T + b_ih + b_hh is batched over all T by the row GEMM.
// One workgroup owns BT <= 16 sequences (columns of the MFMA N dimension; BT is chosen so that the batch
// spreads over up to 256 CUs — the scan is latency-bound, idle MFMA columns cost nothing) for the whole scan;
// wave w owns hidden units [16w, 16w+16) and keeps its slice of W_rec as MFMA A fragments in registers for
// all T steps, the cell state in registers (fp32), and h crosses waves through a double-buffered bf16 LDS
// tile: one barrier per step.  Gate order i, f, g, o (torch).  H <= 128 here; larger hidden sizes:
// scan_cluster.h (four CUs per sequence) and scan256.h (half-resident weights).
#pragma once
#include "common.h"

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + fast_exp2(-1.4426950408889634f * x)); }
__device__ __forceinline__ float tanh_f(float x) { return 2.0f * sigmoid_f(2.0f * x) - 1.0f; }
// store to a wave-uniform base plus a 32-bit BYTE offset per lane as ONE instruction (`global_store_dword voff, vdata, s[base]`).
// hipcc forms the 64-bit address in vector registers instead (a v_lshl_add_u64 per store: 6 of a forward step's ~120 instructions, and
// a wave of these scans issues roughly one instruction per 10 cycles).  The stored values are many instructions old (the stores sit
// behind the step's barrier), so no hazard the assembler statement would hide from the compiler applies.
__device__ __forceinline__ void st_uniform(float* base, unsigned byte_off, float v) {
    asm volatile("global_store_dword %0, %1, %2" :: "v"(byte_off), "v"(v), "s"(base) : "memory");
}

// W_rec (4H,H) fp32 -> forward operand Wf bf16 [4][HP16][HPAD] (Wf[q][j][k] = W[q*H+j][k]) and
// backward operand Wb bf16 [HP16][4*HPAD] (Wb[j][q*HPAD+j'] = W[q*H+j'][j]); zero padded.
// HPAD = hidden size padded to 64 / 128 / 256: every loop bound in the scans is then a compile-time constant.
__global__ void lstm_prep_kernel(const float* __restrict__ W, bf16* __restrict__ Wf, bf16* __restrict__ Wb,
                                 int H, int HP16, int HPAD) {
    const size_t nf = (size_t)4 * HP16 * HPAD, nb = (size_t)HP16 * 4 * HPAD;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nf + nb; idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < nf) {
            const int k = (int)(idx % HPAD), j = (int)((idx / HPAD) % HP16), q = (int)(idx / ((size_t)HPAD * HP16));
            Wf[idx] = (bf16)((j < H && k < H) ? W[((size_t)q * H + j) * H + k] : 0.f);
        } else {
            const size_t i = idx - nf;
            const int c = (int)(i % (4 * HPAD)), j = (int)(i / (4 * HPAD));
            const int q = c / HPAD, jp = c - q * HPAD;
            Wb[i] = (bf16)((j < H && jp < H) ? W[((size_t)q * H + jp) * H + j] : 0.f);
        }
    }
}

// grid = ceil(B/BT); block = 64 * (HP16/16) <= NT.  HPAD = 32*KS.  WREG: W_rec fragments stay in registers for the
// whole scan (HPAD <= 128); otherwise (HPAD = 256: 512 KB of bf16 weights exceed one CU's register file) they are
// re-streamed from L2 every step.  The time loop is straight-line code: lanes outside the batch / hidden range
// compute on clamped addresses and simply do not store.  This is the form for 4..16 sequences per workgroup (batches above 512);
// one or two sequences per workgroup run the kernels of scan_units.h.
template <int KS, int NT, bool WREG, int PF>
__global__ __launch_bounds__(NT) void lstm_scan_fwd_kernel(const float* __restrict__ gx, const bf16* __restrict__ Wf,
                                     const float* __restrict__ h0, const float* __restrict__ c0,
                                     float* __restrict__ h_all, float* __restrict__ c_all, float* __restrict__ acts,
                                     int T, int B, int H, int HP16, int BT) {
    constexpr int KP = 32 * KS, ldh = KP + 8;
    __shared__ __attribute__((aligned(16))) bf16 hbuf[2 * 16 * ldh];
    const int lane = threadIdx.x & 63, jt = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    // BT <= 16 sequences per workgroup: the scan is latency-bound and its per-step traffic must not funnel through a
    // couple of CUs, so small batches are spread over many workgroups and the unused MFMA columns simply idle.
    const int b = blockIdx.x * BT + l15, j0 = jt * 16 + 4 * lq;
    const bool live = (l15 < BT) && (b < B) && (j0 < H);        // H % 4 == 0: the 4 rows of a lane are all in or all out
    const int bc = b < B ? b : B - 1, jc = j0 < H ? j0 : H - 4; // clamped (always valid) coordinates for loads

    const bf16* wrow = Wf + (size_t)(jt * 16 + l15) * KP + 8 * lq;      // + q*HP16*KP + ks*32
    const size_t wq = (size_t)HP16 * KP;
    bf16x8 a[WREG ? 4 : 1][WREG ? KS : 1];
    if (WREG) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a[q][ks] = *reinterpret_cast<const bf16x8*>(wrow + q * wq + ks * 32);
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

    const size_t gstep = (size_t)B * 4 * H;
    // input projections of the next PF steps are always in flight (register ring, statically indexed by unrolling)
    f32x4 ring[PF][4];
    const float* gxl = gx + (size_t)bc * 4 * H + jc;            // + t*gstep + q*H
    auto fetch = [&](f32x4 (&r)[4], int t) {
        const int tl = t < T ? t : T - 1;                       // clamped: the tail re-reads the last step (unused)
#pragma unroll
        for (int q = 0; q < 4; ++q) r[q] = *reinterpret_cast<const f32x4*>(gxl + tl * gstep + (size_t)q * H);
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(ring[d], d);
    int cur = 0;
    auto step = [&](int t, f32x4 (&in)[4]) {
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = in[q];
        fetch(in, t + PF);
        const bf16* hb = hbuf + cur * 16 * ldh + l15 * ldh + 8 * lq;
        if (WREG) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(hb + ks * 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma16(a[WREG ? q : 0][WREG ? ks : 0], bf, acc[q]);
            }
        } else {
            // streamed weights: two k-blocks of fragments in flight; the scheduling fences keep hipcc from hoisting
            // all 4*KS fragment loads to the top of the step (128 VGPRs at KS = 8: spills)
            bf16x8 wa[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) wa[0][q] = *reinterpret_cast<const bf16x8*>(wrow + q * wq);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) wa[(ks + 1) & 1][q] = *reinterpret_cast<const bf16x8*>(wrow + q * wq + (ks + 1) * 32);
                }
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(hb + ks * 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma16(wa[ks & 1][q], bf, acc[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        f32x4 ig, fg, gg, og, hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ig[r] = sigmoid_f(acc[0][r]); fg[r] = sigmoid_f(acc[1][r]); gg[r] = tanh_f(acc[2][r]); og[r] = sigmoid_f(acc[3][r]);
            c[r] = fg[r] * c[r] + ig[r] * gg[r];
            hn[r] = live ? og[r] * tanh_f(c[r]) : 0.f;      // pad lanes keep h = 0 in LDS
        }
        bf16x4 hb4;
#pragma unroll
        for (int r = 0; r < 4; ++r) hb4[r] = (bf16)hn[r];
        *reinterpret_cast<bf16x4*>(hbuf + (cur ^ 1) * 16 * ldh + l15 * ldh + jt * 16 + 4 * lq) = hb4;
        lds_barrier();                                      // h_t visible to every wave; global traffic stays in flight
        if (live) {
            const size_t o = ((size_t)t * B + b) * H + j0;
            *reinterpret_cast<f32x4*>(h_all + o) = hn;
            *reinterpret_cast<f32x4*>(c_all + o) = c;
            float* ap = acts + ((size_t)t * B + b) * 4 * H + j0;
            *reinterpret_cast<f32x4*>(ap) = ig; *reinterpret_cast<f32x4*>(ap + H) = fg;
            *reinterpret_cast<f32x4*>(ap + 2 * H) = gg; *reinterpret_cast<f32x4*>(ap + 3 * H) = og;
        }
        cur ^= 1;
    };
    int t0 = 0;
    for (; t0 + PF <= T; t0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) step(t0 + d, ring[d]);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d) if (t0 + d < T) step(t0 + d, ring[d]);
}

// Backward through time.  dG[t] (gate pre-activation gradients, fp32 (T,B,4H)) is also what the batched
// input-projection / weight gradients consume afterwards.  KS4 = 4*HPAD/32.
template <int KS4, int NT, bool WREG, int PF>
__global__ __launch_bounds__(NT) void lstm_scan_bwd_kernel(const float* __restrict__ dh_ext, const float* __restrict__ dc_ext,
                                     const bf16* __restrict__ Wb, const float* __restrict__ c0,
                                     const float* __restrict__ c_all, const float* __restrict__ acts,
                                     float* __restrict__ dG, float* __restrict__ dh0, float* __restrict__ dc0,
                                     int T, int B, int H, int HP16, int BT) {
    constexpr int KP4 = 32 * KS4, HPAD = KP4 / 4, ldg = KP4 + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* gbuf = reinterpret_cast<bf16*>(smem);                 // [2][16][ldg]
    const int lane = threadIdx.x & 63, jt = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x * BT + l15, j0 = jt * 16 + 4 * lq;
    const bool live = (l15 < BT) && (b < B) && (j0 < H);
    const int bc = b < B ? b : B - 1, jc = j0 < H ? j0 : H - 4;
    const float lv = live ? 1.f : 0.f;

    const bf16* wrow = Wb + (size_t)(jt * 16 + l15) * KP4 + 8 * lq;
    bf16x8 a[WREG ? KS4 : 1];
    if (WREG) {
#pragma unroll
        for (int ks = 0; ks < KS4; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
    }
    for (int i = threadIdx.x; i < 2 * 16 * ldg; i += blockDim.x) gbuf[i] = (bf16)0.f;
    __syncthreads();

    f32x4 dh_rec = {0.f, 0.f, 0.f, 0.f}, dc = {0.f, 0.f, 0.f, 0.f};
    // saved activations / cell states / external gradients of the next PF steps (going backwards) stay in flight
    struct StepIn { f32x4 ig, fg, gg, og, ct, cp, dhe, dce; };
    const size_t ostep = (size_t)B * H;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    StepIn ring[PF];
    const float* actl = acts + (size_t)bc * 4 * H + jc;
    const float* cl = c_all + (size_t)bc * H + jc;
    f32x4 c0v = zero4;
    if (c0) c0v = *reinterpret_cast<const f32x4*>(c0 + (size_t)bc * H + jc);
    auto fetch = [&](StepIn& r, int t) {
        const int tc = t > 0 ? t : 0;                           // clamped, branch-free
        const float* ap = actl + (size_t)tc * ostep * 4;
        r.ig = *reinterpret_cast<const f32x4*>(ap); r.fg = *reinterpret_cast<const f32x4*>(ap + H);
        r.gg = *reinterpret_cast<const f32x4*>(ap + 2 * H); r.og = *reinterpret_cast<const f32x4*>(ap + 3 * H);
        r.ct = *reinterpret_cast<const f32x4*>(cl + (size_t)tc * ostep);
        r.cp = *reinterpret_cast<const f32x4*>(cl + (size_t)(tc > 0 ? tc - 1 : 0) * ostep);
        if (t <= 0) r.cp = c0v;
        r.dhe = dh_ext ? *reinterpret_cast<const f32x4*>(dh_ext + (size_t)bc * H + jc + (size_t)tc * ostep) : zero4;
        r.dce = dc_ext ? *reinterpret_cast<const f32x4*>(dc_ext + (size_t)bc * H + jc + (size_t)tc * ostep) : zero4;
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(ring[d], T - 1 - d);
    int cur = 0;
    auto step = [&](int t, StepIn& slot) {
        const StepIn in = slot;
        fetch(slot, t - PF);
        f32x4 dgi, dgf, dgg, dgo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float dh = dh_rec[r] + in.dhe[r];
            const float th = tanh_f(in.ct[r]);
            const float dct = dc[r] + in.dce[r] + dh * in.og[r] * (1.f - th * th);
            dgo[r] = lv * dh * th * in.og[r] * (1.f - in.og[r]);
            dgi[r] = lv * dct * in.gg[r] * in.ig[r] * (1.f - in.ig[r]);
            dgf[r] = lv * dct * in.cp[r] * in.fg[r] * (1.f - in.fg[r]);
            dgg[r] = lv * dct * in.ig[r] * (1.f - in.gg[r] * in.gg[r]);
            dc[r] = dct * in.fg[r];
        }
        bf16* gw = gbuf + cur * 16 * ldg + l15 * ldg + jt * 16 + 4 * lq;
        bf16x4 p0, p1, p2, p3;
#pragma unroll
        for (int r = 0; r < 4; ++r) { p0[r] = (bf16)dgi[r]; p1[r] = (bf16)dgf[r]; p2[r] = (bf16)dgg[r]; p3[r] = (bf16)dgo[r]; }
        *reinterpret_cast<bf16x4*>(gw) = p0; *reinterpret_cast<bf16x4*>(gw + HPAD) = p1;
        *reinterpret_cast<bf16x4*>(gw + 2 * HPAD) = p2; *reinterpret_cast<bf16x4*>(gw + 3 * HPAD) = p3;
        lds_barrier();
        if (live) {
            float* gp = dG + ((size_t)t * B + b) * 4 * H + j0;
            *reinterpret_cast<f32x4*>(gp) = dgi; *reinterpret_cast<f32x4*>(gp + H) = dgf;
            *reinterpret_cast<f32x4*>(gp + 2 * H) = dgg; *reinterpret_cast<f32x4*>(gp + 3 * H) = dgo;
        }
        const bf16* gb = gbuf + cur * 16 * ldg + l15 * ldg + 8 * lq;
        // four independent accumulation chains (one per gate block of the contraction): a single chain of KS4
        // dependent MFMAs would serialise on the accumulator latency
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // the B operand (all 4H gate gradients of a sequence) is the LDS-bandwidth term of a step: 16 KB per wave
        // at H = 128.  Only the BT live MFMA columns are read; the other lanes keep whatever their registers
        // held (a column of D depends on the same column of B only, and dead columns are never stored).
        constexpr int KG = WREG ? KS4 : 8;                      // B fragments fetched per group (register budget)
#pragma unroll
        for (int k0 = 0; k0 < KS4; k0 += KG) {
            bf16x8 bfr[KG];
            if (l15 < BT) {
#pragma unroll
                for (int ks = 0; ks < KG; ++ks) bfr[ks] = *reinterpret_cast<const bf16x8*>(gb + (k0 + ks) * 32);
            }
#pragma unroll
            for (int ks = 0; ks < KG; ++ks) {
                const bf16x8 af = WREG ? a[WREG ? k0 + ks : 0] : *reinterpret_cast<const bf16x8*>(wrow + (k0 + ks) * 32);
                acc[ks & 3] = mfma16(af, bfr[ks], acc[ks & 3]);
            }
            if (!WREG) __builtin_amdgcn_sched_barrier(0);       // keep the next group's fragment loads below this point
        }
        dh_rec = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        cur ^= 1;
    };
    int tb = T - 1;
    for (; tb - PF + 1 >= 0; tb -= PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) step(tb - d, ring[d]);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d) if (tb - d >= 0) step(tb - d, ring[d]);
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
        float* __restrict__ mem_all, float* __restrict__ u_all, float* __restrict__ g_all, int T, int B, int BT, DropCfg drop_in,
        const uint64_t* __restrict__ seedword) {
    const DropCfg drop = drop_resolve(drop_in, seedword);
    __shared__ __attribute__((aligned(16))) bf16 membuf[16 * (MFN_MD + 8)];
    __shared__ __attribute__((aligned(16))) bf16 ubuf[16 * (MFN_U + 8)];
    constexpr int LDM = MFN_MD + 8, LDU = MFN_U + 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x * BT + l15, j0 = w * 16 + 4 * lq;
    const bool live = (l15 < BT) && (b < B);
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
    // the step inputs of the next PF steps stay in flight (register ring, statically indexed by unrolling): a load issued at
    // the top of the step it feeds costs the step a full memory round trip (~1 us of the former 1.1 us per step)
    constexpr int PF = 4;
    const int bc = b < B ? b : B - 1;
    struct In { f32x4 a, c; };
    In ring[PF];
    auto fetch = [&](In& r, int t) {
        const size_t rw = (size_t)(t < T ? t : T - 1) * B + bc;
        r.a = *reinterpret_cast<const f32x4*>(apre + rw * MFN_U + j0);
        r.c = *reinterpret_cast<const f32x4*>(chat + rw * MFN_MD + j0);
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(ring[d], d);
    auto step = [&](int t, In& slot) {
        const size_t row = (size_t)t * B + b;
        f32x4 acc = slot.a;
        const f32x4 ch = slot.c;
        fetch(slot, t + PF);
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
        lds_barrier();
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
        lds_barrier();
    };
    int t0 = 0;
    for (; t0 + PF <= T; t0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) step(t0 + d, ring[d]);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d) if (t0 + d < T) step(t0 + d, ring[d]);
}

// Backward through time: emits dchat (T,B,MD), dapre (T,B,U) and dz (T,B,2MD) (pre-sigmoid gate gradients);
// the batched weight gradients (dWm = dapre^T mem_prev, dW2 = dz^T u, db2 = sum dz) are formed afterwards.
__global__ __launch_bounds__(512) void mfn_mem_scan_bwd_kernel(
        const float* __restrict__ dmem_ext, const float* __restrict__ chat, const float* __restrict__ mem_all,
        const float* __restrict__ u_all, const float* __restrict__ g_all, const bf16* __restrict__ WmB, const bf16* __restrict__ W2B,
        float* __restrict__ dchat, float* __restrict__ dapre, float* __restrict__ dz_all, int T, int B, int BT, float drop_scale) {
    __shared__ __attribute__((aligned(16))) bf16 zbuf[16 * (2 * MFN_MD + 8)];
    __shared__ __attribute__((aligned(16))) bf16 pbuf[16 * (MFN_U + 8)];
    constexpr int LDZ = 2 * MFN_MD + 8, LDP = MFN_U + 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x * BT + l15, j0 = w * 16 + 4 * lq;
    const bool live = (l15 < BT) && (b < B);
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
    constexpr int PF = 2;                                       // saved tensors of the next PF steps (going backwards) in flight
    const int bc = b < B ? b : B - 1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    struct In { f32x4 dme, g1, g2, ch, mp, uu; };
    In ring[PF];
    auto fetch = [&](In& r, int t) {
        const int tc = t > 0 ? t : 0;
        const size_t rw = (size_t)tc * B + bc;
        r.dme = dmem_ext ? *reinterpret_cast<const f32x4*>(dmem_ext + rw * MFN_MD + j0) : zero4;
        r.g1 = *reinterpret_cast<const f32x4*>(g_all + rw * 2 * MFN_MD + j0);
        r.g2 = *reinterpret_cast<const f32x4*>(g_all + rw * 2 * MFN_MD + MFN_MD + j0);
        r.ch = *reinterpret_cast<const f32x4*>(chat + rw * MFN_MD + j0);
        r.mp = *reinterpret_cast<const f32x4*>(mem_all + (rw - (tc > 0 ? (size_t)B : 0)) * MFN_MD + j0);
        if (t <= 0) r.mp = zero4;                               // mem_{-1} = 0
        r.uu = *reinterpret_cast<const f32x4*>(u_all + rw * MFN_U + j0);
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(ring[d], T - 1 - d);
    auto step = [&](int t, In& slot) {
        const size_t row = (size_t)t * B + b;
        const In in = slot;
        fetch(slot, t - PF);
        f32x4 dz1, dz2, dmg, dch;
        const f32x4 dm = dcarry + in.dme;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dz1[r] = dm[r] * in.mp[r] * in.g1[r] * (1.f - in.g1[r]);
            dz2[r] = dm[r] * in.ch[r] * in.g2[r] * (1.f - in.g2[r]);
            dch[r] = dm[r] * in.g2[r];
            dmg[r] = dm[r] * in.g1[r];
        }
        if (live) {
            *reinterpret_cast<f32x4*>(dchat + row * MFN_MD + j0) = dch;
            *reinterpret_cast<f32x4*>(dz_all + row * 2 * MFN_MD + j0) = dz1;
            *reinterpret_cast<f32x4*>(dz_all + row * 2 * MFN_MD + MFN_MD + j0) = dz2;
#pragma unroll
            for (int r = 0; r < 4; ++r) { zbuf[l15 * LDZ + j0 + r] = (bf16)dz1[r]; zbuf[l15 * LDZ + MFN_MD + j0 + r] = (bf16)dz2[r]; }
        }
        lds_barrier();
        // du rows [16w,16w+16) = W2_g^T dz_g ;  dpre = du * relu'(u)
        f32x4 du = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            du = mfma16(a2[ks], *reinterpret_cast<const bf16x8*>(zbuf + l15 * LDZ + gsel * MFN_MD + ks * 32 + 8 * lq), du);
#pragma unroll
        for (int r = 0; r < 4; ++r) du[r] = (in.uu[r] > 0.f) ? du[r] * drop_scale : 0.f;   // u_all > 0 <=> relu passed AND kept
        if (live) {
            *reinterpret_cast<f32x4*>(dapre + row * MFN_U + j0) = du;
#pragma unroll
            for (int r = 0; r < 4; ++r) pbuf[l15 * LDP + j0 + r] = (bf16)du[r];
        }
        lds_barrier();
        f32x4 rec = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            rec = mfma16(am[ks], *reinterpret_cast<const bf16x8*>(pbuf + l15 * LDP + ks * 32 + 8 * lq), rec);
        dcarry = dmg + rec;
    };
    int tb = T - 1;
    for (; tb - PF + 1 >= 0; tb -= PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) step(tb - d, ring[d]);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d) if (tb - d >= 0) step(tb - d, ring[d]);
}

// ---- the memory recurrence for ONE or TWO sequences per workgroup (batches of up to 512 sequences: 256 workgroups are in flight)
// Same mathematics and the same dropout counters as the kernels above, with the MFMA operands swapped as in the cooperative LSTM
// scans: the products are (state row) x W^T, so sequence r is register r of lanes 0..15 and every such lane owns ONE unit
// j = 16 w + l15 of the wave (memory unit and row of u alike).  With the units in the registers and the sequence on the lane
// (the general form above) four lanes of a wave carried four units each: every sigmoid, hash, pack and address of a step was
// issued four times over for four live lanes.  Loads and stores are scalar per lane: 16 lanes x 4 bytes are one 64-byte segment.
template <int NR>
__global__ __launch_bounds__(512) void mfn_mem_scan_fwd_sw_kernel(
        const float* __restrict__ apre, const float* __restrict__ chat, const bf16* __restrict__ WmF,
        const bf16* __restrict__ W2F, const float* __restrict__ b2,
        float* __restrict__ mem_all, float* __restrict__ u_all, float* __restrict__ g_all, int T, int B, DropCfg drop_in,
        const uint64_t* __restrict__ seedword) {
    const DropCfg drop = drop_resolve(drop_in, seedword);
    __shared__ __attribute__((aligned(16))) bf16 membuf[16 * (MFN_MD + 8)];
    __shared__ __attribute__((aligned(16))) bf16 ubuf[16 * (MFN_U + 8)];
    constexpr int LDM = MFN_MD + 8, LDU = MFN_U + 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * NR, j = w * 16 + l15;
    const int nb = (B - b0) < NR ? (B - b0) : NR;
    const bool own = lq == 0;
    bf16x8 am[4], a1[2], a2[2];                              // B fragments: column = this lane's unit, 8 consecutive k per lane quarter
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) am[ks] = *reinterpret_cast<const bf16x8*>(WmF + (size_t)j * MFN_MD + ks * 32 + 8 * lq);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        a1[ks] = *reinterpret_cast<const bf16x8*>(W2F + (size_t)j * MFN_HG + ks * 32 + 8 * lq);
        a2[ks] = *reinterpret_cast<const bf16x8*>(W2F + (size_t)(MFN_MD + j) * MFN_HG + ks * 32 + 8 * lq);
    }
    const float bias1 = b2[j], bias2 = b2[MFN_MD + j];
    for (int i = threadIdx.x; i < 16 * LDM; i += blockDim.x) membuf[i] = (bf16)0.f;
    for (int i = threadIdx.x; i < 16 * LDU; i += blockDim.x) ubuf[i] = (bf16)0.f;
    __syncthreads();
    float mem[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) mem[r] = 0.f;
    constexpr int PF = 4;
    struct In { float a[NR], c[NR]; };
    In ring[PF];
    auto fetch = [&](In& q, int t) {
        const size_t rw = (size_t)(t < T ? t : T - 1) * B + b0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const size_t rr = rw + (r < nb ? r : 0);
            q.a[r] = apre[rr * MFN_U + j];
            q.c[r] = chat[rr * MFN_MD + j];
        }
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(ring[d], d);
    const unsigned jo = 4u * (unsigned)j;
    auto step = [&](int t, In& slot) {
        const size_t row0 = (size_t)t * B + b0;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float ch[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) { acc[r] = slot.a[r]; ch[r] = slot.c[r]; }
        fetch(slot, t + PF);
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f};                  // two accumulation chains
#pragma unroll
        for (int ks = 0; ks < 4; ks += 2) {
            acc = mfma16(*reinterpret_cast<const bf16x8*>(membuf + l15 * LDM + ks * 32 + 8 * lq), am[ks], acc);
            acc2 = mfma16(*reinterpret_cast<const bf16x8*>(membuf + l15 * LDM + (ks + 1) * 32 + 8 * lq), am[ks + 1], acc2);
        }
        float u[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            u[r] = fmaxf(acc[r] + acc2[r], 0.f);
            if (drop.thr16) {      // gamma{1,2}_dropout on relu(fc1) (reference :222-223); u_all keeps the dropped values.  Same pair words as above
                const uint32_t wd = drop_pair(drop, (uint64_t)(row0 + r) * MFN_U + (j & ~1));
                u[r] = (j & 1) ? drop_hi(drop, wd, u[r]) : drop_lo(drop, wd, u[r]);
            }
            if (own && r < nb) ubuf[r * LDU + j] = (bf16)u[r];
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < NR; ++r) if (own && r < nb) st_uniform(u_all + (row0 + r) * MFN_U, jo, u[r]);
        f32x4 z1 = {bias1, bias1, bias1, bias1}, z2 = {bias2, bias2, bias2, bias2};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            z1 = mfma16(*reinterpret_cast<const bf16x8*>(ubuf + l15 * LDU + ks * 32 + 8 * lq), a1[ks], z1);
            z2 = mfma16(*reinterpret_cast<const bf16x8*>(ubuf + l15 * LDU + MFN_HG + ks * 32 + 8 * lq), a2[ks], z2);
        }
        float g1[NR], g2[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            g1[r] = sigmoid_f(z1[r]); g2[r] = sigmoid_f(z2[r]);
            mem[r] = g1[r] * mem[r] + g2[r] * ch[r];
            if (own && r < nb) membuf[r * LDM + j] = (bf16)mem[r];
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (own && r < nb) {
                st_uniform(mem_all + (row0 + r) * MFN_MD, jo, mem[r]);
                float* gp = g_all + (row0 + r) * 2 * MFN_MD;
                st_uniform(gp, jo, g1[r]); st_uniform(gp + MFN_MD, jo, g2[r]);
            }
    };
    int t0 = 0;
    for (; t0 + PF <= T; t0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) step(t0 + d, ring[d]);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d) if (t0 + d < T) step(t0 + d, ring[d]);
}

template <int NR>
__global__ __launch_bounds__(512) void mfn_mem_scan_bwd_sw_kernel(
        const float* __restrict__ dmem_ext, const float* __restrict__ chat, const float* __restrict__ mem_all,
        const float* __restrict__ u_all, const float* __restrict__ g_all, const bf16* __restrict__ WmB, const bf16* __restrict__ W2B,
        float* __restrict__ dchat, float* __restrict__ dapre, float* __restrict__ dz_all, int T, int B, float drop_scale) {
    __shared__ __attribute__((aligned(16))) bf16 zbuf[16 * (2 * MFN_MD + 8)];
    __shared__ __attribute__((aligned(16))) bf16 pbuf[16 * (MFN_U + 8)];
    constexpr int LDZ = 2 * MFN_MD + 8, LDP = MFN_U + 8;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b0 = blockIdx.x * NR, j = w * 16 + l15;
    const int nb = (B - b0) < NR ? (B - b0) : NR;
    const bool own = lq == 0;
    const int gsel = (w * 16) / MFN_HG;                         // which gate MLP this wave's u rows belong to
    bf16x8 a2[4], am[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        a2[ks] = *reinterpret_cast<const bf16x8*>(W2B + (size_t)j * MFN_MD + ks * 32 + 8 * lq);
        am[ks] = *reinterpret_cast<const bf16x8*>(WmB + (size_t)j * MFN_U + ks * 32 + 8 * lq);
    }
    for (int i = threadIdx.x; i < 16 * LDZ; i += blockDim.x) zbuf[i] = (bf16)0.f;
    for (int i = threadIdx.x; i < 16 * LDP; i += blockDim.x) pbuf[i] = (bf16)0.f;
    __syncthreads();
    float dcarry[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) dcarry[r] = 0.f;
    constexpr int PF = 3;
    struct In { float dme[NR], g1[NR], g2[NR], ch[NR], mp[NR], uu[NR]; };
    In ring[PF];
    auto fetch = [&](In& q, int t) {
        const int tc = t > 0 ? t : 0;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const size_t rw = (size_t)tc * B + b0 + (r < nb ? r : 0);
            q.dme[r] = dmem_ext ? dmem_ext[rw * MFN_MD + j] : 0.f;
            q.g1[r] = g_all[rw * 2 * MFN_MD + j];
            q.g2[r] = g_all[rw * 2 * MFN_MD + MFN_MD + j];
            q.ch[r] = chat[rw * MFN_MD + j];
            q.mp[r] = mem_all[(rw - (tc > 0 ? (size_t)B : 0)) * MFN_MD + j];
            if (t <= 0) q.mp[r] = 0.f;                          // mem_{-1} = 0
            q.uu[r] = u_all[rw * MFN_U + j];
        }
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(ring[d], T - 1 - d);
    const unsigned jo = 4u * (unsigned)j;
    auto step = [&](int t, In& slot) {
        const size_t row0 = (size_t)t * B + b0;
        const In in = slot;
        fetch(slot, t - PF);
        float dz1[NR], dz2[NR], dmg[NR], dch[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float dm = dcarry[r] + in.dme[r];
            dz1[r] = dm * in.mp[r] * in.g1[r] * (1.f - in.g1[r]);
            dz2[r] = dm * in.ch[r] * in.g2[r] * (1.f - in.g2[r]);
            dch[r] = dm * in.g2[r];
            dmg[r] = dm * in.g1[r];
            if (own && r < nb) { zbuf[r * LDZ + j] = (bf16)dz1[r]; zbuf[r * LDZ + MFN_MD + j] = (bf16)dz2[r]; }
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (own && r < nb) {
                st_uniform(dchat + (row0 + r) * MFN_MD, jo, dch[r]);
                float* zp = dz_all + (row0 + r) * 2 * MFN_MD;
                st_uniform(zp, jo, dz1[r]); st_uniform(zp + MFN_MD, jo, dz2[r]);
            }
        // du of unit j = (W2_g^T dz_g)[j] ;  dpre = du * relu'(u)
        f32x4 dua = {0.f, 0.f, 0.f, 0.f}, dub = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ks += 2) {
            dua = mfma16(*reinterpret_cast<const bf16x8*>(zbuf + l15 * LDZ + gsel * MFN_MD + ks * 32 + 8 * lq), a2[ks], dua);
            dub = mfma16(*reinterpret_cast<const bf16x8*>(zbuf + l15 * LDZ + gsel * MFN_MD + (ks + 1) * 32 + 8 * lq), a2[ks + 1], dub);
        }
        float du[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            du[r] = (in.uu[r] > 0.f) ? (dua[r] + dub[r]) * drop_scale : 0.f;      // u_all > 0 <=> relu passed AND kept
            if (own && r < nb) pbuf[r * LDP + j] = (bf16)du[r];
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < NR; ++r) if (own && r < nb) st_uniform(dapre + (row0 + r) * MFN_U, jo, du[r]);
        f32x4 ra = {0.f, 0.f, 0.f, 0.f}, rb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ks += 2) {
            ra = mfma16(*reinterpret_cast<const bf16x8*>(pbuf + l15 * LDP + ks * 32 + 8 * lq), am[ks], ra);
            rb = mfma16(*reinterpret_cast<const bf16x8*>(pbuf + l15 * LDP + (ks + 1) * 32 + 8 * lq), am[ks + 1], rb);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) dcarry[r] = dmg[r] + (ra[r] + rb[r]);
    };
    int tb = T - 1;
    for (; tb - PF + 1 >= 0; tb -= PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) step(tb - d, ring[d]);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d) if (tb - d >= 0) step(tb - d, ring[d]);
}
