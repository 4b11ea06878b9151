// LSTM scans for ONE or TWO sequences per workgroup (batches of up to 512 sequences: every sequence — or pair — has a CU of its own).
//
// These workgroups run two waves per SIMD, and a wave in that regime issues roughly one instruction per 10 cycles whatever the
// instruction is (tools/valu_micro.hip): a time step costs what its instruction stream costs.  Two things follow.
//
// Units on the lanes.  The general kernels (scan.h) put W_rec in the MFMA A operand: the accumulator then holds 4 hidden units per lane
// and the sequence in the lane index, so with one live sequence four lanes of a wave carry 16 pre-activations each — either ten
// transcendentals per unit are issued for four useful lanes, or the accumulators go through an LDS patch to be re-dealt one unit per
// lane (round 2: a write, a wait and four reads on every step's critical path).  Here the operands are swapped: the product is
// h W^T (A = the state rows, B = the wave's weight rows — the SAME register contents as before), so sequence r is register r of lanes
// 0..15 and each of those lanes owns ONE hidden unit `ud`; gate math, cell state, LDS exchange and global accesses all follow that.
//
// Inputs straight into registers.  A lane needs 4 (forward) or 8 (backward) floats per step and sequence, so the step inputs of the next
// PF steps simply stay in flight in a register ring per lane (24 / 48 VGPRs at PF = 6).  (With four units per lane that ring was
// 16-32 VGPRs per step, which is why round 2 fetched the inputs cooperatively through an LDS ring: a global load, an LDS write, four to
// eight LDS reads and a page of scalar address arithmetic per step.)  16 lanes x 4 bytes are one 64-byte segment per gate and wave.
//
// Stores go out as `global_store_dword voff, vdata, s[base]` (scan.h st_uniform): one instruction each.
// Gate order i, f, g, o (torch).  Reference: nn.LSTMCell loop transformer/MFT/multiTransformer.py:200-208, nn.LSTM step loop
// transformer/SFT/multiTransformer.py:471-476.
#pragma once
#include "scan.h"

// grid = ceil(B / NR); block = 64 * (HP16/16) <= NT.  HPAD = 32*KS.  WREG: W_rec fragments in registers for the whole scan (HPAD <= 128),
// else re-streamed from L2 every step.
template <int KS, int NT, bool WREG, int PF, int NR>
__global__ __launch_bounds__(NT) void lstm_scan_fwd_u_kernel(const float* __restrict__ gx, const bf16* __restrict__ Wf,
                                     const float* __restrict__ h0, const float* __restrict__ c0,
                                     float* __restrict__ h_all, float* __restrict__ c_all, float* __restrict__ acts,
                                     int T, int B, int H, int HP16) {
    constexpr int KP = 32 * KS, ldh = KP + 8;
    __shared__ __attribute__((aligned(16))) bf16 hbuf[2 * 16 * ldh];            // [2][16 rows: sequence r in row r, the others zero][ldh]
    const int lane = threadIdx.x & 63, jt = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int bd0 = blockIdx.x * NR;
    const int nb = (B - bd0) < NR ? (B - bd0) : NR;
    const int ud = jt * 16 + l15, udc = ud < H ? ud : H - 1;
    const bool ulive = (lq == 0) && (ud < H);
    const unsigned uo[4] = {4u * (unsigned)ud, 4u * (unsigned)(ud + H), 4u * (unsigned)(ud + 2 * H), 4u * (unsigned)(ud + 3 * H)};   // byte offsets

    // B fragments: column = unit jt*16 + l15, 8 consecutive k per lane quarter
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
    float cd[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        cd[r] = 0.f;
        if (ulive && r < nb) {
            if (c0) cd[r] = c0[(size_t)(bd0 + r) * H + ud];
            if (h0) hbuf[r * ldh + ud] = (bf16)h0[(size_t)(bd0 + r) * H + ud];
        }
    }
    __syncthreads();

    // this lane's gate inputs of the next PF steps (the four lane quarters ask for the same addresses: one access)
    const size_t gstep = (size_t)B * 4 * H;
    const float* gxl = gx + (size_t)bd0 * 4 * H + udc;
    struct In { float g[NR][4]; };
    In ring[PF];
    // (uniform offsets are carried from step to step — an add and a select — instead of being formed as 64-bit products of the step index)
    size_t foff = 0;                                            // float offset of the next step to fetch; stops at the last step
    int tf = 0;
    const size_t r1 = (size_t)(nb > 1 ? 1 : 0) * 4 * H;
    auto fetch = [&](In& q) {
        const float* p = gxl + foff;
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int g = 0; g < 4; ++g) q.g[r][g] = p[(r ? r1 : 0) + (size_t)g * H];
        foff += (tf < T - 1) ? gstep : 0;                       // the tail re-reads the last step (unused)
        ++tf;
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(ring[d]);
    size_t soff = (size_t)bd0 * H;                              // float offset of (step t, sequence bd0) in h_all / c_all
    int cur = 0;
    auto step = [&](int t, In& slot) {
        const In in = slot;
        fetch(slot);
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16* hb = hbuf + cur * 16 * ldh + l15 * ldh + 8 * lq;    // A fragments: row l15 = sequence l15
        if (WREG) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 bf = *reinterpret_cast<const bf16x8*>(hb + ks * 32);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = mfma16(bf, a[WREG ? q : 0][WREG ? ks : 0], acc[q]);
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
                for (int q = 0; q < 4; ++q) acc[q] = mfma16(bf, wa[ks & 1][q], acc[q]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float ig[NR], fg[NR], gg[NR], og[NR], hn[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            ig[r] = sigmoid_f(acc[0][r] + in.g[r][0]); fg[r] = sigmoid_f(acc[1][r] + in.g[r][1]);
            gg[r] = tanh_f(acc[2][r] + in.g[r][2]); og[r] = sigmoid_f(acc[3][r] + in.g[r][3]);
            cd[r] = fg[r] * cd[r] + ig[r] * gg[r];
            hn[r] = og[r] * tanh_f(cd[r]);
            // rows >= nb and units >= H of the h tile stay 0 (never written); lanes lq != 0 hold products of those zero rows
            if (ulive && r < nb) hbuf[(cur ^ 1) * 16 * ldh + r * ldh + ud] = (bf16)hn[r];
        }
        lds_barrier();                                      // h_t visible to every wave; global traffic stays in flight
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (ulive && r < nb) {
                const size_t o = soff + (size_t)r * H;
                st_uniform(h_all + o, uo[0], hn[r]);
                st_uniform(c_all + o, uo[0], cd[r]);
                float* ap = acts + o * 4;
                st_uniform(ap, uo[0], ig[r]); st_uniform(ap, uo[1], fg[r]); st_uniform(ap, uo[2], gg[r]); st_uniform(ap, uo[3], og[r]);
            }
        soff += (size_t)B * H;
        (void)t;
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

// Backward through time.  dG[t] (gate pre-activation gradients, fp32 (T,B,4H)) is also what the batched input-projection / weight
// gradients consume afterwards.  KS4 = 4*HPAD/32.  Dynamic LDS: 2 * 16 * (32*KS4 + 8) bf16 (the gate-gradient tile, double buffered).
// Everything of a step that does not depend on the recurrence (the derivative factors of the saved activations, one tanh) is computed
// a step AHEAD, under the MFMAs of the step before: six multiply-adds stay on the dh -> gate gradients -> dh chain.
template <int KS4, int NT, bool WREG, int PF, int NR>
__global__ __launch_bounds__(NT) void lstm_scan_bwd_u_kernel(const float* __restrict__ dh_ext, const float* __restrict__ dc_ext,
                                     const bf16* __restrict__ Wb, const float* __restrict__ c0,
                                     const float* __restrict__ c_all, const float* __restrict__ acts,
                                     float* __restrict__ dG, float* __restrict__ dh0, float* __restrict__ dc0,
                                     int T, int B, int H, int HP16) {
    constexpr int KP4 = 32 * KS4, HPAD = KP4 / 4, ldg = KP4 + 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* gbuf = reinterpret_cast<bf16*>(smem);                 // [2][16 rows: sequence r in row r][ldg]: k = gate*HPAD + unit
    const int lane = threadIdx.x & 63, jt = threadIdx.x >> 6, l15 = lane & 15, lq = lane >> 4;
    const int bd0 = blockIdx.x * NR;
    const int nb = (B - bd0) < NR ? (B - bd0) : NR;
    const int ud = jt * 16 + l15, udc = ud < H ? ud : H - 1;
    const bool ulive = (lq == 0) && (ud < H);
    const unsigned uo[4] = {4u * (unsigned)ud, 4u * (unsigned)(ud + H), 4u * (unsigned)(ud + 2 * H), 4u * (unsigned)(ud + 3 * H)};   // byte offsets

    const bf16* wrow = Wb + (size_t)(jt * 16 + l15) * KP4 + 8 * lq;     // B fragments: column = unit jt*16 + l15 of dh
    bf16x8 a[WREG ? KS4 : 1];
    if (WREG) {
#pragma unroll
        for (int ks = 0; ks < KS4; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(wrow + ks * 32);
    }
    for (int i = threadIdx.x; i < 2 * 16 * ldg; i += blockDim.x) gbuf[i] = (bf16)0.f;
    __syncthreads();

    // saved activations / cell states / external gradients of this lane's unit for the next PF steps (going backwards)
    struct In { float ig[NR], fg[NR], gg[NR], og[NR], ct[NR], cp[NR], dhe[NR], dce[NR]; };
    const size_t ostep = (size_t)B * H;
    In ring[PF];
    // (uniform offsets are carried from step to step — a subtract and a select — instead of being formed as 64-bit products of the
    // step index: that was half of the scalar instructions of a step)
    // Absent optional inputs (external gradients, c0) read a valid dummy and are scaled by 0: no branches in the time loop.
    int tf = T - 1;                                             // the step the next fetch is for; < 0: re-reads step 0 (unused)
    size_t foff = (size_t)(T - 1) * ostep;                      // tc * ostep, tc = max(tf, 0)
    const float* const dhp = dh_ext ? dh_ext : c_all;
    const float* const dcp = dc_ext ? dc_ext : c_all;
    const float dhs = dh_ext ? 1.f : 0.f, dcs = dc_ext ? 1.f : 0.f;
    size_t bo[NR];
    float c0v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        bo[r] = (size_t)(bd0 + (r < nb ? r : 0)) * H + udc;
        c0v[r] = c0 ? c0[bo[r]] : 0.f;                          // c_{-1} = c0
    }
    auto fetch = [&](In& q) {
        const size_t fprev = foff - (tf > 0 ? ostep : 0);       // (tc - 1) * ostep, clamped
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float* ap = acts + foff * 4 + bo[r] * 4 - 3 * (size_t)udc;      // (foff + seq*H)*4 + udc
            q.ig[r] = ap[0]; q.fg[r] = ap[H]; q.gg[r] = ap[2 * H]; q.og[r] = ap[3 * H];
            q.ct[r] = c_all[foff + bo[r]];
            const float cprev = c_all[fprev + bo[r]];
            q.cp[r] = (tf > 0) ? cprev : c0v[r];
            q.dhe[r] = dhp[foff + bo[r]];
            q.dce[r] = dcp[foff + bo[r]];
        }
        foff = fprev;
        --tf;
    };
    struct Coef { float a, b, ci, cf, cg, f, dhe, dce; };      // dct = dc + dce + dh a;  dgo = dh b;  dgi/dgf/dgg = dct ci/cf/cg;  dc' = dct f
    Coef cf[NR];
    auto coefs = [&](const In& q) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float th = tanh_f(q.ct[r]);
            cf[r].a = q.og[r] * (1.f - th * th); cf[r].b = th * q.og[r] * (1.f - q.og[r]);
            cf[r].ci = q.gg[r] * q.ig[r] * (1.f - q.ig[r]); cf[r].cf = q.cp[r] * q.fg[r] * (1.f - q.fg[r]);
            cf[r].cg = q.ig[r] * (1.f - q.gg[r] * q.gg[r]);
            cf[r].f = q.fg[r]; cf[r].dhe = q.dhe[r] * dhs; cf[r].dce = q.dce[r] * dcs;
        }
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) fetch(ring[d]);
    coefs(ring[0]);
    size_t goff = ((size_t)(T - 1) * B + bd0) * 4 * H;          // float offset of (step t, sequence bd0) in dG
    float dhd[NR], dcd[NR];                                     // dh_rec / dc of the lane's unit
#pragma unroll
    for (int r = 0; r < NR; ++r) { dhd[r] = 0.f; dcd[r] = 0.f; }
    int cur = 0;
    // slot: the ring entry of step t (its factors are in cf already: free to refill); next: the entry of step t - 1
    auto step = [&](int t, In& slot, const In& next) {
        fetch(slot);
        float dgi[NR], dgf[NR], dgg[NR], dgo[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const float dh = dhd[r] + cf[r].dhe;
            const float dct = dcd[r] + cf[r].dce + dh * cf[r].a;
            dgo[r] = dh * cf[r].b;
            dgi[r] = dct * cf[r].ci;
            dgf[r] = dct * cf[r].cf;
            dgg[r] = dct * cf[r].cg;
            dcd[r] = dct * cf[r].f;
            // rows >= nb and pad units of the gradient tile stay 0 (never written)
            if (ulive && r < nb) {
                bf16* gw = gbuf + cur * 16 * ldg + r * ldg + ud;
                gw[0] = (bf16)dgi[r]; gw[HPAD] = (bf16)dgf[r]; gw[2 * HPAD] = (bf16)dgg[r]; gw[3 * HPAD] = (bf16)dgo[r];
            }
        }
        lds_barrier();
#pragma unroll
        for (int r = 0; r < NR; ++r)
            if (ulive && r < nb) {
                float* gp = dG + goff + (size_t)r * 4 * H;
                st_uniform(gp, uo[0], dgi[r]); st_uniform(gp, uo[1], dgf[r]); st_uniform(gp, uo[2], dgg[r]); st_uniform(gp, uo[3], dgo[r]);
            }
        const bf16* gb = gbuf + cur * 16 * ldg + l15 * ldg + 8 * lq;   // A fragments: row l15 = sequence l15 (only rows < NR are read)
        // four independent accumulation chains: a single chain of KS4 dependent MFMAs would serialise on the accumulator latency
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        constexpr int KG = WREG ? KS4 : 8;                      // A fragments fetched per group (register budget)
#pragma unroll
        for (int k0 = 0; k0 < KS4; k0 += KG) {
            bf16x8 afr[KG];
            if (l15 < NR) {                                     // the other rows keep whatever the registers held: row m of D depends on row m of A only
#pragma unroll
                for (int ks = 0; ks < KG; ++ks) afr[ks] = *reinterpret_cast<const bf16x8*>(gb + (k0 + ks) * 32);
            }
#pragma unroll
            for (int ks = 0; ks < KG; ++ks) {
                const bf16x8 wf = WREG ? a[WREG ? k0 + ks : 0] : *reinterpret_cast<const bf16x8*>(wrow + (k0 + ks) * 32);
                acc[ks & 3] = mfma16(afr[ks], wf, acc[ks & 3]);
            }
            if (!WREG) __builtin_amdgcn_sched_barrier(0);       // keep the next group's fragment loads below this point
        }
        coefs(next);                                            // step t-1's factors, under the MFMAs
        const f32x4 dh_rec = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
        for (int r = 0; r < NR; ++r) dhd[r] = dh_rec[r];       // sequence r, this lane's unit (lanes lq == 0)
        goff -= (size_t)B * 4 * H;
        (void)t;
        cur ^= 1;
    };
    int tb = T - 1;
    for (; tb - PF + 1 >= 0; tb -= PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) step(tb - d, ring[d], ring[(d + 1) % PF]);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d) if (tb - d >= 0) step(tb - d, ring[d], ring[(d + 1) % PF]);
#pragma unroll
    for (int r = 0; r < NR; ++r)
        if (ulive && r < nb) {
            if (dh0) dh0[(size_t)(bd0 + r) * H + ud] = dhd[r];
            if (dc0) dc0[(size_t)(bd0 + r) * H + ud] = dcd[r];
        }
}
