// Scaled-dot-product attention core for gfx950, forward and backward, over per-(batch,head)
// fragment-layout operands (common.h).  No (T x T) tensor is ever materialised
// (the reference does: transformer/MFT/multiTransformer.py:27-34).
//
// Mask semantics of the reference (query-ROW blanking, keys never masked, :29-31,48-50) are applied
// by the producers/consumers of these operands, not here: a blanked row arrives as Q' = 0, which
// soft-maxes to exactly 1/T over all T keys like the reference's constant -1e9 row, and its dQ is
// zeroed by dq_finish_kernel.  Only index masking lives here: keys/queries >= T in the last tile.
//
// Scores are kept in the log2 domain: the producer stores Q' = (x Wq^T + bq) * log2(e)/sqrt(d_k), so
// P = 2^(S' - L) with L = rowmax + log2(rowsum) saved per query for the backward pass.
#pragma once
#include "common.h"

// ------------------------------------------------------------------------------------------------
// All three kernels below are barrier-free and LDS-free: grid = (ceil(nt/4), B*h), 4 independent waves per
// workgroup, each owning one 32-window tile and sweeping the other axis, with the next tile's operand
// fragments already in flight (register prefetch) while the current one computes.  Register budgets stay
// near 100 VGPRs so 4-5 waves share a SIMD and hide each other's MFMA / exp latencies.

// per-(batch,head) dropout stream on the probabilities: 32-bit index q*Tp + key, one hash word per key pair
__device__ __forceinline__ void drop_probs_qlane(f32x16& v, const DropCfg& dc, uint32_t base, int hh) {
#pragma unroll
    for (int i = 0; i < 16; i += 2) {       // registers (i, i+1) hold adjacent keys
        const uint32_t w = drop_word(dc.s0, dc.s1, (base + (uint32_t)acc32_row(i, hh)) >> 1);
        v[i] = drop_lo(dc, w, v[i]); v[i + 1] = drop_hi(dc, w, v[i + 1]);
    }
}

// ------------------------------------------------------------------------------------------------
// Forward.  S^T = K Q'^T puts the query on the lane column: running max / sum / rescale are lane-local and
// the only cross-lane step is one exchange between the two 32-lane halves.  After the first key tile the
// running max m is preloaded as the accumulator (S' = S - m straight out of the MFMA) and the state is only
// rescaled when some row's tile maximum exceeds m by more than RESCALE_THR (then P <= 2^THR, harmless in
// bf16/fp32): the common path per score is max3 + exp + convert.  P^T (keys in registers) is fed straight
// back as the B operand of O^T += V^T P^T.  For DKP == 16 the idle half of that MFMA also forms the row sums
// (row 16 of V^T is synthesised as ones) unless dropout is on (the normaliser uses the undropped P).
#define MMT_RESCALE_THR 8.0f

template <int DKP, bool DROP>
__global__ __launch_bounds__(MMT_THREADS) void attn_fwd_kernel(
        const bf16* __restrict__ Qr, const bf16* __restrict__ Kr, const bf16* __restrict__ Vt,
        bf16* __restrict__ ctx, bf16* __restrict__ ctxT, float* __restrict__ lse,
        int h, int T, int nt, int ldc, int MP, DropCfg drop) {
    constexpr int KS = DKP / 16;
    constexpr bool ONES = (DKP == 16) && !DROP;        // row sums through the MFMA
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int qt = blockIdx.x * 4 + wave;
    if (qt >= nt) return;
    const int bh = blockIdx.y, b = bh / h, head = bh - b * h;
    const int Tp = nt * 32;
    const bf16* Qb = Qr + (size_t)bh * fragR_elems(Tp, DKP);
    const bf16* Kb = Kr + (size_t)bh * fragR_elems(Tp, DKP);
    const bf16* Vb = Vt + (size_t)bh * fragT_elems(Tp);
    DropCfg dc = drop;
    dc.s0 += (uint32_t)bh * 0x7F4A7C15u;

    bf16x8 qf[KS], kf[KS], vf[2], kn[KS], vn[2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        qf[s] = *reinterpret_cast<const bf16x8*>(Qb + ((size_t)(qt * (DKP / 8) + 2 * s + hh) * 32 + r) * 8);
        kf[s] = *reinterpret_cast<const bf16x8*>(Kb + ((size_t)(2 * s + hh) * 32 + r) * 8);
    }
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) vf[s2] = *reinterpret_cast<const bf16x8*>(Vb + ((size_t)(s2 * 2 + hh) * 32 + r) * 8);

    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float mrun = 0.f, lrun = 0.f;

    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) {
#pragma unroll
            for (int s = 0; s < KS; ++s)
                kn[s] = *reinterpret_cast<const bf16x8*>(Kb + ((size_t)((kt + 1) * (DKP / 8) + 2 * s + hh) * 32 + r) * 8);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                vn[s2] = *reinterpret_cast<const bf16x8*>(Vb + ((size_t)(((kt + 1) * 2 + s2) * 2 + hh) * 32 + r) * 8);
        }
        f32x16 s;
        const float init = (kt == 0) ? 0.f : -mrun;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = init;
#pragma unroll
        for (int ss = 0; ss < KS; ++ss) s = mfma32(kf[ss], qf[ss], s);
        if (kt == nt - 1 && (T & 31)) {                // keys >= T do not exist
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (kt * 32 + acc32_row(i, hh) >= T) s[i] = -INFINITY;
        }
        float tmax = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) tmax = fmaxf(fmaxf(tmax, s[i]), s[i + 1]);
        tmax = fmaxf(tmax, s[15]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));      // finite: every tile holds >= 1 real key in one of the halves
        if (kt == 0 || __any(tmax > MMT_RESCALE_THR)) {
            // move the reference to the new running max (first tile: from 0 to the tile max, with o = l = 0)
            const float dlt = (kt == 0) ? tmax : fmaxf(tmax, 0.f);
            const float alpha = fast_exp2(-dlt);
            mrun += dlt;
            lrun *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[i] *= alpha; s[i] -= dlt; }
        }
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = fast_exp2(s[i]); if (!ONES) psum += s[i]; }
        if (!ONES) lrun += psum;
        if (DROP) drop_probs_qlane(s, dc, (uint32_t)(qt * 32 + r) * (uint32_t)Tp + (uint32_t)(kt * 32), hh);
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 va = vf[s2];
            if (ONES && r == DKP) va = ones;           // V^T rows >= DKP are zero in memory; row DKP becomes the ones row
            o = mfma32(va, pack8(s, s2), o);
        }
        if (kt + 1 < nt) {
#pragma unroll
            for (int ss = 0; ss < KS; ++ss) kf[ss] = kn[ss];
            vf[0] = vn[0]; vf[1] = vn[1];
        }
    }
    float ltot;
    if (ONES) ltot = __shfl(o[8], r);                  // O^T row 16 = (register 8, lower half): the row sums
    else ltot = lrun + __shfl_xor(lrun, 32);
    const float inv = 1.0f / ltot;
    const int t = qt * 32 + r;
    if (t < T) {
        const size_t m = (size_t)b * T + t;
        if (hh == 0) lse[(size_t)bh * Tp + t] = mrun + fast_log2(ltot);
        // O^T rows (head features) live in registers: e = acc32_row(i, hh); groups of 4 are contiguous
#pragma unroll
        for (int g = 0; g < DKP / 8; ++g) {
            bf16x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (bf16)(o[4 * g + i] * inv);
            const int e0 = 8 * g + 4 * hh;
            *reinterpret_cast<bf16x4*>(ctx + m * ldc + head * DKP + e0) = v;
#pragma unroll
            for (int i = 0; i < 4; ++i) ctxT[(size_t)(head * DKP + e0 + i) * MP + m] = v[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward, part A: dK, dV.  Wave w owns key tile kt = 4*blockIdx.x + w (the key on the lane column) and
// sweeps all query tiles, keeping dK^T, dV^T in accumulators.  Per query tile:
//     S'  = Q' K^T - L      (L preloaded as the accumulator rows: P = 2^S' needs no subtraction)
//     dPc = dO V^T - delta  (same trick with delta = rowsum(dO . O))
//     dS  = P * dPc ;  dV^T += dO^T P ;  dK^T += Q'^T dS     (P, dS accumulators ARE the B operands)
template <int DKP, bool DROP>
__global__ __launch_bounds__(MMT_THREADS) void attn_bwd_dkv_kernel(
        const bf16* __restrict__ Qr, const bf16* __restrict__ Qt, const bf16* __restrict__ Kr, const bf16* __restrict__ Vr,
        const bf16* __restrict__ dOr, const bf16* __restrict__ dOt,
        const float* __restrict__ lse, const float* __restrict__ delta,
        bf16* __restrict__ dkv, int lddkv,      // row-major [M][lddkv]; dK at column HD, dV at 2*HD
        bf16* __restrict__ dkvT, int MP,        // T layout  [3*HD rows][MP]
        int h, int T, int nt, DropCfg drop) {
    constexpr int KS = DKP / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int kt = blockIdx.x * 4 + wave;
    if (kt >= nt) return;
    const int bh = blockIdx.y, b = bh / h, head = bh - b * h;
    const int Tp = nt * 32, HD = h * DKP;
    const size_t offR = (size_t)bh * fragR_elems(Tp, DKP), offT = (size_t)bh * fragT_elems(Tp);
    const bf16 *Qrb = Qr + offR, *Krb = Kr + offR, *Vrb = Vr + offR, *dOrb = dOr + offR;
    const bf16 *Qtb = Qt + offT, *dOtb = dOt + offT;
    const float* lseb = lse + (size_t)bh * Tp;
    const float* delb = delta + (size_t)bh * Tp;
    DropCfg dc = drop;
    dc.s0 += (uint32_t)bh * 0x7F4A7C15u;

    bf16x8 kfr[KS], vfr[KS];
#pragma unroll
    for (int ss = 0; ss < KS; ++ss) {
        const size_t off = ((size_t)(kt * (DKP / 8) + 2 * ss + hh) * 32 + r) * 8;
        kfr[ss] = *reinterpret_cast<const bf16x8*>(Krb + off);
        vfr[ss] = *reinterpret_cast<const bf16x8*>(Vrb + off);
    }
    f32x16 dKacc, dVacc;
#pragma unroll
    for (int j = 0; j < 16; ++j) { dKacc[j] = 0.f; dVacc[j] = 0.f; }
    const bool key_tail = (kt == nt - 1) && (T & 31);
    const bool key_ok = (kt * 32 + r) < T;
    const uint32_t kcol = (uint32_t)(kt * 32 + r);

    for (int qt = 0; qt < nt; ++qt) {
        bf16x8 qa[KS], da[KS], qT[2], dT[2];
#pragma unroll
        for (int ss = 0; ss < KS; ++ss) {
            const size_t off = ((size_t)(qt * (DKP / 8) + 2 * ss + hh) * 32 + r) * 8;
            qa[ss] = *reinterpret_cast<const bf16x8*>(Qrb + off);
            da[ss] = *reinterpret_cast<const bf16x8*>(dOrb + off);
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const size_t off = ((size_t)((qt * 2 + s2) * 2 + hh) * 32 + r) * 8;
            qT[s2] = *reinterpret_cast<const bf16x8*>(Qtb + off);
            dT[s2] = *reinterpret_cast<const bf16x8*>(dOtb + off);
        }
        f32x16 s, dp;                   // row constants (4 consecutive queries per register group) as the accumulators
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lseb + qt * 32 + 8 * g + 4 * hh);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(delb + qt * 32 + 8 * g + 4 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) { s[4 * g + i] = -l4[i]; dp[4 * g + i] = -d4[i]; }
        }
        f32x16 negD;
        if (DROP) negD = dp;
#pragma unroll
        for (int ss = 0; ss < KS; ++ss) { s = mfma32(qa[ss], kfr[ss], s); dp = mfma32(da[ss], vfr[ss], dp); }
#pragma unroll
        for (int j = 0; j < 16; ++j) s[j] = fast_exp2(s[j]);
        if (key_tail) {                                 // keys >= T do not exist
#pragma unroll
            for (int j = 0; j < 16; ++j) s[j] = key_ok ? s[j] : 0.f;
        }
        if (qt == nt - 1 && (T & 31)) {                 // neither do queries >= T
#pragma unroll
            for (int j = 0; j < 16; ++j) if (qt * 32 + acc32_row(j, hh) >= T) s[j] = 0.f;
        }
        if (DROP) {
            // dropped probabilities Pd = P*m/(1-p):  dV^T += dO^T Pd ;  dS = P * ((dO V^T)*m/(1-p) - delta); delta unchanged
            const uint32_t q0 = (uint32_t)(qt * 32) * (uint32_t)Tp + kcol;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t idx = q0 + (uint32_t)acc32_row(j, hh) * (uint32_t)Tp;
                const uint32_t w = drop_word(dc.s0, dc.s1, idx >> 1);
                const float ms = (((idx & 1) ? (w >> 16) : (w & 0xFFFFu)) >= dc.thr16) ? dc.scale : 0.f;
                dp[j] = s[j] * ((dp[j] - negD[j]) * ms + negD[j]);
                s[j] *= ms;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) dp[j] *= s[j];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            dVacc = mfma32(dT[s2], pack8(s, s2), dVacc);
            dKacc = mfma32(qT[s2], pack8(dp, s2), dKacc);
        }
    }
    // dK = ln2 * acc (scores are in the log2 domain), dV = acc; rows e = acc32_row, column key = r
    const float LN2 = 0.6931471805599453f;
    const int t = kt * 32 + r;
    if (t < T) {
        const size_t m = (size_t)b * T + t;
#pragma unroll
        for (int g = 0; g < DKP / 8; ++g) {
            bf16x4 kv, vv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { kv[j] = (bf16)(dKacc[4 * g + j] * LN2); vv[j] = (bf16)dVacc[4 * g + j]; }
            const int e0 = head * DKP + 8 * g + 4 * hh;
            *reinterpret_cast<bf16x4*>(dkv + m * lddkv + HD + e0) = kv;
            *reinterpret_cast<bf16x4*>(dkv + m * lddkv + 2 * HD + e0) = vv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dkvT[(size_t)(HD + e0 + j) * MP + m] = kv[j];
                dkvT[(size_t)(2 * HD + e0 + j) * MP + m] = vv[j];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward, part B: dQ.  Same orientation as the forward (query on the lane): wave w owns query tile
// qt = 4*blockIdx.x + w and sweeps all key tiles; L and delta are lane constants; the dS^T accumulator
// (keys in registers) is the B operand of dQ^T += K^T dS^T.  The epilogue applies 1/sqrt(d_k) and the
// query-row mask (blanked rows pass no gradient to Q) and writes columns [0,HD) of dQKV in both layouts.
template <int DKP, bool DROP>
__global__ __launch_bounds__(MMT_THREADS) void attn_bwd_dq_kernel(
        const bf16* __restrict__ Qr, const bf16* __restrict__ Kr, const bf16* __restrict__ Kt, const bf16* __restrict__ Vr,
        const bf16* __restrict__ dOr, const float* __restrict__ lse, const float* __restrict__ delta,
        const float* __restrict__ rowmask, float scale,
        bf16* __restrict__ dqkv, int lddqkv, bf16* __restrict__ dqkvT, int MP,
        int h, int T, int nt, DropCfg drop) {
    constexpr int KS = DKP / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int qt = blockIdx.x * 4 + wave;
    if (qt >= nt) return;
    const int bh = blockIdx.y, b = bh / h, head = bh - b * h;
    const int Tp = nt * 32;
    const size_t offR = (size_t)bh * fragR_elems(Tp, DKP), offT = (size_t)bh * fragT_elems(Tp);
    const bf16 *Qrb = Qr + offR, *Krb = Kr + offR, *Vrb = Vr + offR, *dOrb = dOr + offR, *Ktb = Kt + offT;
    DropCfg dc = drop;
    dc.s0 += (uint32_t)bh * 0x7F4A7C15u;

    bf16x8 qf[KS], dof[KS], kf[KS], vf[KS], ktf[2], kn[KS], vn[KS], ktn[2];
#pragma unroll
    for (int ss = 0; ss < KS; ++ss) {
        const size_t off = ((size_t)(qt * (DKP / 8) + 2 * ss + hh) * 32 + r) * 8;
        qf[ss] = *reinterpret_cast<const bf16x8*>(Qrb + off);
        dof[ss] = *reinterpret_cast<const bf16x8*>(dOrb + off);
        const size_t off0 = ((size_t)(2 * ss + hh) * 32 + r) * 8;
        kf[ss] = *reinterpret_cast<const bf16x8*>(Krb + off0);
        vf[ss] = *reinterpret_cast<const bf16x8*>(Vrb + off0);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) ktf[s2] = *reinterpret_cast<const bf16x8*>(Ktb + ((size_t)(s2 * 2 + hh) * 32 + r) * 8);
    const float negL = -lse[(size_t)bh * Tp + qt * 32 + r];
    const float negD = -delta[(size_t)bh * Tp + qt * 32 + r];
    f32x16 dq;
#pragma unroll
    for (int j = 0; j < 16; ++j) dq[j] = 0.f;

    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) {
#pragma unroll
            for (int ss = 0; ss < KS; ++ss) {
                const size_t off = ((size_t)((kt + 1) * (DKP / 8) + 2 * ss + hh) * 32 + r) * 8;
                kn[ss] = *reinterpret_cast<const bf16x8*>(Krb + off);
                vn[ss] = *reinterpret_cast<const bf16x8*>(Vrb + off);
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
                ktn[s2] = *reinterpret_cast<const bf16x8*>(Ktb + ((size_t)(((kt + 1) * 2 + s2) * 2 + hh) * 32 + r) * 8);
        }
        f32x16 s, dp;
#pragma unroll
        for (int j = 0; j < 16; ++j) { s[j] = negL; dp[j] = negD; }
#pragma unroll
        for (int ss = 0; ss < KS; ++ss) { s = mfma32(kf[ss], qf[ss], s); dp = mfma32(vf[ss], dof[ss], dp); }
#pragma unroll
        for (int j = 0; j < 16; ++j) s[j] = fast_exp2(s[j]);
        if (kt == nt - 1 && (T & 31)) {                 // keys >= T do not exist
#pragma unroll
            for (int j = 0; j < 16; ++j) if (kt * 32 + acc32_row(j, hh) >= T) s[j] = 0.f;
        }
        if (DROP) {
            const uint32_t base = (uint32_t)(qt * 32 + r) * (uint32_t)Tp + (uint32_t)(kt * 32);
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const uint32_t w = drop_word(dc.s0, dc.s1, (base + (uint32_t)acc32_row(j, hh)) >> 1);
                const float m0 = ((w & 0xFFFFu) >= dc.thr16) ? dc.scale : 0.f, m1 = ((w >> 16) >= dc.thr16) ? dc.scale : 0.f;
                dp[j] = s[j] * ((dp[j] - negD) * m0 + negD);
                dp[j + 1] = s[j + 1] * ((dp[j + 1] - negD) * m1 + negD);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) dp[j] *= s[j];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) dq = mfma32(ktf[s2], pack8(dp, s2), dq);
        if (kt + 1 < nt) {
#pragma unroll
            for (int ss = 0; ss < KS; ++ss) { kf[ss] = kn[ss]; vf[ss] = vn[ss]; }
            ktf[0] = ktn[0]; ktf[1] = ktn[1];
        }
    }
    const int t = qt * 32 + r;
    if (t < T) {
        const size_t m = (size_t)b * T + t;
        const float sc = (rowmask && rowmask[m] == 0.0f) ? 0.f : scale;     // blanked query rows pass no gradient to Q
#pragma unroll
        for (int g = 0; g < DKP / 8; ++g) {
            bf16x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (bf16)(dq[4 * g + j] * sc);
            const int e0 = head * DKP + 8 * g + 4 * hh;
            *reinterpret_cast<bf16x4*>(dqkv + m * lddqkv + e0) = v;
#pragma unroll
            for (int j = 0; j < 4; ++j) dqkvT[(size_t)(e0 + j) * MP + m] = v[j];
        }
    }
}
