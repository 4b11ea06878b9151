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
// Forward.  grid = (ceil(nt/4), B*h), 4 waves; wave w owns query tile qt = 4*blockIdx.x + w (32 queries,
// one per lane column: S^T = K Q'^T puts the query on the lane, so running max / sum / rescale are
// lane-local and the only cross-lane step is one exchange between the two 32-lane halves).
// P^T (accumulator: keys in registers) is fed straight back as the B operand of O^T += V^T P^T.
template <int DKP>
__global__ __launch_bounds__(MMT_THREADS) void attn_fwd_kernel(
        const bf16* __restrict__ Qr, const bf16* __restrict__ Kr, const bf16* __restrict__ Vt,
        bf16* __restrict__ ctx, bf16* __restrict__ ctxT, float* __restrict__ lse,
        int h, int T, int nt, int ldc, int MP, DropCfg drop) {
    constexpr int KS = DKP / 16;                       // k-steps over the head feature
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int qt = blockIdx.x * 4 + wave;
    if (qt >= nt) return;                              // no barriers in this kernel
    const int bh = blockIdx.y, b = bh / h, head = bh - b * h;
    const int Tp = nt * 32;
    const bf16* Qb = Qr + (size_t)bh * fragR_elems(Tp, DKP);
    const bf16* Kb = Kr + (size_t)bh * fragR_elems(Tp, DKP);
    const bf16* Vb = Vt + (size_t)bh * fragT_elems(Tp);

    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        qf[s] = *reinterpret_cast<const bf16x8*>(Qb + ((size_t)(qt * (DKP / 8) + 2 * s + hh) * 32 + r) * 8);

    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float mrun = -INFINITY, lrun = 0.f;

    for (int kt = 0; kt < nt; ++kt) {
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int ss = 0; ss < KS; ++ss) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kb + ((size_t)(kt * (DKP / 8) + 2 * ss + hh) * 32 + r) * 8);
            s = mfma32(kf, qf[ss], s);
        }
        if (kt == nt - 1 && (T & 31)) {                // keys >= T do not exist
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (kt * 32 + acc32_row(i, hh) >= T) s[i] = -INFINITY;
        }
        float tmax = s[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mnew = fmaxf(mrun, tmax);          // finite: every tile holds >= 1 real key for hh = 0 or 1
        const float alpha = fast_exp2(mrun - mnew);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = fast_exp2(s[i] - mnew); psum += s[i]; }
        lrun = lrun * alpha + psum;
        mrun = mnew;
        if (drop.thr16) {      // dropout on the probabilities (reference :32-33); the normaliser keeps the undropped sum
            const uint64_t base = ((uint64_t)bh * Tp + (uint64_t)(qt * 32 + r)) * Tp + (uint64_t)kt * 32;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {     // keys of registers (i, i+1) are adjacent: one hash word per pair
                const uint32_t w = drop_pair(drop, base + acc32_row(i, hh));
                s[i] = drop_lo(drop, w, s[i]); s[i + 1] = drop_hi(drop, w, s[i + 1]);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] *= alpha;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 vf = *reinterpret_cast<const bf16x8*>(Vb + ((size_t)((kt * 2 + s2) * 2 + hh) * 32 + r) * 8);
            o = mfma32(vf, pack8(s, s2), o);
        }
    }
    const float ltot = lrun + __shfl_xor(lrun, 32);
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
// Backward.  grid = (nkb, B*h), 4 waves; wave w owns KT key tiles (32 keys each, the key on the lane
// column), so a workgroup covers 128*KT keys and keeps dK^T, dV^T for them in accumulators while it
// sweeps all query tiles: no cross-workgroup sum for dK/dV.  Per (query tile, key tile):
//     S'  = Q' K^T - L      (L preloaded as the accumulator: P = 2^S' needs no subtraction)
//     dPc = dO V^T - delta  (same trick with delta = rowsum(dO . O))
//     dS  = P * dPc ;  dV^T += dO^T P ;  dK^T += Q'^T dS     (P, dS accumulators are the B operands)
// dS crosses LDS once (bf16, already in MFMA k-order) and ONE wave per query tile (rotating) forms
// dQ = dS K for all the workgroup's keys, so dQ needs no cross-wave sum either; across key blocks it is
// written to per-block fp32 slabs [nkb][M][HDP] that dq_finish_kernel adds (deterministic, no atomics).
template <int DKP, int KT>
__global__ __launch_bounds__(MMT_THREADS) void attn_bwd_kernel(
        const bf16* __restrict__ Qr, const bf16* __restrict__ Qt,
        const bf16* __restrict__ Kr, const bf16* __restrict__ Kt,
        const bf16* __restrict__ Vr,
        const bf16* __restrict__ dOr, const bf16* __restrict__ dOt,
        const float* __restrict__ lse, const float* __restrict__ delta,
        float* __restrict__ dq_slab,            // [nkb][M][ldq]
        bf16* __restrict__ dkv, int lddkv,      // row-major [M][lddkv]; dK at column HD, dV at 2*HD
        bf16* __restrict__ dkvT, int MP,        // T layout  [3*HD rows][MP]
        int h, int T, int nt, int M, int ldq, DropCfg drop) {
    constexpr int KS = DKP / 16;
    constexpr int KB_TILES = 4 * KT;                   // key tiles per workgroup
    constexpr int LDS_ROW = KB_TILES * 32 + 8;         // bf16 elements per dS row (+8: bank spread)
    __shared__ __attribute__((aligned(16))) bf16 dS_lds[2][32 * LDS_ROW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 31, hh = lane >> 5;
    const int bh = blockIdx.y, b = bh / h, head = bh - b * h;
    const int kb = blockIdx.x;
    const int Tp = nt * 32, HD = h * DKP;
    const size_t offR = (size_t)bh * fragR_elems(Tp, DKP), offT = (size_t)bh * fragT_elems(Tp);
    const bf16 *Qrb = Qr + offR, *Krb = Kr + offR, *Vrb = Vr + offR, *dOrb = dOr + offR;
    const bf16 *Qtb = Qt + offT, *Ktb = Kt + offT, *dOtb = dOt + offT;
    const float* lseb = lse + (size_t)bh * Tp;
    const float* delb = delta + (size_t)bh * Tp;

    // this wave's key tiles; a tile index >= nt is an empty tile (no keys): its P is forced to 0
    int ktile[KT];
    bf16x8 kfr[KT][KS], vfr[KT][KS];
    f32x16 dKacc[KT], dVacc[KT];
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        ktile[i] = (kb * 4 + wave) * KT + i;
        const int kt = ktile[i] < nt ? ktile[i] : nt - 1;     // clamp loads in bounds
#pragma unroll
        for (int ss = 0; ss < KS; ++ss) {
            const size_t off = ((size_t)(kt * (DKP / 8) + 2 * ss + hh) * 32 + r) * 8;
            kfr[i][ss] = *reinterpret_cast<const bf16x8*>(Krb + off);
            vfr[i][ss] = *reinterpret_cast<const bf16x8*>(Vrb + off);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) { dKacc[i][j] = 0.f; dVacc[i][j] = 0.f; }
    }
    // position of this lane's key inside a dS row segment of its tile (MFMA k-order, see fragT_index)
    const int kpos = ((r >> 4) & 1) * 16 + ((r >> 2) & 1) * 8 + 4 * ((r >> 3) & 1) + (r & 3);

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
        // row constants (rows = queries of this tile, 4 consecutive per register group)
        f32x16 negL, negD;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lseb + qt * 32 + 8 * g + 4 * hh);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(delb + qt * 32 + 8 * g + 4 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) { negL[4 * g + i] = -l4[i]; negD[4 * g + i] = -d4[i]; }
        }
        bf16* dSw = dS_lds[qt & 1];
#pragma unroll
        for (int i = 0; i < KT; ++i) {
            f32x16 s = negL, dp = negD;
#pragma unroll
            for (int ss = 0; ss < KS; ++ss) { s = mfma32(qa[ss], kfr[i][ss], s); dp = mfma32(da[ss], vfr[i][ss], dp); }
            const bool key_ok = (ktile[i] * 32 + r) < T;
            f32x16 ds;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const bool ok = key_ok && (qt * 32 + acc32_row(j, hh) < T);
                const float pv = ok ? fast_exp2(s[j]) : 0.f;
                s[j] = pv;
                ds[j] = pv * dp[j];
            }
            if (drop.thr16) {
                // with dropped probabilities Pd = P*m/(1-p):  dV^T += dO^T Pd ;  dS = P * ((dO V^T)*m/(1-p) - delta)
                // (delta = rowsum(dO.O) is unchanged); dp holds dO V^T - delta, negD holds -delta.
                const uint64_t kcol = (uint64_t)ktile[i] * 32 + r;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint64_t idx = ((uint64_t)bh * Tp + (uint64_t)(qt * 32 + acc32_row(j, hh))) * Tp + kcol;
                    const float ms = drop_keep(drop, idx) ? drop.scale : 0.f;
                    ds[j] = s[j] * ((dp[j] - negD[j]) * ms + negD[j]);
                    s[j] *= ms;
                }
            }
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                dVacc[i] = mfma32(dT[s2], pack8(s, s2), dVacc[i]);
                dKacc[i] = mfma32(qT[s2], pack8(ds, s2), dKacc[i]);
            }
            // dS -> LDS [query row][key position], keys of tile (wave, i) at segment (wave*KT + i)*32
            const int seg = (wave * KT + i) * 32 + kpos;
#pragma unroll
            for (int j = 0; j < 16; ++j) dSw[acc32_row(j, hh) * LDS_ROW + seg] = (bf16)ds[j];
        }
        __syncthreads();
        if ((qt & 3) == wave) {
            // dQ tile [32 queries][e] = sum over the workgroup's keys of dS K
            f32x16 dq;
#pragma unroll
            for (int j = 0; j < 16; ++j) dq[j] = 0.f;
            for (int tl = 0; tl < KB_TILES; ++tl) {
                const int kt_g = kb * KB_TILES + tl;
                if (kt_g >= nt) break;
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const bf16x8 af = *reinterpret_cast<const bf16x8*>(dSw + r * LDS_ROW + tl * 32 + s2 * 16 + hh * 8);
                    const bf16x8 bf = *reinterpret_cast<const bf16x8*>(Ktb + ((size_t)((kt_g * 2 + s2) * 2 + hh) * 32 + r) * 8);
                    dq = mfma32(af, bf, dq);
                }
            }
            if (r < DKP) {
                float* slab = dq_slab + (size_t)kb * M * ldq;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int t = qt * 32 + acc32_row(j, hh);
                    if (t < T) slab[((size_t)b * T + t) * ldq + head * DKP + r] = dq[j];
                }
            }
        }
        // (the buffer written at tile qt is next written at qt+2, after the barrier of tile qt+1,
        //  which the dQ wave only reaches once it has finished reading it)
    }

    // dK = ln2 * acc (scores are in the log2 domain), dV = acc; rows e = acc32_row, column key = r
    const float LN2 = 0.6931471805599453f;
#pragma unroll
    for (int i = 0; i < KT; ++i) {
        const int t = ktile[i] * 32 + r;
        if (ktile[i] >= nt || t >= T) continue;
        const size_t m = (size_t)b * T + t;
#pragma unroll
        for (int g = 0; g < DKP / 8; ++g) {
            bf16x4 kv, vv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { kv[j] = (bf16)(dKacc[i][4 * g + j] * LN2); vv[j] = (bf16)dVacc[i][4 * g + j]; }
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

// dQ = rowmask * scale * sum over key-block slabs  ->  bf16 row-major columns [0,HD) of dQKV and its T layout
__global__ void dq_finish_kernel(const float* __restrict__ slab, int nkb, const float* __restrict__ rowmask,
                                 float scale, bf16* __restrict__ dqkv, int ld, bf16* __restrict__ dqkvT, int MP,
                                 int M, int HD, int ldq) {
    const int hd4 = HD >> 2;
    const size_t total = (size_t)M * hd4;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / hd4), c = (int)(idx - (size_t)m * hd4) * 4;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < nkb; ++s) acc += *reinterpret_cast<const f32x4*>(slab + ((size_t)s * M + m) * ldq + c);
        const float sc = (rowmask && rowmask[m] == 0.0f) ? 0.f : scale;   // blanked query rows pass no gradient to Q
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (bf16)(acc[i] * sc);
        *reinterpret_cast<bf16x4*>(dqkv + (size_t)m * ld + c) = o;
#pragma unroll
        for (int i = 0; i < 4; ++i) dqkvT[(size_t)(c + i) * MP + m] = o[i];
    }
}
