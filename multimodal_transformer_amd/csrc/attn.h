// Scaled-dot-product attention core for gfx950, forward and backward, over per-(batch,head)
// fragment-layout operands (common.h).  No (T x T) tensor is ever materialised
// (the reference does: transformer/MFT/multiTransformer.py:27-34).
//
// Mask semantics of the reference (query-ROW blanking, keys never masked, :29-31,48-50) are applied
// by the producers/consumers of these operands, not here: a blanked row arrives as Q' = 0, which
// soft-maxes to exactly 1/T over all T keys like the reference's constant -1e9 row, and its dQ is
// zeroed by attn_bwd_dq_kernel (which takes the row mask).  Otherwise only index masking lives here:
// keys/queries >= T in the last tile.
//
// Scores are kept in the log2 domain: the producer stores Q' = (x Wq^T + bq) * log2(e)/sqrt(d_k), so
// P = 2^(S' - L) with L = rowmax + log2(rowsum) saved per query for the backward pass.
#pragma once
#include <type_traits>
#include "common.h"
#include "attn_mask.h"

// ------------------------------------------------------------------------------------------------
// d_k = 64 (DKP = 64): the score products contract over all 64 features (KS = 4 k-steps), but a 32x32 MFMA accumulator holds only 32
// feature rows of an output (O^T, dK^T, dV^T, dQ^T): each launch produces ONE 32-feature block `fb` of its outputs and the host
// launches twice — the scores are evaluated twice, which this rarely used head size (no BASELINE config has d_k > 32) can afford.
// All three kernels below: a 1-D grid decoded by attn_block() into (tile quad, batch*head), 4 waves per workgroup, each wave owning one 32-window
// tile of its own and sweeping the other axis.  The swept operand tile (K/V for the forward and dQ kernels,
// Q/dO/L/delta for the dK-dV kernel) is the same for the four waves, so the workgroup stages it into LDS
// cooperatively: 2-3 coalesced 16-byte loads per thread, issued one tile AHEAD into registers while the current
// tile computes, written to the other half of a double buffer, one barrier per tile.  The LDS image keeps the
// global fragment layout (R: lane-linear 16-byte pieces), so a wave's MFMA operand for a product that contracts over
// features is a conflict-free ds_read_b128; operands of products that contract over windows come out of the same
// tiles through transposing reads (common.h tr_frag2; those tiles keep their 8-feature groups 576 bytes apart).
// Register budgets stay near 80-130 VGPRs: 3-5 waves share a SIMD.
template <int NSEG, int MAXL = 3>
struct TileStager {
    // segment i: `pieces[i]` 16-byte pieces per tile, read from base[i] + tile * stride[i] (bf16 elements); thread t moves pieces
    // t, t + 256, ... (MAXL = ceil(total pieces / 256) of them).  In LDS segment i starts at piece lds0[i] and leaves pad[i] pieces
    // free after every 32 (an R-layout tile that is read with transposing reads keeps its 8-feature groups 576 bytes apart:
    // the four groups a 32-lane half touches then fall into the four quarters of the 64 banks).
    const bf16* src[MAXL]; int stride[MAXL]; bool on[MAXL]; int dst[MAXL];
    bf16x8 reg[MAXL];
    __device__ __forceinline__ void init(const bf16* const* base, const int* pieces, const int* strides, const int* lds0, const int* pad, int tid) {
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
            const int p = tid + MMT_THREADS * l;
            int acc = 0; on[l] = false; src[l] = base[0]; stride[l] = 0; dst[l] = 0;
#pragma unroll
            for (int sg = 0; sg < NSEG; ++sg) {
                if (!on[l] && p >= acc && p < acc + pieces[sg]) {
                    on[l] = true; src[l] = base[sg] + (size_t)(p - acc) * 8; stride[l] = strides[sg];
                    dst[l] = (lds0[sg] + (p - acc) + ((p - acc) >> 5) * pad[sg]) * 8;
                }
                acc += pieces[sg];
            }
        }
    }
    __device__ __forceinline__ void load(int tile) {
#pragma unroll
        for (int l = 0; l < MAXL; ++l) if (on[l]) reg[l] = *reinterpret_cast<const bf16x8*>(src[l] + (size_t)tile * stride[l]);
    }
    __device__ __forceinline__ void store(bf16* lds) const {
#pragma unroll
        for (int l = 0; l < MAXL; ++l) if (on[l]) *reinterpret_cast<bf16x8*>(lds + dst[l]) = reg[l];
    }
};
#define MMT_TR_OCT 36        // pieces between the 8-feature groups of a padded R tile in LDS (32 windows + 4 free)

// all 16 registers = x in eight 64-bit moves (hipcc writes a splat as sixteen v_mov_b32).  The trailing s_nop covers the two
// wait states an MFMA needs after a VALU write of its SrcC: the hazard recognizer does not see through asm statements.
__device__ __forceinline__ void fill16(f32x16& v, float x) {
    const f32x2 xp = {x, x};
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        f32x2 t;
        if (i < 14) asm volatile("v_mov_b64 %0, %1" : "=v"(t) : "v"(xp));       // volatile: eight separate moves, not one plus copies
        else asm volatile("v_mov_b64 %0, %1\n\ts_nop 1" : "=v"(t) : "v"(xp));
        v[i] = t[0]; v[i + 1] = t[1];
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

// Workgroup -> (tile quad bx, batch*head bh).  The (nt+3)/4 workgroups that sweep the same (batch, head) all stream
// that head's whole K/V (forward) or Q/dO (backward) fragments.  The hardware deals consecutive workgroup ids round-robin
// to the 8 XCDs, each with a private L2, so a plain 2-D grid would put the sharers on different XCDs and fetch the
// operands once per sharer from the fabric (measured with FETCH_SIZE: 2.5-3x the operand bytes).  A 1-D grid of
// nx * 8 * ceil(nbh/8) ids is decoded so that ids congruent mod 8 — same XCD — carry the same head.
// Progress-based wave priority.  The four waves that share a SIMD in these kernels belong to four different workgroups; the SIMD
// arbitrates vector issue by priority, then AGE, so at equal priority the oldest wave runs almost unimpeded and the youngest
// starves: measured wave lifetimes in attn_fwd_kernel spread from 31k to 56k cycles for identical work, and while the last
// waves finish alone the SIMD issues at a fraction of its four-wave rate (a wave by itself issues one vector instruction per
// 6-12 cycles: tools/valu_micro.hip).  Each wave therefore LOWERS its priority as it advances through its sweep (3 in the first
// quarter .. 0 in the last): whoever is behind goes first, the waves of a SIMD finish together, and no tail is left.
__device__ __forceinline__ void progress_prio(int step, int nsteps) {
    const int q1 = (nsteps + 3) >> 2, q2 = (nsteps + 1) >> 1, q3 = (3 * nsteps + 3) >> 2;      // wave-uniform
    if (step == 0) __builtin_amdgcn_s_setprio(3);
    else if (step == q1) __builtin_amdgcn_s_setprio(2);
    else if (step == q2) __builtin_amdgcn_s_setprio(1);
    else if (step == q3) __builtin_amdgcn_s_setprio(0);
}

struct AttnBlock { int bx, bh; bool valid; };
__device__ __forceinline__ AttnBlock attn_block(int nx, int nbh) {
    const int L = blockIdx.x, grp = L / (8 * nx), rem = L - grp * 8 * nx;
    AttnBlock a;
    a.bx = rem >> 3;
    a.bh = grp * 8 + (rem & 7);
    a.valid = a.bh < nbh;
    return a;
}
__host__ inline int attn_grid(int nx, int nbh) { return nx * 8 * ((nbh + 7) / 8); }

// (`//@ name` comment lines mark the phase boundaries at which tools/make_diag.py inserts cycle stamps into the GENERATED stamped twin of this
// kernel — diagnostic builds only; this file holds no diagnostic code)
template <int DKP, bool DROP>
__global__ __launch_bounds__(MMT_THREADS, 2) void attn_fwd_kernel(
        const bf16* __restrict__ Qr, const bf16* __restrict__ Kr, const bf16* __restrict__ Vr,
        bf16* __restrict__ ctx, float* __restrict__ lse,
        int h, int T, int nt, int nbh, int ldc, const uint16_t* __restrict__ maskQ, float drop_scale, int fb) {
    constexpr int KS = DKP / 16;
    constexpr int DKB = DKP < 32 ? DKP : 32;           // feature rows of this launch's output block
    constexpr bool ONES = (DKP == 16) && !DROP;        // row sums through the MFMA
    //@ entry
    // 16-byte pieces per tile: K (R layout, all DKP features) and the 32-feature block `fb` of V — ALSO in the R layout, as the QKV
    // epilogue wrote it: the PV product contracts over keys, its A fragment (8 keys of one feature per lane) comes out of LDS
    // through transposing reads (tr_frag2), so no transposed copy of V exists in memory
    constexpr int PK = DKP * 4, PV = DKB * 4, PVL = (DKB / 8) * MMT_TR_OCT;
    __shared__ __attribute__((aligned(16))) bf16 stage[2][(PK + PVL) * 8];
    __shared__ __attribute__((aligned(16))) bf16 zeros[256];          // what the lanes of the padding feature rows (>= d_k) read
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const AttnBlock ab = attn_block((nt + 3) >> 2, nbh);
    if (!ab.valid) return;                              // whole workgroup, before any barrier
    const int qt = ab.bx * 4 + wave;
    const bool live = qt < nt;                          // idle waves still stage and synchronise
    const int qtc = live ? qt : nt - 1;
    const int bh = ab.bh, b = bh / h, head = bh - b * h;
    const int Tp = nt * 32;
    const bf16* Qb = Qr + (size_t)bh * fragR_elems(Tp, DKP);
    const bf16* Kb = Kr + (size_t)bh * fragR_elems(Tp, DKP);
    const bf16* Vb = Vr + (size_t)bh * fragR_elems(Tp, DKP) + (size_t)fb * 4 * 256;      // feature group 4 fb of tile 0
    // dropout: this lane's 16-bit words of the wave's row of mask blocks (attn_mask.h, LQ layout), one per key tile, fetched a tile ahead
    const uint16_t* mrow = maskQ + ((size_t)bh * nt + qtc) * nt * 64 + lane;
    uint32_t mw = DROP ? mrow[0] : 0u;

    TileStager<2, (PK + PV + MMT_THREADS - 1) / MMT_THREADS> stg;
    {
        const bf16* base[2] = {Kb, Vb};
        const int pieces[2] = {PK, PV}, strides[2] = {32 * DKP, 32 * DKP}, lds0[2] = {0, PK}, pad[2] = {0, MMT_TR_OCT - 32};
        stg.init(base, pieces, strides, lds0, pad, threadIdx.x);
    }
    stg.load(0);
    zeros[threadIdx.x] = (bf16)0.f;
    // transposing-read role of this lane in the V tile: its feature row r = 16 g1 + (lane & 15) is delivered by the hardware; the
    // address it SUPPLIES is that of window q = (lane >> 2) & 3 of a 4-window block, features 16 g1 + 4 pp .. + 3, pp = lane & 3.
    // Key order of the fragment = row order of the P^T accumulator: slot j <-> key 16 s2 + 8 (j >> 2) + 4 hh + (j & 3).
    const int g1 = (lane >> 4) & 1, tq = (lane >> 2) & 3, tpp = lane & 3;
    const int voff = ((2 * g1 + (tpp >> 1)) * MMT_TR_OCT + 4 * hh + tq) * 8 + 4 * (tpp & 1);
    const bool vpad = DKB < 32 && g1;                   // feature rows >= 16 at d_k = 16: zeros (or the ones row, below)
    bf16x8 qf[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s)
        qf[s] = *reinterpret_cast<const bf16x8*>(Qb + ((size_t)(qtc * (DKP / 8) + 2 * s + hh) * 32 + r) * 8);
    bf16x8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (bf16)1.0f;
    stg.store(stage[0]);
    __syncthreads();

    f32x16 o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.f;
    float mrun = 0.f, lrun = 0.f;
    f32x16 mneg;                                        // -mrun in every register: the accumulator init of the score product
#pragma unroll
    for (int i = 0; i < 16; ++i) mneg[i] = 0.f;

    // The tile body is instantiated twice: TAIL = false for the hot loop (no conditional code between an MFMA and the
    // consumers of its result: hipcc's hazard recognizer counts a skippable block's instructions as MFMA->VALU wait
    // states, which is wrong on the taken path — seen as 27 % wrong dQ in attn_bwd_dq_kernel<32>), TAIL = true for the
    // last key tile, whose index masking is then straight-line code.
    //@ prologue
    auto body = [&](auto tail_tag, int kt) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        //@ loop_top
        progress_prio(kt, nt);
        const uint32_t tw = mw;                          // this tile's keep bits
        if (DROP && !TAIL) mw = mrow[(size_t)(kt + 1) * 64];
        if (!TAIL) stg.load(kt + 1);                    // next tile in flight behind this tile's arithmetic
        const bf16* sk = stage[kt & 1];
        const bf16* sv = vpad ? zeros : sk + PK * 8 + voff;
        // S' - m straight out of the MFMA: the accumulator init is a register set that holds -m for the whole sweep and is rewritten
        // only when the reference moves (round 3 splatted -m into the accumulator before every tile: 8 v_mov_b64 of ~100 vector
        // instructions per tile)
        f32x16 s;
        if constexpr (KS == 1) {
            // written as asm because hipcc only knows the tied form (vdst = srcC) of an MFMA on VGPRs and would copy mneg into s first;
            // the trailing s_nop are the wait states between an 8-pass MFMA and a vector instruction that reads its result, which the
            // hazard recognizer cannot add for an asm statement
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(sk + (hh * 32 + r) * 8);
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3\n\ts_nop 11" : "=&v"(s) : "v"(kf), "v"(qf[0]), "v"(mneg));
        } else {
            s = mfma32(*reinterpret_cast<const bf16x8*>(sk + (hh * 32 + r) * 8), qf[0], mneg);
#pragma unroll
            for (int ss = 1; ss < KS; ++ss)
                s = mfma32(*reinterpret_cast<const bf16x8*>(sk + ((2 * ss + hh) * 32 + r) * 8), qf[ss], s);
        }
        if (TAIL) {                                     // keys >= T do not exist
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = (kt * 32 + acc32_row(i, hh) < T) ? s[i] : -INFINITY;
        }
        // (no inline asm here: hipcc's hazard recognizer does not count an asm statement as a reader of MFMA results)
        float tmax = fmaxf(fmaxf(s[0], s[1]), s[2]);
#pragma unroll
        for (int i = 3; i < 15; i += 2) tmax = fmaxf(fmaxf(tmax, s[i]), s[i + 1]);
        tmax = fmaxf(tmax, s[15]);                      // over this lane's 16 keys; the other half of the wave holds the query's other 16
        //@ qk_max
        if (kt == 0 || __any(tmax > MMT_RESCALE_THR)) {      // __any looks at both halves: no exchange on the common path
            // move the reference to the new running max (first tile: from 0 to the tile max, with o = l = 0)
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));  // finite: every tile holds >= 1 real key in one of the halves
            const float dlt = (kt == 0) ? tmax : fmaxf(tmax, 0.f);
            const float alpha = fast_exp2(-dlt);
            mrun += dlt;
            lrun *= alpha;
#pragma unroll
            for (int i = 0; i < 16; ++i) { o[i] *= alpha; s[i] -= dlt; }
            fill16(mneg, -mrun);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = fast_exp2(s[i]);
        if (!ONES) {        // row sum of the tile as ONE chain of plain adds: a wave issues a vector instruction every 6+ cycles
            float a = s[0];             // anyway, so the dependency costs nothing, and a v_add_f32 takes 1.6 cycles of SIMD issue where the
#pragma unroll                          // v_pk_add_f32 that two parallel chains get packed into takes 3.7 (tools/valu_micro.hip)
            for (int i = 1; i < 16; ++i) a += s[i];
            lrun += a;
        }
        //@ exp_sum
        if (DROP) {                                     // zero the dropped probabilities; 1/(1-p) is applied once, to the output row
            static_for<0, 16>([&](auto ic) { constexpr int i = decltype(ic)::value; s[i] = keep_and<i>(s[i], tw); });
        }
        //@ mask
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            bf16x8 va = tr_frag2(sv + 128 * s2, sv + 128 * s2 + 64);       // keys 16 s2 + 4 hh + {0..3}, then + 8
            if (ONES && r == DKP) va = ones;           // V^T rows >= DKP read zeros; row DKP becomes the ones row
            o = mfma32(va, pack8(s, s2), o);
        }
        //@ pv
        if (!TAIL) stg.store(stage[(kt + 1) & 1]);
        //@ stage_wait
        __syncthreads();        // stage[(kt+1)&1] was last read at tile kt-1, i.e. before the previous barrier
        //@ barrier
    };
    for (int kt = 0; kt < nt - 1; ++kt) body(std::false_type{}, kt);
    body(std::true_type{}, nt - 1);
    //@ swept
    if (!live) return;
    float ltot;
    if (ONES) ltot = __shfl(o[8], r);                  // O^T row 16 = (register 8, lower half): the row sums
    else ltot = lrun + __shfl_xor(lrun, 32);
    const float inv = (DROP ? drop_scale : 1.0f) / ltot;    // kept probabilities carry 1/(1-p)
    const int t = qt * 32 + r;
    if (t < T) {
        const size_t m = (size_t)b * T + t;
        if (hh == 0) lse[(size_t)bh * Tp + t] = -(mrun + fast_log2(ltot));   // stored NEGATED: it is the accumulator init of the backward
        // O^T rows (head features) live in registers: e = acc32_row(i, hh); groups of 4 are contiguous
#pragma unroll
        for (int g = 0; g < DKB / 8; ++g) {
            bf16x4 v;
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = (bf16)(o[4 * g + i] * inv);
            const int e0 = 32 * fb + 8 * g + 4 * hh;
            *reinterpret_cast<bf16x4*>(ctx + m * ldc + head * DKP + e0) = v;
        }
    }
    //@ exit
}

// ------------------------------------------------------------------------------------------------
// Backward, part A: dK, dV.  Wave w owns key tile kt = 4*blockIdx.x + w (the key on the lane column) and
// sweeps all query tiles, keeping dK^T, dV^T in accumulators.  Per query tile:
//     S'  = Q' K^T - L      (-L, stored negated, is the accumulator init: P = 2^S' needs no subtraction)
//     dPc = dO V^T - delta  (same trick with -delta = -rowsum(dO . O))
//     dS  = P * dPc ;  dV^T += dO^T P ;  dK^T += Q'^T dS     (P, dS accumulators ARE the B operands)
template <int DKP, bool DROP>
__global__ __launch_bounds__(MMT_THREADS, DKP == 16 ? 4 : 2) void attn_bwd_dkv_kernel(
        const bf16* __restrict__ Qr, const bf16* __restrict__ Kr, const bf16* __restrict__ Vr, const bf16* __restrict__ dOr,
        const float* __restrict__ lse, const float* __restrict__ delta,
        bf16* __restrict__ dkv, int lddkv,      // row-major [M][lddkv]; dK at column HD, dV at 2*HD
        int h, int T, int nt, int nbh, const uint16_t* __restrict__ maskK, float drop_scale, int fb) {
    constexpr int KS = DKP / 16;
    constexpr int DKB = DKP < 32 ? DKP : 32;
    // pieces: R-layout tile (in LDS with its 8-feature groups 576 bytes apart: the products that contract over the queries read
    // their A fragments out of these tiles with transposing reads, like attn_fwd_kernel reads V), 32 fp32 row constants
    constexpr int PR = DKP * 4, PRL = (DKP / 8) * MMT_TR_OCT, PC = 8;
    constexpr int TOTAL = 2 * PR + 2 * PC, TOTAL_LDS = 2 * PRL + 2 * PC;
    __shared__ __attribute__((aligned(16))) bf16 stage[2][TOTAL_LDS * 8];
    __shared__ __attribute__((aligned(16))) bf16 zeros[256];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const AttnBlock ab = attn_block((nt + 3) >> 2, nbh);
    if (!ab.valid) return;                              // whole workgroup, before any barrier
    const int kt = ab.bx * 4 + wave;
    const bool live = kt < nt;
    const int ktc = live ? kt : nt - 1;
    const int bh = ab.bh, b = bh / h, head = bh - b * h;
    const int Tp = nt * 32, HD = h * DKP;
    const size_t offR = (size_t)bh * fragR_elems(Tp, DKP);
    const bf16 *Krb = Kr + offR, *Vrb = Vr + offR;
    const uint16_t* mrow = maskK + ((size_t)bh * nt + ktc) * nt * 64 + lane;      // LK layout: this lane's word of one block per query tile
    uint32_t mw = DROP ? mrow[0] : 0u;
    const uint32_t scale_bits = __builtin_bit_cast(uint32_t, drop_scale);

    TileStager<4, (TOTAL + MMT_THREADS - 1) / MMT_THREADS> stg;
    {
        const bf16* base[4] = {Qr + offR, dOr + offR,
                               reinterpret_cast<const bf16*>(lse + (size_t)bh * Tp), reinterpret_cast<const bf16*>(delta + (size_t)bh * Tp)};
        const int pieces[4] = {PR, PR, PC, PC}, strides[4] = {32 * DKP, 32 * DKP, 64, 64};
        const int lds0[4] = {0, PRL, 2 * PRL, 2 * PRL + PC}, pad[4] = {MMT_TR_OCT - 32, MMT_TR_OCT - 32, 0, 0};
        stg.init(base, pieces, strides, lds0, pad, threadIdx.x);
    }
    stg.load(0);
    zeros[threadIdx.x] = (bf16)0.f;
    bf16x8 kfr[KS], vfr[KS];
#pragma unroll
    for (int ss = 0; ss < KS; ++ss) {
        const size_t off = ((size_t)(ktc * (DKP / 8) + 2 * ss + hh) * 32 + r) * 8;
        kfr[ss] = *reinterpret_cast<const bf16x8*>(Krb + off);
        vfr[ss] = *reinterpret_cast<const bf16x8*>(Vrb + off);
    }
    // DKP == 16: a 32-row A fragment has 16 idle rows, so ONE accumulator serves both products: lanes r < 16 read dO^T (rows 0..15:
    // dV^T) and zeros for Q^T, lanes r >= 16 read Q^T features r - 16 (rows 16..31: dK^T) and zeros for dO^T.  16 VGPRs less
    // puts the kernel at 4 waves per SIMD.  DKP >= 32 keeps two accumulators.
    constexpr bool ONEACC = (DKP == 16);
    f32x16 dKacc, dVacc;
#pragma unroll
    for (int j = 0; j < 16; ++j) { dKacc[j] = 0.f; dVacc[j] = 0.f; }
    // transposing-read role (see attn_fwd_kernel): the lane supplies query tq of a 4-query block, features 4 tpp .. + 3 of its
    // 16-lane group's 16 features, and receives feature row r; slot j <-> query 16 s2 + 8 (j >> 2) + 4 hh + (j & 3)
    const int g1 = (lane >> 4) & 1, tq = (lane >> 2) & 3, tpp = lane & 3;
    const int toff = (((ONEACC ? 0 : 4 * fb + 2 * g1) + (tpp >> 1)) * MMT_TR_OCT + 4 * hh + tq) * 8 + 4 * (tpp & 1);
    const bool key_tail = (ktc == nt - 1) && (T & 31);
    const bool key_ok = (ktc * 32 + r) < T;
    stg.store(stage[0]);
    __syncthreads();

    // Tile body without conditional code (see attn_fwd_kernel): the key-tail mask is a per-lane multiplier that is
    // simply 1 everywhere except in the wave that owns the last key tile, and the query tail is the peeled last tile.
    const float kmul = (key_tail && !key_ok) ? 0.f : 1.f;
    auto body = [&](auto tail_tag, int qt) {
        constexpr bool QTAIL = decltype(tail_tag)::value;
        const bool more = qt + 1 < nt;                  // scalar, loop-invariant except at the very last tile
        progress_prio(qt, nt);
        const uint32_t tw = mw;
        if (DROP && more) mw = mrow[(size_t)(qt + 1) * 64];
        if (more) stg.load(qt + 1);
        const bf16* sq = stage[qt & 1];
        const bf16* sdo = sq + PRL * 8;
        const float* sl = reinterpret_cast<const float*>(sq + 2 * PRL * 8);
        const float* sd = sl + 32;
        const bf16* ado = (ONEACC && g1) ? zeros : sdo + toff;
        const bf16* aq = (ONEACC && !g1) ? zeros : sq + toff;
        f32x16 s, dp;                   // row constants (4 consecutive queries per register group) as the accumulators
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(sl + 8 * g + 4 * hh);
            f32x4 d4 = {0.f, 0.f, 0.f, 0.f};                                   // DROP: -delta enters after the mask
            if (!DROP) d4 = *reinterpret_cast<const f32x4*>(sd + 8 * g + 4 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) { s[4 * g + i] = l4[i]; dp[4 * g + i] = d4[i]; }      // both stored negated
        }
#pragma unroll
        for (int ss = 0; ss < KS; ++ss) {
            const int o8 = ((2 * ss + hh) * MMT_TR_OCT + r) * 8;
            s = mfma32(*reinterpret_cast<const bf16x8*>(sq + o8), kfr[ss], s);
            dp = mfma32(*reinterpret_cast<const bf16x8*>(sdo + o8), vfr[ss], dp);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float pv = fast_exp2(s[j]);
            if (QTAIL) pv = (qt * 32 + acc32_row(j, hh) < T) ? pv : 0.f;      // queries >= T do not exist
            s[j] = pv;
        }
        if (key_tail) {                                 // wave-uniform, loop-invariant: only the last key tile's wave pays
#pragma unroll
            for (int j = 0; j < 16; ++j) s[j] *= kmul;
        }
        if (DROP) {
            // dropped probabilities Pd = P*m/(1-p):  dV^T += dO^T Pd ;  dS = P * ((dO V^T)*m/(1-p) - delta); delta unchanged.
            // m = bit j of the lane's tile word (LK layout: the key on the lane, query acc32_row(j, hh) in register j)
            static_for<0, 4>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(sd + 8 * g + 4 * hh);
                static_for<0, 4>([&](auto ic) {
                    constexpr int i = decltype(ic)::value, j = 4 * g + i;
                    const float ms = __builtin_bit_cast(float, scale_bits & keep_bits<j>(tw));
                    dp[j] = s[j] * fmaf(dp[j], ms, d4[i]);
                    s[j] *= ms;
                });
            });
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) dp[j] *= s[j];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            dVacc = mfma32(tr_frag2(ado + 128 * s2, ado + 128 * s2 + 64), pack8(s, s2), dVacc);
            if (ONEACC) dVacc = mfma32(tr_frag2(aq + 128 * s2, aq + 128 * s2 + 64), pack8(dp, s2), dVacc);
            else dKacc = mfma32(tr_frag2(aq + 128 * s2, aq + 128 * s2 + 64), pack8(dp, s2), dKacc);
        }
        if (more) stg.store(stage[(qt + 1) & 1]);
        __syncthreads();
    };
    for (int qt = 0; qt < nt - 1; ++qt) body(std::false_type{}, qt);
    if (T & 31) body(std::true_type{}, nt - 1); else body(std::false_type{}, nt - 1);
    if (!live) return;
    // dK = ln2 * acc (scores are in the log2 domain), dV = acc; rows e = acc32_row, column key = r
    const float LN2 = 0.6931471805599453f;
    const int t = kt * 32 + r;
    if (t < T) {
        const size_t m = (size_t)b * T + t;
#pragma unroll
        for (int g = 0; g < DKB / 8; ++g) {
            bf16x4 kv, vv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kv[j] = (bf16)((ONEACC ? dVacc[8 + 4 * g + j] : dKacc[4 * g + j]) * LN2);      // ONEACC: rows 16 + e sit in registers 8..15
                vv[j] = (bf16)dVacc[4 * g + j];
            }
            const int e0 = head * DKP + 32 * fb + 8 * g + 4 * hh;
            *reinterpret_cast<bf16x4*>(dkv + m * lddkv + HD + e0) = kv;
            *reinterpret_cast<bf16x4*>(dkv + m * lddkv + 2 * HD + e0) = vv;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Backward, part B: dQ.  Same orientation as the forward (query on the lane): wave w owns query tile
// qt = 4*blockIdx.x + w and sweeps all key tiles; L and delta are lane constants; the dS^T accumulator
// (keys in registers) is the B operand of dQ^T += K^T dS^T.  The epilogue applies 1/sqrt(d_k) and the
// query-row mask (blanked rows pass no gradient to Q) and writes columns [0,HD) of dQKV in both layouts.
template <int DKP, bool DROP>
__global__ __launch_bounds__(MMT_THREADS, 2) void attn_bwd_dq_kernel(
        const bf16* __restrict__ Qr, const bf16* __restrict__ Kr, const bf16* __restrict__ Vr,
        const bf16* __restrict__ dOr, const float* __restrict__ lse, const float* __restrict__ delta,
        const float* __restrict__ rowmask, float scale,
        bf16* __restrict__ dqkv, int lddqkv,
        int h, int T, int nt, int nbh, const uint16_t* __restrict__ maskQ, float drop_scale, int fb) {
    constexpr int KS = DKP / 16;
    constexpr int DKB = DKP < 32 ? DKP : 32;
    constexpr int PR = DKP * 4, PRL = (DKP / 8) * MMT_TR_OCT;      // K tile padded in LDS: K^T fragments by transposing reads
    __shared__ __attribute__((aligned(16))) bf16 stage[2][(PRL + PR) * 8];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const AttnBlock ab = attn_block((nt + 3) >> 2, nbh);
    if (!ab.valid) return;                              // whole workgroup, before any barrier
    const int qt = ab.bx * 4 + wave;
    const bool live = qt < nt;
    const int qtc = live ? qt : nt - 1;
    const int bh = ab.bh, b = bh / h, head = bh - b * h;
    const int Tp = nt * 32;
    const size_t offR = (size_t)bh * fragR_elems(Tp, DKP);
    const bf16 *Qrb = Qr + offR, *dOrb = dOr + offR;
    const uint16_t* mrow = maskQ + ((size_t)bh * nt + qtc) * nt * 64 + lane;      // LQ layout
    uint32_t mw = DROP ? mrow[0] : 0u;
    const uint32_t scale_bits = __builtin_bit_cast(uint32_t, drop_scale);

    TileStager<2, (2 * PR + MMT_THREADS - 1) / MMT_THREADS> stg;
    {
        const bf16* base[2] = {Kr + offR, Vr + offR};
        const int pieces[2] = {PR, PR}, strides[2] = {32 * DKP, 32 * DKP}, lds0[2] = {0, PRL}, pad[2] = {MMT_TR_OCT - 32, 0};
        stg.init(base, pieces, strides, lds0, pad, threadIdx.x);
    }
    stg.load(0);
    // transposing-read role (see attn_fwd_kernel): feature row 32 fb + r of K^T; slot j <-> key 16 s2 + 8 (j >> 2) + 4 hh + (j & 3).
    // d_k = 16: the lanes of rows >= 16 (nobody reads their dQ^T rows) fetch what lanes r - 16 fetch.
    const int g1 = (lane >> 4) & 1, tq = (lane >> 2) & 3, tpp = lane & 3;
    const int toff = (((DKP == 16 ? 0 : 4 * fb + 2 * g1) + (tpp >> 1)) * MMT_TR_OCT + 4 * hh + tq) * 8 + 4 * (tpp & 1);
    bf16x8 qf[KS], dof[KS];
#pragma unroll
    for (int ss = 0; ss < KS; ++ss) {
        const size_t off = ((size_t)(qtc * (DKP / 8) + 2 * ss + hh) * 32 + r) * 8;
        qf[ss] = *reinterpret_cast<const bf16x8*>(Qrb + off);
        dof[ss] = *reinterpret_cast<const bf16x8*>(dOrb + off);
    }
    const float negL = lse[(size_t)bh * Tp + qtc * 32 + r];          // both stored negated by their producers
    const float negD = delta[(size_t)bh * Tp + qtc * 32 + r];
    f32x16 dq;
#pragma unroll
    for (int j = 0; j < 16; ++j) dq[j] = 0.f;
    stg.store(stage[0]);
    __syncthreads();

    // Tile body without conditional code between MFMAs and their consumers (see attn_fwd_kernel); key tail peeled.
    auto body = [&](auto tail_tag, int kt) {
        constexpr bool TAIL = decltype(tail_tag)::value;
        const bool more = kt + 1 < nt;
        progress_prio(kt, nt);
        const uint32_t tw = mw;
        if (DROP && more) mw = mrow[(size_t)(kt + 1) * 64];
        if (more) stg.load(kt + 1);
        const bf16* sk = stage[kt & 1];
        const bf16* sv = sk + PRL * 8;
        const bf16* skt = sk + toff;
        f32x16 s, dp;
        fill16(s, negL);
        if (DROP) {                                     // -delta enters after the mask
#pragma unroll
            for (int j = 0; j < 16; ++j) dp[j] = 0.f;
        } else fill16(dp, negD);
#pragma unroll
        for (int ss = 0; ss < KS; ++ss) {
            const int o8 = ((2 * ss + hh) * 32 + r) * 8;
            s = mfma32(*reinterpret_cast<const bf16x8*>(sk + o8 + (2 * ss + hh) * (MMT_TR_OCT - 32) * 8), qf[ss], s);
            dp = mfma32(*reinterpret_cast<const bf16x8*>(sv + o8), dof[ss], dp);
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float pv = fast_exp2(s[j]);
            if (TAIL) pv = (kt * 32 + acc32_row(j, hh) < T) ? pv : 0.f;       // keys >= T do not exist
            s[j] = pv;
        }
        if (DROP) {
            // dS = P * ((dO V^T) * m/(1-p) - delta)
            static_for<0, 16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                dp[j] = s[j] * fmaf(dp[j], __builtin_bit_cast(float, scale_bits & keep_bits<j>(tw)), negD);
            });
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) dp[j] *= s[j];
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
            dq = mfma32(tr_frag2(skt + 128 * s2, skt + 128 * s2 + 64), pack8(dp, s2), dq);
        if (more) stg.store(stage[(kt + 1) & 1]);
        __syncthreads();
    };
    for (int kt = 0; kt < nt - 1; ++kt) body(std::false_type{}, kt);
    if (T & 31) body(std::true_type{}, nt - 1); else body(std::false_type{}, nt - 1);
    if (!live) return;
    const int t = qt * 32 + r;
    if (t < T) {
        const size_t m = (size_t)b * T + t;
        const float sc = (rowmask && rowmask[m] == 0.0f) ? 0.f : scale;     // blanked query rows pass no gradient to Q
#pragma unroll
        for (int g = 0; g < DKB / 8; ++g) {
            bf16x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (bf16)(dq[4 * g + j] * sc);
            const int e0 = head * DKP + 32 * fb + 8 * g + 4 * hh;
            *reinterpret_cast<bf16x4*>(dqkv + m * lddqkv + e0) = v;
        }
    }
}
