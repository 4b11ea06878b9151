// Attention backward in ONE kernel for d_k <= 16 and 9..16 key tiles (T = 257..512): dQ, dK and dV from a single evaluation of
// P = 2^(S'-L) and dS = P (dP - delta) per score, instead of once in attn_bwd_dkv_kernel and once more in attn_bwd_dq_kernel
// (attn.h).  Both of those are VALU-issue bound (exp, the dropout hash, the bf16 converts), so the second evaluation is what the
// fusion removes; the MFMA work is unchanged.
//
// One workgroup of 16 waves owns one (batch, head).  Wave w owns key tile w and keeps dK^T / dV^T of that tile in ONE accumulator
// (attn_bwd_dkv_kernel's d_k = 16 trick); all waves sweep the query tiles together, sharing the staged Q / dO / L / delta tile like
// the 4-wave kernels do.  Per query tile each wave also forms its 32-key share of dQ^T = K^T dS^T:
//   * dS sits in the accumulator layout with the key on the lane; the dQ product contracts over keys, so dS goes through a
//     wave-private LDS patch to become a B operand: each lane writes its key's 16 packed values as they lie in its registers (two
//     ds_write_b128; round 1 scattered them with 16 ds_write_b16) and the query-major fragments come back through transposing reads;
//   * every operand exists in memory in ONE layout, the R fragment layout (window-major).  The products that contract over windows
//     (dV^T += dO^T P, dK^T += Q'^T dS over the queries, dQ^T = K^T dS^T over the keys) take their A fragments out of the staged
//     tiles with transposing LDS reads (common.h tr_frag2); round 1/2 staged a second, transposed copy of Q and dO (2 KB of the
//     3.2 KB per tile, half of it zero padding) and kept a transposed copy of the own K tile;
//   * the fp32 partial (8 registers for d_k <= 16) overwrites the patch, and after the tile's barrier all 16 waves sum the 16
//     partials in a fixed order (bit-reproducible, no atomics), apply 1/sqrt(d_k) and the query-row mask and store both layouts of
//     dQ — each wave 32 outputs per layout, coalesced.  Patch/partial regions are double-buffered by tile parity, so one barrier
//     per tile orders everything;
//   * the score products of tile t+1 are issued before the barrier that closes tile t (three-deep staging ring), see "Pipeline".
// Registers: 16 waves per CU means 128 VGPRs per wave; the kernel is written to fit (the own tile's K and V fragments live in
// LDS instead of registers, the staging ring needs one 16-byte piece per thread).
//
// Reference semantics: transformer/MFT/multiTransformer.py:22-34 (scaled dot-product attention) under autograd.
#pragma once
#include "attn.h"

#define MMT_FUSED_NW 16
#define MMT_FUSED_THREADS (MMT_FUSED_NW * 64)
#define MMT_FUSED_PATCH_LD 40                               // bf16 per patch row (one key): 2 halves x 16 accumulator slots + 8 pad (80-byte rows: conflict-free b128 writes)
#define MMT_FUSED_PART_LD 68                                // floats per partial register row: 64 lanes + 4 pad
#define MMT_FUSED_REGION_BYTES 2560                         // max(32 * 40 * 2, 8 * 68 * 4)
#define MMT_FUSED_RT_PIECES (2 * MMT_TR_OCT)                // an R tile of d_k = 16 in LDS: two 8-feature groups, 576 bytes apart (attn.h TileStager)
#define MMT_FUSED_STAGE_PIECES (2 * MMT_FUSED_RT_PIECES + 16)        // LDS pieces per staged query tile: Q, dO, 2 * 8 pieces of row constants
#define MMT_FUSED_STAGE_LOADS 144                           // 16-byte pieces fetched per query tile: 2 * 64 + 2 * 8
#define MMT_FUSED_ZERO_BYTES 512
#define MMT_FUSED_LDS_BYTES (3 * MMT_FUSED_STAGE_PIECES * 16 + MMT_FUSED_ZERO_BYTES + MMT_FUSED_NW * 2 * MMT_FUSED_RT_PIECES * 16 \
                             + MMT_FUSED_NW * 2 * MMT_FUSED_REGION_BYTES)     // 126,976 of 163,840

__host__ inline bool attn_bwd_fused_ok(int DKP, int nt) { return DKP == 16 && nt > 8 && nt <= MMT_FUSED_NW; }

// STAMP (diagnostic build, -DMMT_ABLATIONS): s_memtime at six points of the tile body, summed per wave into g_attn_stamps
#ifdef MMT_ABLATIONS
#define FB_STAMP(n) do { if (STAMP) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
    st_acc[n] += t_ - st_prev; st_prev = t_; } } while (0)
#else
#define FB_STAMP(n)
#endif
template <bool DROP, bool STAMP = false>
__global__ __launch_bounds__(MMT_FUSED_THREADS) void attn_bwd_fused16_kernel(
        const bf16* __restrict__ Qr, const bf16* __restrict__ Kr, const bf16* __restrict__ Vr, const bf16* __restrict__ dOr,
        const float* __restrict__ lse, const float* __restrict__ delta, const float* __restrict__ rowmask, float scale,
        bf16* __restrict__ dqkv, int lddkv,     // row-major [M][lddkv]: dQ at column 0, dK at HD, dV at 2*HD
        int h, int T, int nt, const uint16_t* __restrict__ maskK, float drop_scale) {
    constexpr int DKP = 16, PR = 64, RT = MMT_FUSED_RT_PIECES, PC = 8, TOTAL = MMT_FUSED_STAGE_PIECES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* const stage0 = reinterpret_cast<bf16*>(smem);                                 // [3][TOTAL * 8] bf16: ring of query tiles
    const bf16* const zeros = reinterpret_cast<const bf16*>(smem + 3 * TOTAL * 16);     // what the padding feature rows of an A fragment read
    char* const kvl0 = smem + 3 * TOTAL * 16 + MMT_FUSED_ZERO_BYTES;                    // [NW][2][RT * 16]: own K and V tiles (R layout, padded groups)
    char* const reg0 = kvl0 + MMT_FUSED_NW * 2 * RT * 16;                               // [NW][2][REGION]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int kt = wave;
    const bool live = kt < nt;                          // idle waves stage, synchronise and take their share of the dQ reduction
    const int bh = blockIdx.x, b = bh / h, head = bh - b * h;
    const int Tp = nt * 32, HD = h * DKP;
    const size_t offR = (size_t)bh * fragR_elems(Tp, DKP);
    // dropout: this lane's words of the wave's row of mask blocks (attn_mask.h, LK layout: key on the lane), one per query tile, a tile ahead
    const uint16_t* mrow = maskK + ((size_t)bh * nt + (live ? kt : 0)) * nt * 64 + lane;
    uint32_t mw = DROP ? mrow[0] : 0u;
    const uint32_t scale_bits = __builtin_bit_cast(uint32_t, drop_scale);

    // ---- staging ring: thread p < 144 moves piece p of every tile (segments: Q, dO in the R layout, L, delta)
    const bf16* ssrc; int sstride, sdst; const bool son = tid < MMT_FUSED_STAGE_LOADS;
    {
        const bf16* base[4] = {Qr + offR, dOr + offR,
                               reinterpret_cast<const bf16*>(lse + (size_t)bh * Tp), reinterpret_cast<const bf16*>(delta + (size_t)bh * Tp)};
        const int pieces[4] = {PR, PR, PC, PC}, strides[4] = {32 * DKP, 32 * DKP, 64, 64}, lds0[4] = {0, RT, 2 * RT, 2 * RT + PC};
        int acc = 0; ssrc = base[0]; sstride = 0; sdst = 0;
#pragma unroll
        for (int sg = 0; sg < 4; ++sg) {
            if (tid >= acc && tid < acc + pieces[sg]) {
                const int q = tid - acc;
                ssrc = base[sg] + (size_t)q * 8; sstride = strides[sg];
                sdst = (lds0[sg] + q + (sg < 2 ? (q >> 5) * (MMT_TR_OCT - 32) : 0)) * 8;
            }
            acc += pieces[sg];
        }
    }
    bf16x8 sreg;
    auto stage_load = [&](int tile) { if (son) sreg = *reinterpret_cast<const bf16x8*>(ssrc + (size_t)tile * sstride); };
    auto stage_store = [&](int buf) {
        int to = sdst;
        asm volatile("" : "+v"(to));
        if (son) *reinterpret_cast<bf16x8*>(stage0 + (size_t)buf * TOTAL * 8 + to) = sreg;
    };
    if (tid < MMT_FUSED_ZERO_BYTES / 4) reinterpret_cast<unsigned*>(smem + 3 * TOTAL * 16)[tid] = 0u;
    stage_load(0);

    // ---- per-wave constants
    char* const myreg = reg0 + wave * 2 * MMT_FUSED_REGION_BYTES;
    char* const mykv = kvl0 + wave * (2 * RT * 16) + (hh * MMT_TR_OCT + r) * 16;
    {
        const int ktc = live ? kt : 0;
        const size_t off = ((size_t)(ktc * (DKP / 8) + hh) * 32 + r) * 8;
        *reinterpret_cast<bf16x8*>(mykv) = *reinterpret_cast<const bf16x8*>(Kr + offR + off);
        *reinterpret_cast<bf16x8*>(mykv + RT * 16) = *reinterpret_cast<const bf16x8*>(Vr + offR + off);
        if (!live) {                                    // an idle wave's partials are zero forever
#pragma unroll
            for (int par = 0; par < 2; ++par)
#pragma unroll
                for (int j = 0; j < 8; ++j) reinterpret_cast<float*>(myreg + par * MMT_FUSED_REGION_BYTES)[j * MMT_FUSED_PART_LD + lane] = 0.f;
        }
    }
    f32x16 acc;                                         // rows 0..15: dV^T, rows 16..31: dK^T (see attn_bwd_dkv_kernel)
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = 0.f;
    const bool key_tail = live && (kt == nt - 1) && (T & 31);
    const uint32_t kcol = (uint32_t)(kt * 32 + r);
    // dQ reduction: the 512 outputs (32 queries x 16 features) of a tile: lanes i and 32 + i of wave w both own query 2w + (i >> 4),
    // feature i & 15; the lower lane sums the partials of waves 0..7, the upper lane those of waves 8..15, one half-wave exchange
    // (v_permlane32_swap) adds the two, and the upper lane stores (32 contiguous bytes per query row).  Partial word of (e, q):
    // register (e&3) + 4*(e>>3), lane q + 32*((e>>2)&1).
    const int oe = lane & 15, oq = 2 * wave + ((lane >> 4) & 1);
    const int poff = ((oe & 3) + 4 * (oe >> 3)) * MMT_FUSED_PART_LD + oq + 32 * ((oe >> 2) & 1) + hh * 8 * (2 * MMT_FUSED_REGION_BYTES / 4);
    const float* const pbase = reinterpret_cast<const float*>(reg0) + poff;
    // its destination is linear in the tile index: element offset from dqkv and the step per tile
    const uint32_t m0 = (uint32_t)b * (uint32_t)T;
    uint32_t doff = (m0 + oq) * (uint32_t)lddkv + head * DKP + oe;
    uint32_t rmoff = m0 + oq;

    // Pipeline.  The staged ring is three tiles deep: tile t+2 is fetched during tile t, so tile t+1 is already visible while tile t
    // is processed, and each wave issues the score products S(t+1), dP(t+1) BEFORE the barrier that closes tile t.  After a barrier
    // every wave therefore starts with VALU work on finished MFMA results (and the dQ reduction of the tile just closed) instead of
    // all 16 waves queueing on LDS reads and the MFMA pipe at once — with one workgroup per CU nothing else would fill that bubble.
    stage_store(0);
    if (nt > 1) { stage_load(1); stage_store(1); }
    __syncthreads();

    f32x16 s, dp;                       // S' - L and dP (- delta) of the current tile, produced one tile ahead
    // LDS addresses derived from the lane id are recomputed where they are used, from an opaque copy of it: kept as loop invariants
    // they are what spills at 128 VGPRs, and a scratch reload's vmcnt(0) would also wait for the staging prefetch.
    auto opaque = [](int x) { asm volatile("" : "+v"(x)); return x; };
    auto scores = [&](int buf) {        // row constants (4 consecutive queries per register group) are the accumulator init
        const int lo = opaque(lane), r = lo & 31, hh = lo >> 5;
        char* const mykv = kvl0 + wave * (2 * RT * 16) + (hh * MMT_TR_OCT + r) * 16;
        const bf16* sq = stage0 + (size_t)buf * TOTAL * 8;
        const bf16* sdo = sq + RT * 8;
        const float* sl = reinterpret_cast<const float*>(sq + 2 * RT * 8);
        const float* sd = sl + 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(sl + 8 * g + 4 * hh);
            f32x4 d4 = {0.f, 0.f, 0.f, 0.f};                                   // DROP: -delta enters after the mask
            if (!DROP) d4 = *reinterpret_cast<const f32x4*>(sd + 8 * g + 4 * hh);
#pragma unroll
            for (int i = 0; i < 4; ++i) { s[4 * g + i] = l4[i]; dp[4 * g + i] = d4[i]; }      // both stored negated
        }
        const int o8 = (hh * MMT_TR_OCT + r) * 8;
        s = mfma32(*reinterpret_cast<const bf16x8*>(sq + o8), *reinterpret_cast<const bf16x8*>(mykv), s);
        dp = mfma32(*reinterpret_cast<const bf16x8*>(sdo + o8), *reinterpret_cast<const bf16x8*>(mykv + RT * 16), dp);
    };
    if (live) scores(0);
    else {
#pragma unroll
        for (int j = 0; j < 16; ++j) { s[j] = 0.f; dp[j] = 0.f; }
    }
    int cur = 0, nxt = 1, nn = 2;       // ring slots of tiles qt, qt+1, qt+2
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_prev = 0, st_entry = 0;
    (void)st_acc; (void)st_prev; (void)st_entry;
#ifdef MMT_ABLATIONS
    if (STAMP) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev) :: "memory"); st_entry = st_prev; }
#endif

    // dQ reduction of one tile (all 16 partials are in LDS once the tile's barrier has been passed): fixed-order sum (8 per lane as 4
    // packed pairs, then the two half-waves), scale, query-row mask, store.  (Running it one tile late, in the slack in front of the
    // next tile's barrier, was measured: 47.4 -> 49.2 us per launch; it stays directly behind its own barrier.)
    auto reduce_dq = [&](int qt, float rm, bool orow) {
        asm volatile("" : "+v"(rm));                    // first use of the mask value long after its load
        const float* pp = pbase + (qt & 1) * (MMT_FUSED_REGION_BYTES / 4);
        f32x2 v2 = {pp[0], pp[2 * MMT_FUSED_REGION_BYTES / 4]};
#pragma unroll
        for (int w2 = 2; w2 < MMT_FUSED_NW / 2; w2 += 2) {
            const f32x2 t = {pp[w2 * (2 * MMT_FUSED_REGION_BYTES / 4)], pp[(w2 + 1) * (2 * MMT_FUSED_REGION_BYTES / 4)]};
            v2 += t;
        }
        const float half = v2[0] + v2[1];
        // v_permlane32_swap a, b: lanes 32..63 of a <-> lanes 0..31 of b.  Written as asm: hipcc 7.2 folded the second result of
        // __builtin_amdgcn_permlane32_swap into the first here (v_add v, v, v).  The s_nop covers the VALU-write -> permlane hazard.
        float lo8 = half, hi8 = half;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo8), "+v"(hi8));      // every lane: lo8 = waves 0..7, hi8 = waves 8..15
        const float v = (lo8 + hi8) * ((rm == 0.0f) ? 0.f : scale);     // blanked query rows pass no gradient to Q
        if ((opaque(lane) >> 5) && orow) dqkv[doff] = (bf16)v;
        doff += 32u * (uint32_t)lddkv;
    };
    auto body = [&](auto tail_tag, auto next_tag, int qt) {
        constexpr bool QTAIL = decltype(tail_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value;                        // a tile qt+1 exists
        const bool more = qt + 2 < nt;
        if (more) stage_load(qt + 2);
        // the mask value of this lane's output row, fetched a tile-time before use
        const bool orow = !QTAIL || (qt * 32 + oq) < T;
        float rm = rowmask ? rowmask[orow ? rmoff : m0] : 1.f;
        char* const region = myreg + (qt & 1) * MMT_FUSED_REGION_BYTES;
        FB_STAMP(0);                                    // loop top: prefetch issue
        if (live) {
            const uint32_t tw = mw;
            if (DROP && NEXT) mw = mrow[(size_t)(qt + 1) * 64];
            const bf16* const sqc = stage0 + (size_t)cur * TOTAL * 8;                       // this tile's Q (then dO, L, delta)
            const float* sd = reinterpret_cast<const float*>(sqc + 2 * RT * 8) + 32;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                float pv = fast_exp2(s[j]);
                if (QTAIL) pv = (qt * 32 + acc32_row(j, hh) < T) ? pv : 0.f;  // queries >= T do not exist
                s[j] = pv;
            }
            if (key_tail) {                             // wave-uniform, loop-invariant: only the last key tile's wave pays
                const float kmul = ((int)kcol < T) ? 1.f : 0.f;        // keys >= T do not exist
#pragma unroll
                for (int j = 0; j < 16; ++j) s[j] *= kmul;
            }
            if (DROP) {
                static_for<0, 4>([&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    const f32x4 d4 = *reinterpret_cast<const f32x4*>(sd + 8 * g + 4 * hh);
                    static_for<0, 4>([&](auto ic) {
                        constexpr int i = decltype(ic)::value, j = 4 * g + i;
                        const float ms = __builtin_bit_cast(float, scale_bits & keep_bits<j>(tw));     // 1/(1-p) where (query of register j, this lane's key) was kept
                        dp[j] = s[j] * fmaf(dp[j], ms, d4[i]);
                        s[j] *= ms;
                    });
                });
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) dp[j] *= s[j];
            }
            FB_STAMP(1);                                // exponentials, dropout, dS
            // dV^T / dK^T, and dS into the patch on the way.  A fragments by transposing reads of the staged dO / Q tiles: the lane
            // SUPPLIES the address of query tq of a 4-query block, features 4 tpp .. + 3, and receives feature (lane & 15); fragment slot j
            // <-> query 16 s2 + 8 (j >> 2) + 4 hh + (j & 3), the row order of the P / dS accumulators.  dV^T lives in accumulator rows
            // 0..15 (lanes r < 16 read dO, the others zeros), dK^T in rows 16..31 (lanes r >= 16 read Q, the others zeros).
            const int lo = opaque(lane), r = lo & 31, hh = lo >> 5;
            const int tq = (lo >> 2) & 3, tpp = lo & 3, toff = (tpp >> 1) * (MMT_TR_OCT * 8) + 4 * (tpp & 1) + tq * 8;
            const bool up = (lo >> 4) & 1;
            const bf16* const ado = up ? zeros : sqc + RT * 8 + toff + 32 * hh;
            const bf16* const aq = up ? sqc + toff + 32 * hh : zeros;
            bf16* const patch = reinterpret_cast<bf16*>(region);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pds = pack8(dp, s2);
                acc = mfma32(tr_frag2(ado + 128 * s2, ado + 128 * s2 + 64), pack8(s, s2), acc);
                acc = mfma32(tr_frag2(aq + 128 * s2, aq + 128 * s2 + 64), pds, acc);
                // patch row = this lane's key; slots [hh][8 s2 + j] = query acc32_row(8 s2 + j, hh): every aligned group of 4 slots
                // is 4 consecutive queries, the unit a transposing read hands out
                *reinterpret_cast<bf16x8*>(patch + r * MMT_FUSED_PATCH_LD + 16 * hh + 8 * s2) = pds;
            }
            FB_STAMP(2);                                // packs, dV/dK products, patch writes
            // dQ^T share of this key tile: K^T (features x keys) times dS^T (keys x queries).  K^T fragments by transposing reads of
            // the own K tile: slot j <-> key 16 s2 + 8 hh + j, the patch's column order; lanes r >= 16 would produce the padding
            // feature rows, which nobody reads: they supply (and receive) the same as lanes r - 16
            const bf16* const ak = reinterpret_cast<const bf16*>(kvl0 + wave * (2 * RT * 16)) + toff + 64 * hh;
            // dS^T fragments: this lane supplies key tq of a 4-key block and the 4 queries 16 up + 4 tpp .. + 3 (patch slots
            // [tpp & 1][4 (2 up + (tpp >> 1)) ..]) and receives its own query r for those keys; slot j <-> key 16 s2 + 8 hh + j
            const bf16* const pb = patch + (8 * hh + tq) * MMT_FUSED_PATCH_LD + 16 * (tpp & 1) + 4 * (2 * (int)up + (tpp >> 1));
            f32x16 dqp;
#pragma unroll
            for (int j = 0; j < 16; ++j) dqp[j] = 0.f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 bfrag = tr_frag2(pb + 16 * MMT_FUSED_PATCH_LD * s2, pb + 16 * MMT_FUSED_PATCH_LD * s2 + 4 * MMT_FUSED_PATCH_LD);
                dqp = mfma32(tr_frag2(ak + 128 * s2, ak + 128 * s2 + 32), bfrag, dqp);
            }
            float* const part = reinterpret_cast<float*>(region);
#pragma unroll
            for (int j = 0; j < 8; ++j) part[j * MMT_FUSED_PART_LD + lo] = dqp[j];          // feature rows >= 16 are padding
            FB_STAMP(3);                                // patch reads, dQ product, partial writes
            if (NEXT) scores(nxt);                      // tile qt+1 has been visible since the previous barrier
            FB_STAMP(4);                                // next tile's row constants, operand reads and score products
        }
        if (more) stage_store(nn);
        __syncthreads();
        FB_STAMP(5);                                    // staging store + barrier
        reduce_dq(qt, rm, orow);
        FB_STAMP(6);                                    // dQ reduction and store
        rmoff += 32u;
        const int t3 = cur; cur = nxt; nxt = nn; nn = t3;
    };
    for (int qt = 0; qt < nt - 1; ++qt) body(std::false_type{}, std::true_type{}, qt);
    if (T & 31) body(std::true_type{}, std::false_type{}, nt - 1); else body(std::false_type{}, std::false_type{}, nt - 1);
#ifdef MMT_ABLATIONS
    if (STAMP && g_attn_stamps && lane == 0) {
        unsigned long long t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        unsigned long long* q = g_attn_stamps + ((size_t)blockIdx.x * MMT_FUSED_NW + wave) * 16;
        for (int i = 0; i < 7; ++i) q[i] = st_acc[i];
        q[7] = t1 - st_entry; q[9] = live ? 1 : 2;
    }
#endif
    if (!live) return;
    // dK = ln2 * acc rows 16.. (scores are in the log2 domain), dV = acc rows 0..15; column key = r
    const float LN2 = 0.6931471805599453f;
    const int t = kt * 32 + r;
    if (t < T) {
        const size_t m = (size_t)m0 + t;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bf16x4 kv, vv;
#pragma unroll
            for (int j = 0; j < 4; ++j) { kv[j] = (bf16)(acc[8 + 4 * g + j] * LN2); vv[j] = (bf16)acc[4 * g + j]; }
            const int e0 = head * DKP + 8 * g + 4 * hh;
            *reinterpret_cast<bf16x4*>(dqkv + m * lddkv + HD + e0) = kv;
            *reinterpret_cast<bf16x4*>(dqkv + m * lddkv + 2 * HD + e0) = vv;
        }
    }
}
