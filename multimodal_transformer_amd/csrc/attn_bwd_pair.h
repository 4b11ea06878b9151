// Attention backward in ONE kernel for d_k <= 16 and 9..16 key tiles (T = 257..512: configs[1], configs[3]): dQ, dK and dV from a
// single evaluation of P = 2^(S'-L) and dS per score.  Round 5 form: 8 waves per (batch, head), each wave owning TWO adjacent key tiles.
//
// What round 4's kernel (16 waves x one key tile, attn_bwd_diag.h) spent its time on, by its own stamps and counters: the vector phase
// 40 %, the barrier 23 %, the dQ read-add-write in LDS 14 %, LDS instructions active 83 % of the CU's time.  Priced with the LDS table of
// MI355X_MICROARCH.md a (query tile, key tile) pair cost it ~154 LDS-array cycles (row constants 32, operand fragments 16 + 16, patch
// 26 + 8, K^T 8, dQ accumulate 48) — 39 k cycles per head, as much as its vector work (tools/valu_micro.hip, round 5: exp 8 cycles,
// v_bfe / v_cvt_pk / three-operand forms 4.2, plain fp32 / logic 2.3 per wave64 instruction and SIMD: ~475 cycles per pair and wave, 30 k
// per SIMD and head).  Two key tiles per wave share everything that belongs to the QUERY tile of a step:
//   * the row constants (L as the accumulator init of both score products still costs a read per key tile; delta is read once),
//   * the Q' / dO operand fragments of the score products and the transposed dO^T / Q'^T fragments of the dV / dK products,
//   * ONE dQ^T share per step (the two key tiles' products accumulate in the same registers): half the read-add-write traffic and half
//     the participants of every barrier;
// and what belongs to the KEY tiles lives in registers for the whole sweep (K, V score operands and the transposed K^T fragments of the
// dQ product: 32 VGPRs; round 4 re-read them from LDS every step).  ~87 LDS-array cycles per pair instead of ~154.
//
// Vector work per score (train mode), with the factor 1/(1-p) = c folded into the three output scales and delta pre-divided by c while
// it is staged:   P = exp2(s);  m = bfe(word, j);  Pm = P & m;  t = P * (-delta/c);  dS/c = fma(Pm, dP, t)   and one pack each for Pm, dS:
// exp 8 + bfe 4.2 + and 2.3 + mul 2.3 + fma 2.4 + cvt 4.2 = 23.4 cycles of SIMD issue (round 4: and + fma + two mul + bfe: 25.7, plus
// the scale's AND).  dV = c Pm^T dO,  dK = c ln2 (dS/c)^T Q',  dQ = c scale (dS/c) K.
//
// Sweep: DIAGONAL, and without workgroup barriers.  The wave that owns key-tile pair p works at step t on query tile (2 p + t) mod ntp
// (ntp = nt rounded up to even; an odd nt gets one all-blank query tile), so the live waves are on different query tiles and each fp32
// dQ^T accumulator in LDS is touched by one wave per step: a plain read-add-write, no atomics.  Who touches a tile before whom is
// fixed: the tile of (p, t) was last touched by (p + 1, t - 2) — TWO steps earlier — so instead of a barrier per step (round 4; the
// two-barrier complementary-phase form of this kernel measured 38.7 us: every interval lasted as long as its slower phase) wave p
// only checks, before its read-add-write of step t, a progress word that wave p + 1 publishes after each of its steps (LDS operations of
// one wave execute in order: the word is written behind the data).  The check practically never waits, the order of the additions
// stays fixed (bit-reproducible), and the two waves of a SIMD drift into complementary phases by themselves: the matrix phase of a
// step runs at raised priority (a handful of MFMA / LDS instructions that must not queue behind the other wave's vector stream),
// the vector phase at priority 0.
// The accumulators are zero-filled by the prologue (no first-step variant), 16 bytes per lane and register group (ds_read_b128 /
// ds_write_b128).  The lane's dropout words are fetched TWO steps ahead (a step ahead, the second word of a step stalled the vector
// phase for ~750 cycles: stamps of the barrier form).
// LDS: all query tiles' Q', dO (R layout, 8-feature groups 576 bytes apart for the transposing reads), L, delta: 40 KB; two dS patches
// per wave: 40 KB; fp32 dQ^T accumulators: 32 KB; zeros, progress words: 1 KB.  114 KB.
//
// Reference semantics: transformer/MFT/multiTransformer.py:22-34 (scaled dot-product attention with dropout on the probabilities) under autograd.
#pragma once
#include "attn.h"

#define MMT_PAIR_NW 8
#define MMT_PAIR_THREADS (MMT_PAIR_NW * 64)
#define MMT_PAIR_MAXT 16                                     // query / key tiles per head
#define MMT_PAIR_PATCH_LD 40                                 // bf16 per patch row (one key): 2 halves x 16 accumulator slots + 8 pad (80-byte rows: conflict-free b128 writes)
#define MMT_PAIR_PATCH_BYTES (32 * MMT_PAIR_PATCH_LD * 2)    // 2560 per key tile
#define MMT_PAIR_ACC_FLOATS 512                              // per query tile: [2 register groups][64 lanes][4]: feature rows < 16 of a 32x32 accumulator
#define MMT_PAIR_RT_PIECES (2 * MMT_TR_OCT)                  // an R tile of d_k = 16 in LDS: two 8-feature groups, 576 bytes apart
#define MMT_PAIR_QD_PIECES (2 * MMT_PAIR_RT_PIECES + 16)     // LDS pieces per query tile: Q', dO, 8 + 8 pieces of row constants
#define MMT_PAIR_QD_LOADS 144                                // 16-byte pieces fetched per query tile: 2 * 64 + 2 * 8
#define MMT_PAIR_ZERO_BYTES 512
#define MMT_PAIR_FLAG_BYTES 512                               // progress words, one per wave, 64 bytes apart
#define MMT_PAIR_LDS_BYTES (MMT_PAIR_MAXT * MMT_PAIR_QD_PIECES * 16 + MMT_PAIR_ZERO_BYTES + MMT_PAIR_FLAG_BYTES + MMT_PAIR_NW * 2 * MMT_PAIR_PATCH_BYTES \
                            + MMT_PAIR_MAXT * MMT_PAIR_ACC_FLOATS * 4)      // 40,960 + 512 + 512 + 40,960 + 32,768 = 115,712

__host__ inline bool attn_bwd_fused_ok(int DKP, int nt) { return DKP == 16 && nt > 8 && nt <= MMT_PAIR_MAXT; }

template <bool DROP>
__global__ __launch_bounds__(MMT_PAIR_THREADS) void attn_bwd_pair16_kernel(
        const bf16* __restrict__ Qr, const bf16* __restrict__ Kr, const bf16* __restrict__ Vr, const bf16* __restrict__ dOr,
        const float* __restrict__ lse, const float* __restrict__ delta, const float* __restrict__ rowmask, float scale,
        bf16* __restrict__ dqkv, int lddkv,     // row-major [M][lddkv]: dQ at column 0, dK at HD, dV at 2*HD
        int h, int T, int nt, const uint16_t* __restrict__ maskK, float drop_scale) {
    //@ entry
    constexpr int DKP = 16, RT = MMT_PAIR_RT_PIECES, TOTAL = MMT_PAIR_QD_PIECES, NW = MMT_PAIR_NW, PLD = MMT_PAIR_PATCH_LD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* const qd0 = reinterpret_cast<bf16*>(smem);                                        // [MAXT][TOTAL * 8] bf16: every query tile of the head
    const bf16* const zeros = reinterpret_cast<const bf16*>(smem + MMT_PAIR_MAXT * TOTAL * 16);   // what the padding feature rows of an A fragment read
    // [NW] steps completed, 16 ints apart.  LDS-typed pointers: through a generic pointer the volatile accesses become flat_load / flat_store,
    // which count in vmcnt too, and every poll then waited for the dropout words in flight (603 cycles per step in the stamps)
    typedef __attribute__((address_space(3))) int lds_int;
    volatile lds_int* const flags = (volatile lds_int*)(smem + MMT_PAIR_MAXT * TOTAL * 16 + MMT_PAIR_ZERO_BYTES);
    char* const patch0 = smem + MMT_PAIR_MAXT * TOTAL * 16 + MMT_PAIR_ZERO_BYTES + MMT_PAIR_FLAG_BYTES;     // [NW][2][PATCH_BYTES]: wave-private dS patches
    float* const dqacc0 = reinterpret_cast<float*>(patch0 + NW * 2 * MMT_PAIR_PATCH_BYTES); // [MAXT query tiles][ACC_FLOATS] fp32
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    // key-tile pairs are dealt to the waves so that the live ones spread over both phase groups (waves 0..3 / 4..7, below) when nt < 16
    const int pr = ((wave & 3) << 1) | (wave >> 2);
    const int ktA = 2 * pr, ktB = 2 * pr + 1;
    const bool live = ktA < nt;                         // idle waves take part in the staging, the barriers and the dQ output only
    const bool haveB = ktB < nt;                        // odd nt: the last live wave owns one key tile; its second one is all "keys >= T"
    const int ktAc = live ? ktA : 0, ktBc = haveB ? ktB : ktAc;         // tiles whose operands are fetched (always inside the head)
    const int bh = blockIdx.x, b = bh / h, head = bh - b * h;
    const int Tp = nt * 32, HD = h * DKP;
    const int npairs = (nt + 1) >> 1, ntp = 2 * npairs;  // steps of the sweep = query tiles incl. the blank one of an odd nt
    const size_t offR = (size_t)bh * fragR_elems(Tp, DKP);
    const float c_out = DROP ? drop_scale : 1.0f;       // 1/(1-p): applied to dQ, dK, dV once, at the end

    // ---- prologue.  Own key tiles: the score products' B fragments straight from memory (K[key r][features 8 hh ..]: 16 bytes per
    // lane), kept for the whole sweep; K also goes through the wave's patch area once, to come back transposed (below).
    bf16x8 kfA, kfB, vfA, vfB;
    {
        const size_t offA = ((size_t)(ktAc * (DKP / 8) + hh) * 32 + r) * 8, offB = ((size_t)(ktBc * (DKP / 8) + hh) * 32 + r) * 8;
        kfA = *reinterpret_cast<const bf16x8*>(Kr + offR + offA); vfA = *reinterpret_cast<const bf16x8*>(Vr + offR + offA);
        kfB = *reinterpret_cast<const bf16x8*>(Kr + offR + offB); vfB = *reinterpret_cast<const bf16x8*>(Vr + offR + offB);
    }
    // dropout: this lane's words of the two rows of mask blocks (attn_mask.h, LK layout: key on the lane), one per query tile, a step ahead
    const uint16_t* const mrowA = maskK + ((size_t)bh * nt + ktAc) * nt * 64 + lane;
    const uint16_t* const mrowB = maskK + ((size_t)bh * nt + ktBc) * nt * 64 + lane;
    int qt = live ? ktA : 0;                            // the wave's query tile of step t: (2 pr + t) mod ntp
    // (the blank tile of an odd nt reads the words of tile nt - 1: its probabilities are 0 whatever the mask says)
    auto mask_words = [&](int tile, uint32_t& wa, uint32_t& wb) {
        const size_t o = (size_t)(tile < nt ? tile : nt - 1) * 64;
        wa = mrowA[o]; wb = mrowB[o];
    };
    uint32_t mwA = 0u, mwB = 0u, mwA1 = 0u, mwB1 = 0u;  // this step's and the next step's words; the fetch runs two steps ahead
    if (DROP) { mask_words(qt, mwA, mwB); mask_words(qt + 1 == ntp ? 0 : qt + 1, mwA1, mwB1); }
    {
        // every query tile of the head -> LDS: thread p moves pieces p, p + 512, ... of the nt * 144
        const bf16* const lsrc = reinterpret_cast<const bf16*>(lse + (size_t)bh * Tp);
        const bf16* const dsrc = reinterpret_cast<const bf16*>(delta + (size_t)bh * Tp);
        const int npieces = ntp * MMT_PAIR_QD_LOADS;      // odd nt: tile nt is staged blank (Q' = dO = 0, L = -inf: P = 0 for all its queries)
        constexpr int NL = (MMT_PAIR_MAXT * MMT_PAIR_QD_LOADS + MMT_PAIR_THREADS - 1) / MMT_PAIR_THREADS;     // 5
        const float inv_c = DROP ? 1.0f / drop_scale : 1.0f;
        bf16x8 reg[NL]; int dst[NL]; int kind[NL];      // kind: 0 operand piece, 1 + first query: an L piece, -1 a delta piece
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int p = tid + i * MMT_PAIR_THREADS;
            dst[i] = -1; kind[i] = 0;
            if (p < npieces) {
                const int tile = p / MMT_PAIR_QD_LOADS, q = p - tile * MMT_PAIR_QD_LOADS;
                const bf16* src; int d;
                if (q < 64) { src = Qr + offR + (size_t)tile * 32 * DKP + q * 8; d = q + (q >> 5) * (MMT_TR_OCT - 32); }
                else if (q < 128) { src = dOr + offR + (size_t)tile * 32 * DKP + (q - 64) * 8; d = RT + (q - 64) + ((q - 64) >> 5) * (MMT_TR_OCT - 32); }
                else if (q < 136) { src = lsrc + (size_t)tile * 64 + (q - 128) * 8; d = 2 * RT + (q - 128); kind[i] = 1 + tile * 32 + (q - 128) * 4; }
                else { src = dsrc + (size_t)tile * 64 + (q - 136) * 8; d = 2 * RT + 8 + (q - 136); kind[i] = -1; }
                if (tile < nt) reg[i] = *reinterpret_cast<const bf16x8*>(src);
                else reg[i] = __builtin_bit_cast(bf16x8, f32x4{0.f, 0.f, 0.f, 0.f});       // (its L pieces become -inf below: their queries are >= T)
                dst[i] = (tile * TOTAL + d) * 8;
            }
        }
        if (tid < (MMT_PAIR_ZERO_BYTES + MMT_PAIR_FLAG_BYTES) / 4) reinterpret_cast<unsigned*>(smem + MMT_PAIR_MAXT * TOTAL * 16)[tid] = 0u;
        {   // dQ^T accumulators start at zero: nt * 512 floats, 16 bytes per thread and round
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            for (int i = tid; i < ntp * (MMT_PAIR_ACC_FLOATS / 4); i += MMT_PAIR_THREADS) reinterpret_cast<f32x4*>(dqacc0)[i] = z4;
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (dst[i] < 0) continue;
            if (kind[i] > 0) {                           // -L of queries >= T: -inf, so that P = 2^(S' - L) = 0 exactly for them
                f32x4 v = __builtin_bit_cast(f32x4, reg[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (kind[i] - 1 + e < T) ? v[e] : -INFINITY;
                reg[i] = __builtin_bit_cast(bf16x8, v);
            } else if (DROP && kind[i] < 0) {            // -delta / c (see the head comment)
                f32x4 v = __builtin_bit_cast(f32x4, reg[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= inv_c;
                reg[i] = __builtin_bit_cast(bf16x8, v);
            }
            *reinterpret_cast<bf16x8*>(qd0 + dst[i]) = reg[i];
        }
    }
    // transposing-read role of this lane (common.h tr_frag2): it SUPPLIES the address of window tq of a 4-window block, features
    // 4 tpp .. + 3, and receives feature (lane & 15); fragment slot j <-> window 16 s2 + 8 (j >> 2) + 4 hh + (j & 3), the row order of
    // the P / dS accumulators
    const int tq = (lane >> 2) & 3, tpp = lane & 3, toff = (tpp >> 1) * (MMT_TR_OCT * 8) + 4 * (tpp & 1) + tq * 8;
    const bool up = (lane >> 4) & 1;
    char* const mypatch = patch0 + wave * (2 * MMT_PAIR_PATCH_BYTES);
    // K^T fragments of the dQ product (features x keys; slot j <-> key 16 s2 + 8 hh + j, the patch's column order): each K tile is
    // parked in the wave's (still unused) patch area in the padded R layout and read back transposed, once.  Lanes r >= 16 would
    // produce the padding feature rows, which nobody reads: they supply (and receive) the same as lanes r - 16.
    bf16x8 akA[2], akB[2];
    {
        bf16* const kl = reinterpret_cast<bf16*>(mypatch);
        *reinterpret_cast<bf16x8*>(kl + (hh * MMT_TR_OCT + r) * 8) = kfA;
        *reinterpret_cast<bf16x8*>(kl + (RT + hh * MMT_TR_OCT + r) * 8) = kfB;
        const bf16* const a = kl + toff + 64 * hh;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            akA[s2] = tr_frag2(a + 128 * s2, a + 128 * s2 + 32);
            akB[s2] = tr_frag2(a + RT * 8 + 128 * s2, a + RT * 8 + 128 * s2 + 32);
        }
    }

    f32x16 accA, accB;                                  // rows 0..15: dV^T, rows 16..31: dK^T of key tiles A and B (the d_k = 16 trick of attn_bwd_dkv_kernel)
#pragma unroll
    for (int j = 0; j < 16; ++j) { accA[j] = 0.f; accB[j] = 0.f; }
    const bool tailA = live && (ktA == nt - 1) && (T & 31);
    const bool tailB = live && (!haveB || ((ktB == nt - 1) && (T & 31)));
    const float kbiasA = (ktA * 32 + r < T) ? 0.f : -INFINITY, kbiasB = (haveB && ktB * 32 + r < T) ? 0.f : -INFINITY;      // keys >= T do not exist
    const uint32_t m0 = (uint32_t)b * (uint32_t)T;
    __syncthreads();                                    // the staged tiles, the zeroed accumulators; the K tiles in the patch area have been read back
    //@ staged

    f32x16 sA, dpA, sB, dpB;                            // S' - L and dP (- delta) of the wave's next query tile, produced one step ahead
    f32x4 dl[4];                                        // train mode: -delta / c of that tile's queries, in accumulator row order
    // score products of query tile `tile` against both key tiles; row constants (4 consecutive queries per register group) are the accumulator init
    bf16x8 qf_n, dof_n;
    auto score_operands = [&](int tile) {
        const bf16* sq = qd0 + (size_t)tile * TOTAL * 8;
        const bf16* sdo = sq + RT * 8;
        const float* sl = reinterpret_cast<const float*>(sq + 2 * RT * 8);
        const float* sd = sl + 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(sl + 8 * g + 4 * hh);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(sd + 8 * g + 4 * hh);
            dl[g] = d4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                       // both stored negated; DROP: -delta / c enters behind the mask
                sA[4 * g + i] = l4[i]; sB[4 * g + i] = l4[i];
                dpA[4 * g + i] = DROP ? 0.f : d4[i]; dpB[4 * g + i] = DROP ? 0.f : d4[i];
            }
        }
        const int o8 = (hh * MMT_TR_OCT + r) * 8;
        qf_n = *reinterpret_cast<const bf16x8*>(sq + o8); dof_n = *reinterpret_cast<const bf16x8*>(sdo + o8);
    };
    auto score_products = [&]() {
        sA = mfma32(qf_n, kfA, sA);
        dpA = mfma32(dof_n, vfA, dpA);
        sB = mfma32(qf_n, kfB, sB);
        dpB = mfma32(dof_n, vfB, dpB);
    };
    auto scores = [&](int tile) { score_operands(tile); score_products(); };
    if (live) scores(qt);
    else {
#pragma unroll
        for (int j = 0; j < 16; ++j) { sA[j] = 0.f; dpA[j] = 0.f; sB[j] = 0.f; dpB[j] = 0.f; }
#pragma unroll
        for (int g = 0; g < 4; ++g) dl[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // the vector phase of one key tile: s -> P (dropped: Pm), dp -> dS (/ c), both packed to MFMA operand fragments
    auto vector_phase = [&](const int which, f32x16& s, f32x16& dp, const uint32_t tw, const bool tail, const float kbias, bf16x8 (&pm)[2], bf16x8 (&ds)[2]) {
        if (tail) {                                     // wave-uniform, loop-invariant: only the wave of the last key tile pays.  A bias in FRONT of
            // the exponential: 2^(s - inf) = 0 whatever s is (a factor 0 behind it would turn an overflow into NaN).  The empty asm keeps
            // the block a branch: hipcc otherwise computes the sums speculatively in EVERY wave (16 v_pk_add_f32 + selects per step)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) s[j] += kbias;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) s[j] = fast_exp2(s[j]);                    // queries >= T: exactly 0 (their L was staged as -inf)
        //@ vx_exp[which]
        if (DROP) {
            static_for<0, 16>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                // (the empty asm statements keep hipcc's SLP pass from pairing neighbouring registers into v_pk_mul_f32 / v_pk_fma_f32, which
                // cost a SIMD more beside MFMAs than the two plain instructions they replace: MI355X_MICROARCH.md "price of one filler beside
                // MFMAs", tools/pair_micro.hip 7.2 against 5.2 cycles; -fno-slp-vectorize on the whole library: this kernel 36.1 -> 35.4 us)
                float t = s[j] * dl[j >> 2][j & 3];                             // P (-delta / c)
                asm("" : "+v"(t));
                s[j] = keep_and<j>(s[j], tw);                                   // Pm: P where (query of register j, this lane's key) was kept
                float ds_ = fmaf(s[j], dp[j], t);                               // dS / c = Pm dP - P delta / c
                asm("" : "+v"(ds_));
                dp[j] = ds_;
            });
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) dp[j] *= s[j];
        }
        //@ vx_ds[which]
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { pm[s2] = pack8(s, s2); ds[s2] = pack8(dp, s2); }
        //@ vx_pack[which]
    };

    bf16* const patchA = reinterpret_cast<bf16*>(mypatch), * const patchB = reinterpret_cast<bf16*>(mypatch + MMT_PAIR_PATCH_BYTES);
    // dS^T fragments of the dQ product: this lane supplies key tq of a 4-key block and the 4 queries 16 up + 4 tpp .. + 3 (patch slots
    // [tpp & 1][4 (2 up + (tpp >> 1)) ..]) and receives its own query r for those keys; slot j <-> key 16 s2 + 8 hh + j
    const int pboff = (8 * hh + tq) * PLD + 16 * (tpp & 1) + 4 * (2 * (int)up + (tpp >> 1));
    const int pwoff = r * PLD + 16 * hh;                // patch row = this lane's key; slots [hh][8 s2 + j] = query acc32_row(8 s2 + j, hh)

    // A wave's step: a vector phase X (exponentials, mask, dS, packs: no MFMA, no LDS) and a matrix phase Y (12 product MFMAs, the
    // patch round trip, the dQ read-add-write, the next tile's 4 score MFMAs and their LDS reads: hardly any vector instruction).
    bf16x8 pmA[2], dsA[2], pmB[2], dsB[2];
    const int pnext = (pr + 1 == npairs) ? 0 : pr + 1;  // the pair whose wave touched this step's dQ accumulator two steps ago
    const int wnext = (pnext >> 1) | ((pnext & 1) << 2);
    volatile lds_int* const flag_in = flags + 16 * wnext;
    volatile lds_int* const flag_out = flags + 16 * wave;
    auto body = [&](auto next_tag, auto final_tag, const int t) {
        constexpr bool NEXT = decltype(next_tag)::value;  // a step t + 1 exists (compile-time: no skippable block between an MFMA and its readers)
        constexpr bool FINAL = decltype(final_tag)::value;  // one of the last two steps: this addition completes the query tile's dQ
        const int qn = (qt + 1 == ntp) ? 0 : qt + 1, qnn = (qn + 1 == ntp) ? 0 : qn + 1;
        // ---- X
        __builtin_amdgcn_s_setprio(0);
        const uint32_t twA = mwA, twB = mwB;
        mwA = mwA1; mwB = mwB1;
        if (DROP) mask_words(qnn, mwA1, mwB1);          // step t + 2's words (past the end: a harmless re-read)
        vector_phase(0, sA, dpA, twA, tailA, kbiasA, pmA, dsA);
        vector_phase(1, sB, dpB, twB, tailB, kbiasB, pmB, dsB);
        //@ x
        // ---- Y
        __builtin_amdgcn_s_setprio(2);
        if (t >= 2) {                                   // the accumulator's previous addition (wave of pair p + 1, step t - 2) must be in LDS
            int polls = 0;
            while (*flag_in < t - 1 && ++polls < (1 << 20)) __builtin_amdgcn_s_sleep(1);
        }
        //@ flag
        const bf16* const sqc = qd0 + (size_t)qt * TOTAL * 8;                   // this tile's Q' (then dO, L, delta)
        f32x4* const slot = reinterpret_cast<f32x4*>(dqacc0 + (size_t)qt * MMT_PAIR_ACC_FLOATS) + lane;      // feature rows >= 16 (registers 8..15) are padding
        const f32x4 o0 = slot[0], o1 = slot[64];        // the accumulator's old value: no other wave touches this tile before this wave has published step t
        *reinterpret_cast<bf16x8*>(patchA + pwoff) = dsA[0];
        *reinterpret_cast<bf16x8*>(patchA + pwoff + 8) = dsA[1];
        *reinterpret_cast<bf16x8*>(patchB + pwoff) = dsB[0];
        *reinterpret_cast<bf16x8*>(patchB + pwoff + 8) = dsB[1];
        // A fragments of the dV^T / dK^T products, by transposing reads of the resident dO / Q' tiles, shared by both key tiles:
        // dV^T lives in accumulator rows 0..15 (lanes r < 16 read dO, the others zeros), dK^T in rows 16..31 (the other way round)
        const bf16* const ado = up ? zeros : sqc + RT * 8 + toff + 32 * hh;
        const bf16* const aq = up ? sqc + toff + 32 * hh : zeros;
        bf16x8 fdo[2], fq[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) { fdo[s2] = tr_frag2(ado + 128 * s2, ado + 128 * s2 + 64); fq[s2] = tr_frag2(aq + 128 * s2, aq + 128 * s2 + 64); }
        if (NEXT) score_operands(qn);                   // the next query tile of this wave: resident in LDS since the prologue
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            accA = mfma32(fdo[s2], pmA[s2], accA);
            accB = mfma32(fdo[s2], pmB[s2], accB);
            accA = mfma32(fq[s2], dsA[s2], accA);
            accB = mfma32(fq[s2], dsB[s2], accB);
        }
        // dS^T fragments back out of the patches (the writes above are long done), then the next tile's score products BEFORE the dQ
        // chain: the next vector phase starts with their results, the dQ chain's result is only needed by this phase's last instructions
        bf16x8 pfa[2], pfb[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16* const pa = patchA + pboff + 16 * PLD * s2;
            const bf16* const pb = patchB + pboff + 16 * PLD * s2;
            pfa[s2] = tr_frag2(pa, pa + 4 * PLD); pfb[s2] = tr_frag2(pb, pb + 4 * PLD);
        }
        if (NEXT) score_products();
        __builtin_amdgcn_sched_barrier(0);
        // dQ^T share of the two key tiles: K^T (features x keys) times dS^T (keys x queries), 64 keys contracted in one accumulator
        f32x16 dqp;
#pragma unroll
        for (int j = 0; j < 16; ++j) dqp[j] = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            dqp = mfma32(akA[s2], pfa[s2], dqp);
            dqp = mfma32(akB[s2], pfb[s2], dqp);
        }
        const f32x4 n0 = {o0[0] + dqp[0], o0[1] + dqp[1], o0[2] + dqp[2], o0[3] + dqp[3]};
        const f32x4 n1 = {o1[0] + dqp[4], o1[1] + dqp[5], o1[2] + dqp[6], o1[3] + dqp[7]};
        if (!FINAL) {
            slot[0] = n0; slot[64] = n1;
            if (lane == 0) *flag_out = t + 1;           // behind the two stores, in this wave's LDS order: step t is published
        } else {
            // every tile's last addition happens in one of the last two steps (tile (2 p + t) mod ntp is touched every other step): the sum
            // goes straight from the registers to memory — no final barrier, no read-back.  The lane holds query r, features 4 hh .. + 3
            // and 8 + 4 hh .. + 3: two 8-byte pieces of the query's 32-byte row
            const int tqr = qt * 32 + r;
            if (tqr < T) {
                const float rm = rowmask ? rowmask[m0 + tqr] : 1.f;
                const float sc = (rm == 0.0f) ? 0.f : scale * c_out;            // blanked query rows pass no gradient to Q
                bf16x4 lo, hi;
#pragma unroll
                for (int i = 0; i < 4; ++i) { lo[i] = (bf16)(n0[i] * sc); hi[i] = (bf16)(n1[i] * sc); }
                bf16* const row = dqkv + (size_t)(m0 + tqr) * lddkv + head * DKP + 4 * hh;
                *reinterpret_cast<bf16x4*>(row) = lo;
                *reinterpret_cast<bf16x4*>(row + 8) = hi;
            }
        }
        qt = qn;
        //@ y
    };
    //@ sweep
    if (live) {
        for (int t = 0; t < ntp - 2; ++t) body(std::true_type{}, std::false_type{}, t);
        body(std::true_type{}, std::true_type{}, ntp - 2);
        body(std::false_type{}, std::true_type{}, ntp - 1);
    }
    __builtin_amdgcn_s_setprio(0);
    //@ swept
    // dK = c ln2 * acc rows 16.. (scores are in the log2 domain), dV = c * acc rows 0..15; column key = r.  Stored as soon as the wave's
    // own sweep ends: nothing here depends on the other waves
    if (live) {
        const float LN2 = 0.6931471805599453f;
        const float ck = c_out * LN2;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const f32x16& acc = which ? accB : accA;
            const int tk = (which ? ktB : ktA) * 32 + r;
            if (tk < T) {
                const size_t m = (size_t)m0 + tk;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    bf16x4 kv, vv;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { kv[j] = (bf16)(acc[8 + 4 * g + j] * ck); vv[j] = (bf16)(acc[4 * g + j] * c_out); }
                    const int e0 = head * DKP + 8 * g + 4 * hh;
                    *reinterpret_cast<bf16x4*>(dqkv + m * lddkv + HD + e0) = kv;
                    *reinterpret_cast<bf16x4*>(dqkv + m * lddkv + 2 * HD + e0) = vv;
                }
            }
        }
    }
    //@ dkv_stored
    //@ exit
}
