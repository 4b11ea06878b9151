// Vector-issue probe for gfx950, round 5 form.  Question: how many cycles does one wave64 vector instruction of each kind cost a SIMD
// with 1 / 2 / 4 / 8 waves resident ON THAT SIMD — measured so that the answer can be checked against itself.
//
// Round 4's table (profiles/r04_valu_micro.txt) was wrong in two ways: its 8-waves column divided a wave's own
// time by a LABELLED occupancy that the waves did not have (the launch took 2.7x as long as its median wave lived), and its loop body
// was 16 instructions long, so a lone wave measured the taken branch and the instruction fetch behind it, not the instruction (12.2
// cycles per v_add_f32 against the 4 of MI355X_MICROARCH.md's constants table).  This form
//   * records for EVERY wave where it ran (HW_REG_HW_ID: SE / SH / CU / SIMD, HW_REG_XCC_ID) and when (s_memrealtime at entry and exit,
//     100 MHz, plus s_memtime for the clock), and derives the occupancy from that record: per SIMD the number of waves whose lifetimes
//     overlap at the SIMD's busiest moment ("res"), and the SIMD's throughput as ALL instructions its waves retired over the span from its
//     first wave's entry to its last wave's exit ("/SIMD");
//   * prints the wall-clock figure beside it (launch time by HIP events / instructions per SIMD): both must agree within 10 %;
//   * places one workgroup of 4 w waves per CU for w <= 4 (one workgroup cannot be split over CUs, and its waves go round the four
//     SIMDs), and TWO workgroups of 16 waves for w = 8, which the dispatcher may or may not co-schedule: the record tells;
//   * runs 128 instructions per loop iteration (8 x 16 independent chains) and measures the empty loop, which is subtracted;
//   * converts every row to FLOP/s where that means something and refuses to print a row above the 157.3 TFLOP/s vector peak
//     without flagging it.
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_micro tools/valu_micro.hip && tools/bin/valu_micro > profiles/r05_valu_micro.txt
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { EMPTY, ADD, FMA, PK_FMA, PK_MUL, EXP, CVT_PK_BF16, BFE_I32, AND, MOV, CNDMASK_SGPR, BFI, MAX3, PERM32SWAP, PERM16SWAP, DPP_MOV,
          MFMA32, MFMA16, MFMA32_FILL6, MFMA16_FILL2, MFMA32_FILL12, EXP_FMA_MIX, BWD_MIX, NOPS };
struct OpInfo { const char* name; int per_iter; double flop_per_lane; };       // flop_per_lane: FLOP per lane and instruction (0: none)
static const OpInfo ops[NOPS] = {
    {"(empty loop)", 1, 0}, {"v_add_f32", 128, 1}, {"v_fma_f32", 128, 2}, {"v_pk_fma_f32", 128, 4}, {"v_pk_mul_f32", 128, 2},
    {"v_exp_f32", 128, 0}, {"v_cvt_pk_bf16_f32", 128, 0}, {"v_bfe_i32", 128, 0}, {"v_and_b32", 128, 0}, {"v_mov_b32", 128, 0},
    {"v_cndmask_b32 (sgpr pair)", 128, 0}, {"v_bfi_b32", 128, 0}, {"v_max3_f32", 128, 0}, {"v_permlane32_swap_b32", 128, 0},
    {"v_permlane16_swap_b32", 128, 0}, {"v_mov_b32 dpp quad_perm", 128, 0},
    {"mfma 32x32x16 bf16 (one chain)", 8, 0}, {"mfma 16x16x32 bf16 (one chain)", 8, 0},
    {"mfma32 + 6 v_fma per gap (per MFMA)", 8, 0}, {"mfma16 + 2 v_fma per gap (per MFMA)", 8, 0}, {"mfma32 + 12 v_fma per gap (per MFMA)", 8, 0},
    {"8 v_exp + 8 v_fma interleaved", 128, 0},
    {"bwd score mix: exp + 2 cndmask + mul + 0.5 cvt_pk (x16 = 72)", 72 * 2, 0}};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct Rec { unsigned long long rt0, rt1, cyc; unsigned hwid, xcc; };

template <int OP>
__global__ __launch_bounds__(1024) void probe(Rec* out, int iters) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = 0.001f * (float)(threadIdx.x + 7 * i + 1);
    float k = 1.0001f, z = 0.5f;
    unsigned long long mask = 0xF0F0F0F0A5A5A5A5ull ^ (unsigned long long)blockIdx.x;
    mask = __builtin_amdgcn_readfirstlane((unsigned)mask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mask >> 32)) << 32);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4_ a4 = {0.f, 0.f, 0.f, 0.f};
    bf16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (float)(threadIdx.x & 15)); fb[i] = (__bf16)0.5f; }
    asm volatile("" : "+v"(k), "+v"(z));
    __syncthreads();                                    // the workgroup's waves start the timed part together
    unsigned long long t0, t1, q0, q1;
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hwid), "=s"(xcc));
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %1\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(q0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#define R16(STMT) _Pragma("unroll") for (int i = 0; i < 16; ++i) { STMT; }
#define R128(STMT) _Pragma("unroll") for (int u = 0; u < 8; ++u) { R16(STMT) }
        if (OP == EMPTY) asm volatile("s_nop 0");
        else if (OP == ADD) R128(asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k)))
        else if (OP == FMA) R128(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z)))
        else if (OP == PK_FMA) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 16; i += 2)
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 2) & 15])));
        } else if (OP == PK_MUL) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 16; i += 2)
                    asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 2) & 15])));
        } else if (OP == EXP) R128(asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])))
        else if (OP == CVT_PK_BF16) R128(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k)))
        else if (OP == BFE_I32) R128(asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(r[i])))
        else if (OP == AND) R128(asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(k)))
        else if (OP == MOV) R128(asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(k)))
        else if (OP == CNDMASK_SGPR) R128(asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(r[i]) : "s"(mask)))
        else if (OP == BFI) R128(asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "v"(k), "v"(z)))
        else if (OP == MAX3) R128(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z)))
        else if (OP == PERM32SWAP) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 16; i += 2) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
        } else if (OP == PERM16SWAP) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 16; i += 2) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
        } else if (OP == DPP_MOV) R128(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i])))
        else if (OP == MFMA32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        } else if (OP == MFMA16) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4, 0, 0, 0);
        } else if (OP == MFMA32_FILL6 || OP == MFMA32_FILL12) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < (OP == MFMA32_FILL6 ? 6 : 12); ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z));
            }
        } else if (OP == MFMA16_FILL2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4, 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z));
            }
        } else if (OP == EXP_FMA_MIX) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[8 + i]) : "v"(k), "v"(z));
                }
        } else if (OP == BWD_MIX) {
            // what one 32x32 tile of the one-pass attention backward asks of the vector pipe per accumulator register pair, in its
            // SGPR-mask form: P = exp2(s); Pm = mask ? P : 0; x = mask ? dp : nd; dS = P x; then two packs per register pair
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(r[i]), "+v"(r[i + 1]));
                    asm volatile("v_cndmask_b32 %0, 0, %1, %2" : "=v"(z) : "v"(r[i]), "s"(mask));
                    asm volatile("v_cndmask_b32 %0, 0, %1, %2" : "=v"(k) : "v"(r[i + 1]), "s"(mask));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(z) : "v"(k));
                    asm volatile("v_cndmask_b32 %0, %1, %0, %2\n\tv_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(z), "s"(mask));
                    asm volatile("v_cndmask_b32 %0, %1, %0, %2\n\tv_mul_f32 %0, %0, %1" : "+v"(r[i + 1]) : "v"(z), "s"(mask));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(k) : "v"(r[i]), "v"(r[i + 1]));
                }
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(q1) :: "memory");
    float s = a4[0] + a4[1] + a4[2] + a4[3] + k + z;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + acc[i];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { Rec rc; rc.rt0 = q0; rc.rt1 = q1; rc.cyc = t1 - t0; rc.hwid = hwid; rc.xcc = xcc; out[wave] = rc; }
    if (s == 123.456f) out[0].cyc = 0;
}

struct Row { double cyc_wave, resident, cyc_simd_resident, cyc_simd_wall, ghz; };

template <int OP>
static Row measure(Rec* dbuf, int cus, int w, int iters, double empty_cyc_per_iter) {
    const int wg_waves = std::min(16, 4 * w), blocks = cus * (4 * w / wg_waves), waves = blocks * wg_waves;
    const double n_instr = (double)iters * ops[OP].per_iter;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(wg_waves * 64), 0, 0, dbuf, iters);
    CHECK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(wg_waves * 64), 0, 0, dbuf, iters);
    CHECK(hipEventRecord(b, 0));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<Rec> h(waves);
    CHECK(hipMemcpy(h.data(), dbuf, waves * sizeof(Rec), hipMemcpyDeviceToHost));
    // per SIMD: the waves that ran there; the window in which ALL of them were alive (if the dispatcher ran them one after the other the
    // window is empty and the SIMD counts with the number that did overlap pairwise at the median wave's midpoint instead)
    std::map<unsigned, std::vector<int>> simd;
    std::vector<double> cyc(waves), ghz(waves);
    for (int i = 0; i < waves; ++i) {
        const unsigned id = h[i].hwid;
        const unsigned key = ((h[i].xcc & 0xF) << 16) | (((id >> 13) & 7) << 12) | (((id >> 12) & 1) << 11) | (((id >> 8) & 0xF) << 4) | ((id >> 4) & 3);
        simd[key].push_back(i);
        cyc[i] = (double)h[i].cyc;
        ghz[i] = (double)h[i].cyc / (double)(h[i].rt1 - h[i].rt0) * 0.1;
    }
    std::vector<double> resid, rate, gh(ghz);
    for (auto& kv : simd) {
        // sweep: at the midpoint of the SIMD's busiest interval, who is alive?
        std::vector<std::pair<unsigned long long, int>> ev;
        for (int i : kv.second) { ev.push_back({h[i].rt0, +1}); ev.push_back({h[i].rt1, -1}); }
        std::sort(ev.begin(), ev.end());
        int cur = 0, best = 0; unsigned long long bs = 0, be = 0;
        for (size_t e = 0; e + 1 < ev.size(); ++e) {
            cur += ev[e].second;
            if (cur > best || (cur == best && ev[e + 1].first - ev[e].first > be - bs)) { best = cur; bs = ev[e].first; be = ev[e + 1].first; }
        }
        if (be <= bs) continue;
        // The SIMD's throughput: every instruction its waves retired, over the span from its first wave's entry to its last wave's exit.
        // (NOT a per-wave time divided by the occupancy, and not an "overlap window": arbitration is oldest-first, so the waves of a SIMD
        // do not share it evenly — the oldest runs at its lone-wave rate and finishes first; see the wave-life figures of VM_DEBUG.)
        unsigned long long s0 = ~0ull, s1 = 0; double cyc_per_tick = 0.0; int alive = 0;
        for (int i : kv.second) {
            s0 = std::min(s0, h[i].rt0); s1 = std::max(s1, h[i].rt1);
            cyc_per_tick += cyc[i] / (double)(h[i].rt1 - h[i].rt0);
            if (h[i].rt0 <= bs && h[i].rt1 >= be) ++alive;
        }
        const double instr = n_instr * (double)kv.second.size();
        const double win_cyc = (double)(s1 - s0) * cyc_per_tick / (double)kv.second.size();
        if (!alive) continue;
        resid.push_back((double)alive);
        rate.push_back(win_cyc / instr);
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
    if (getenv("VM_DEBUG")) {
        unsigned long long lo = ~0ull, hi = 0;
        for (int i = 0; i < waves; ++i) { lo = std::min(lo, h[i].rt0); hi = std::max(hi, h[i].rt1); }
        std::map<int, int> hist;
        for (auto& kv : simd) hist[(int)kv.second.size()]++;
        std::vector<double> c2(cyc); std::sort(c2.begin(), c2.end());
        std::vector<double> life; for (int i = 0; i < waves; ++i) life.push_back((double)(h[i].rt1 - h[i].rt0) * 0.01);
        std::sort(life.begin(), life.end());
        printf("\n   [w=%d %s] event %.1f us, device span %.1f us, wave life min/med/max %.1f/%.1f/%.1f us, cycles min/med/max %.0f/%.0f/%.0f, SIMDs seen %zu, waves-per-SIMD histogram:",
               w, ops[OP].name, ms * 1e3, (double)(hi - lo) * 0.01, life.front(), life[waves / 2], life.back(), c2.front(), c2[waves / 2], c2.back(), simd.size());
        for (auto& kv : hist) printf(" %dx%d", kv.first, kv.second);
        printf("\n");
    }
    Row r;
    const double g = med(gh);
    r.ghz = g;
    r.cyc_wave = (med(cyc) - empty_cyc_per_iter * iters) / n_instr;
    r.resident = med(resid);
    r.cyc_simd_resident = med(rate);
    // wall clock: the launch's time, minus nothing, over the instructions one SIMD was handed (waves / (cus * 4) per SIMD)
    r.cyc_simd_wall = (double)ms * 1e-3 * g * 1e9 / (n_instr * ((double)waves / (cus * 4.0)));
    return r;
}

template <int OP>
static void run(Rec* dbuf, int cus, const double* empty) {
    const int iters = (OP >= MFMA32 && OP <= MFMA32_FILL12) ? 4000 : 600;
    printf("%-58s", ops[OP].name);
    for (int wi = 0; wi < 4; ++wi) {
        const int w = 1 << wi;
        const Row r = measure<OP>(dbuf, cus, w, iters, empty ? empty[wi] : 0.0);
        const double agree = r.cyc_simd_wall > 0 ? r.cyc_simd_resident / r.cyc_simd_wall : 0;
        printf(" | w=%d res %.1f: %6.2f /wave %5.2f /SIMD (wall %5.2f, x%.2f) %.2f GHz", w, r.resident, r.cyc_wave, r.cyc_simd_resident, r.cyc_simd_wall, agree, r.ghz);
        if (ops[OP].flop_per_lane > 0 && r.cyc_simd_resident > 0) {
            const double tf = 64.0 * ops[OP].flop_per_lane / r.cyc_simd_resident * 4 * cus * r.ghz * 1e9 / 1e12;
            printf(" %5.1f TF%s", tf, tf > 157.3 * 1.02 ? " !!ABOVE-PEAK" : "");
        }
    }
    printf("\n");
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("# tools/valu_micro.hip (round 5): cycles per wave64 instruction.  /wave: as the median wave sees it (empty loop subtracted).  res: waves\n"
           "# observed resident together on a SIMD at its busiest moment (HW_ID + s_memrealtime record).  /SIMD: device-side span of the SIMD (first\n"
           "# entry to last exit of its waves) per instruction its waves retired.  wall: launch time by HIP events x clock / instructions handed to one SIMD; x = /SIMD : wall (1.00 = they agree).\n"
           "device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    Rec* dbuf;
    CHECK(hipMalloc(&dbuf, (size_t)cus * 32 * sizeof(Rec)));
    {   // bring the clocks to their loaded steady state: ~2 s of back-to-back launches
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float ms = 0.f, tot = 0.f;
        while (tot < 2000.f) {
            CHECK(hipEventRecord(a, 0));
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<FMA>, dim3(cus), dim3(1024), 0, 0, dbuf, 500);
            CHECK(hipEventRecord(b, 0)); CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms, a, b)); tot += ms;
        }
    }
    double empty[4];
    printf("%-58s", ops[EMPTY].name);
    for (int wi = 0; wi < 4; ++wi) {
        const Row r = measure<EMPTY>(dbuf, cus, 1 << wi, 20000, 0.0);
        empty[wi] = r.cyc_wave;                         // cycles per (empty) iteration as a wave sees it
        printf(" | w=%d res %.1f: %6.2f cycles per iteration and wave", 1 << wi, r.resident, r.cyc_wave);
    }
    printf("\n");
    run<ADD>(dbuf, cus, empty); run<FMA>(dbuf, cus, empty); run<PK_FMA>(dbuf, cus, empty); run<PK_MUL>(dbuf, cus, empty);
    run<EXP>(dbuf, cus, empty); run<CVT_PK_BF16>(dbuf, cus, empty); run<BFE_I32>(dbuf, cus, empty); run<AND>(dbuf, cus, empty);
    run<MOV>(dbuf, cus, empty); run<CNDMASK_SGPR>(dbuf, cus, empty); run<BFI>(dbuf, cus, empty); run<MAX3>(dbuf, cus, empty);
    run<PERM32SWAP>(dbuf, cus, empty); run<PERM16SWAP>(dbuf, cus, empty); run<DPP_MOV>(dbuf, cus, empty);
    run<MFMA32>(dbuf, cus, empty); run<MFMA16>(dbuf, cus, empty); run<MFMA32_FILL6>(dbuf, cus, empty); run<MFMA16_FILL2>(dbuf, cus, empty);
    run<MFMA32_FILL12>(dbuf, cus, empty); run<EXP_FMA_MIX>(dbuf, cus, empty); run<BWD_MIX>(dbuf, cus, empty);
    return 0;
}
