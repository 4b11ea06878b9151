// VALU issue-rate probe for gfx950: how many cycles does one wave64 vector instruction of each kind cost a SIMD, alone and
// with 2 / 4 waves sharing the SIMD?  The attention kernels are VALU-issue bound; their slot accounting
// (4 cycles per plain VALU instruction, 16 per transcendental) is checked here, and so are the candidates for a cheaper
// inner loop (packed fp32, SGPR-masked selects, exp2 by polynomial).
//
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/valu_micro tools/valu_micro.hip && tools/bin/valu_micro
//
// Each kernel runs ITERS x 16 independent instructions of one kind between two s_memtime stamps; printed: cycles per
// instruction as one wave sees it (median over waves) and the same divided into the waves on the SIMD (= SIMD throughput).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { ADD, FMA, MUL, PK_FMA, PK_ADD, PK_MUL, EXP, EXP_F16, LDEXP, CNDMASK_VCC, CNDMASK_SGPR, CVT_PK_BF16, MAX3, MAD_U24, XAD, LSHR,
          CMP_U16, DPP_MOV, AND, EXP_FMA_MIX, EXP_PKFMA_MIX, PERM, FRACT, CVT_I32, MFMA_EXP_MIX, CMP16_CND, CMPSDWA_CND, MOV_B64, DROP_FIN, SUB, MFMA_ONLY,
          PK_LSHL16, PK_ASHR16, PERMSWAP, DOT2C_BF16, BFE_I32, MOV_B32, PK_MUL_F16, MFMA16_ONLY, PK_MASK3, BFE_AND2, MFMA_C_OTHER, NOPS };
static const char* const op_names[NOPS] = {
    "v_add_f32", "v_fma_f32", "v_mul_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_exp_f32", "v_exp_f16", "v_ldexp_f32",
    "v_cndmask_b32 (vcc)", "v_cndmask_b32 (sgpr pair)", "v_cvt_pk_bf16_f32", "v_max3_f32", "v_mad_u32_u24", "v_xad_u32", "v_lshrrev_b32",
    "v_cmp_le_u16 (->vcc)", "v_mov_b32 dpp quad_perm", "v_and_b32", "8 v_exp + 8 v_fma interleaved", "8 v_exp + 8 v_pk_fma interleaved",
    "v_perm_b32", "v_fract_f32", "v_cvt_i32_f32", "1 mfma32x32x16 + 4 v_exp + 8 v_fma",
    "v_cmp_le_u16 + v_cndmask(vcc) pair", "v_cmp_ge_u32_sdwa + v_cndmask pair", "v_mov_b64", "drop_fin (6 dependent int ops)", "v_sub_f32", "mfma32x32x16 only",
    "v_pk_lshlrev_b16", "v_pk_ashrrev_i16", "v_permlane32_swap_b32", "v_dot2c_f32_bf16", "v_bfe_i32", "v_mov_b32", "v_pk_mul_f16", "mfma16x16x32 only",
    "pair mask: pk_lshl16 + pk_ashr16 + and (x8 = 24 instr)", "score mask: bfe_i32 + and (x8 = 16 instr)", "mfma32x32x16, C = other registers"};

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int OP>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, int iters) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = 0.001f * (float)(threadIdx.x + 7 * i + 1);
    float k = 1.0001f, z = 0.5f;
    unsigned long long mask = 0xF0F0F0F0A5A5A5A5ull ^ (unsigned long long)blockIdx.x;
    mask = __builtin_amdgcn_readfirstlane((unsigned)mask) | ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(mask >> 32)) << 32);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    bf16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (float)(threadIdx.x & 15)); fb[i] = (__bf16)0.5f; }
    asm volatile("" : "+v"(k), "+v"(z));
    unsigned long long t0, t1, q0, q1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %1\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(q0) :: "memory");
    for (int it = 0; it < iters; ++it) {
#define R16(STMT) _Pragma("unroll") for (int i = 0; i < 16; ++i) { STMT; }
        if (OP == ADD) R16(asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k)))
        else if (OP == FMA) R16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z)))
        else if (OP == MUL) R16(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k)))
        else if (OP == PK_FMA) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {       // 8 registers pairs, twice
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 2) & 15])));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 4) & 15])));
            }
        } else if (OP == PK_ADD) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 2) & 15])));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 4) & 15])));
            }
        } else if (OP == PK_MUL) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 2) & 15])));
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 4) & 15])));
            }
        } else if (OP == EXP) R16(asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])))
        else if (OP == EXP_F16) R16(asm volatile("v_exp_f16 %0, %0" : "+v"(r[i])))
        else if (OP == LDEXP) R16(asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(r[i]) : "v"(1)))
        else if (OP == CNDMASK_VCC) R16(asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(r[i]) :: "vcc"))
        else if (OP == CNDMASK_SGPR) R16(asm volatile("v_cndmask_b32 %0, 0, %0, %1" : "+v"(r[i]) : "s"(mask)))
        else if (OP == CVT_PK_BF16) R16(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k)))
        else if (OP == MAX3) R16(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z)))
        else if (OP == MAD_U24) R16(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z)))
        else if (OP == XAD) R16(asm volatile("v_xad_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z)))
        else if (OP == LSHR) R16(asm volatile("v_lshrrev_b32 %0, 3, %0" : "+v"(r[i])))
        else if (OP == CMP_U16) R16(asm volatile("v_cmp_le_u16 vcc, %0, %1" :: "v"(r[i]), "v"(k) : "vcc"))
        else if (OP == DPP_MOV) R16(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i])))
        else if (OP == AND) R16(asm volatile("v_and_b32 %0, %0, %1" : "+v"(r[i]) : "v"(k)))
        else if (OP == EXP_FMA_MIX) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[8 + i]) : "v"(k), "v"(z));
            }
        } else if (OP == EXP_PKFMA_MIX) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&r[8 + (i & 6)])) : "v"(*reinterpret_cast<double*>(&r[8 + ((i + 2) & 6)])));
            }
        } else if (OP == PERM) R16(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z)))
        else if (OP == FRACT) R16(asm volatile("v_fract_f32 %0, %0" : "+v"(r[i])))
        else if (OP == CVT_I32) R16(asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(r[i])))
        else if (OP == CMP16_CND) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cmp_le_u16 vcc, %1, %2\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(r[i]) : "v"(k), "v"(r[8 + i]) : "vcc");
        } else if (OP == CMPSDWA_CND) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cmp_ge_u32_sdwa vcc, %1, %2 src0_sel:WORD_1 src1_sel:DWORD\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(r[i]) : "v"(r[8 + i]), "v"(k) : "vcc");
        } else if (OP == MOV_B64) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                asm volatile("v_mov_b64 %0, %1" : "=v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 2) & 15])));
                asm volatile("v_mov_b64 %0, %1" : "=v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 4) & 15])));
            }
        } else if (OP == DROP_FIN) {
            // 16 instruction groups of 6: counted as 16 "instructions" of 6 ops each (divide the printed figure by 6)
            R16(asm volatile("v_lshrrev_b32 %1, 15, %0\n\tv_xad_u32 %0, %0, %1, %2\n\tv_lshrrev_b32 %1, 24, %0\n\tv_mad_u32_u24 %0, %0, %3, %1\n\t"
                             "v_lshrrev_b32 %1, 14, %0\n\tv_xor_b32 %0, %0, %1" : "+v"(r[i]), "+v"(z) : "v"(k), "v"(0x9E3779)))
        } else if (OP == SUB) R16(asm volatile("v_sub_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k)))
        else if (OP == MFMA_ONLY) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
        else if (OP == PK_LSHL16) R16(asm volatile("v_pk_lshlrev_b16 %0, 1, %0" : "+v"(r[i])))
        else if (OP == PK_ASHR16) R16(asm volatile("v_pk_ashrrev_i16 %0, 15, %0" : "+v"(r[i])))
        else if (OP == PERMSWAP) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
                asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[i + 1]));
            }
        }
        else if (OP == DOT2C_BF16) R16(asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z)))
        else if (OP == BFE_I32) R16(asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(r[i])))
        else if (OP == MOV_B32) R16(asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(k)))
        else if (OP == PK_MUL_F16) R16(asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(r[i]) : "v"(k)))
        else if (OP == MFMA16_ONLY) {
            typedef float f32x4_ __attribute__((ext_vector_type(4)));
            f32x4_ a4 = {acc[0], acc[1], acc[2], acc[3]};
            a4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4, 0, 0, 0);
            acc[0] = a4[0]; acc[1] = a4[1]; acc[2] = a4[2]; acc[3] = a4[3];
        }
        else if (OP == PK_MASK3) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_pk_lshlrev_b16 %1, 3, %2\n\tv_pk_ashrrev_i16 %1, 15, %1\n\tv_and_b32 %0, %0, %1" : "+v"(r[i]), "+v"(r[8 + i]) : "v"(k));
        }
        else if (OP == BFE_AND2) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_bfe_i32 %1, %2, 3, 1\n\tv_and_b32 %0, %0, %1" : "+v"(r[i]), "+v"(r[8 + i]) : "v"(k));
        }
        else if (OP == MFMA_C_OTHER) {
            f32x16 cc;
#pragma unroll
            for (int i = 0; i < 16; ++i) cc[i] = k;
            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, cc, 0, 0, 0);
            asm volatile("" : "+v"(d));
            r[0] += d[0];
        }
        else if (OP == MFMA_EXP_MIX) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[8 + i]) : "v"(k), "v"(z));
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(q1) :: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + acc[i];
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if ((threadIdx.x & 63) == 0) { out[2 * wave] = t1 - t0; out[2 * wave + 1] = q1 - q0; }
    if (s == 123.456f) out[0] = 0;
}

template <int OP>
static void run(unsigned long long* dbuf, int cus) {
    const int iters = 4000;
    const int per_iter = (OP == MFMA_EXP_MIX) ? 13 : ((OP == MFMA_ONLY || OP == MFMA16_ONLY || OP == MFMA_C_OTHER) ? 1 : (OP == PK_MASK3 ? 24 : 16));
    printf("%-36s", op_names[OP]);
    for (int w : {1, 2, 4, 8}) {
        const int blocks = cus * w, waves = blocks * 4;
        for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, dbuf, iters);
        CHECK(hipDeviceSynchronize());
        hipEvent_t a, b;
        CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        CHECK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, dbuf, iters);
        CHECK(hipEventRecord(b, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, a, b));
        std::vector<unsigned long long> h2(2 * waves), h(waves), q(waves);
        CHECK(hipMemcpy(h2.data(), dbuf, 2 * waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int i = 0; i < waves; ++i) { h[i] = h2[2 * i]; q[i] = h2[2 * i + 1]; }
        std::sort(h.begin(), h.end()); std::sort(q.begin(), q.end());
        const double med = (double)h[waves / 2] / ((double)iters * per_iter);
        const double ghz = (double)h[waves / 2] / (double)q[waves / 2] * 0.1;      // s_memrealtime ticks at 100 MHz
        // s_memtime counts at a fixed 100 MHz on this part?  print the wall-derived figure too: ns per instruction per SIMD
        const double ns_simd = (double)ms * 1e6 / ((double)iters * per_iter * w);
        printf("  w=%d: %6.2f cyc/instr/wave %6.2f /SIMD %5.2f GHz (%5.3f ns wall) |", w, med, med / w, ghz, ns_simd);
    }
    printf("\n");
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    unsigned long long* dbuf;
    CHECK(hipMalloc(&dbuf, (size_t)cus * 8 * 4 * sizeof(unsigned long long) * 4));
    {   // bring the clocks to their loaded steady state: ~2 s of back-to-back launches
        hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
        float ms = 0.f, tot = 0.f;
        while (tot < 2000.f) {
            CHECK(hipEventRecord(a, 0));
            for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(probe<FMA>, dim3(cus * 4), dim3(256), 0, 0, dbuf, 4000);
            CHECK(hipEventRecord(b, 0)); CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms, a, b)); tot += ms;
        }
    }
    run<ADD>(dbuf, cus); run<FMA>(dbuf, cus); run<MUL>(dbuf, cus); run<PK_FMA>(dbuf, cus); run<PK_ADD>(dbuf, cus); run<PK_MUL>(dbuf, cus);
    run<EXP>(dbuf, cus); run<EXP_F16>(dbuf, cus); run<LDEXP>(dbuf, cus); run<CNDMASK_VCC>(dbuf, cus); run<CNDMASK_SGPR>(dbuf, cus);
    run<CVT_PK_BF16>(dbuf, cus); run<MAX3>(dbuf, cus); run<MAD_U24>(dbuf, cus); run<XAD>(dbuf, cus); run<LSHR>(dbuf, cus);
    run<CMP_U16>(dbuf, cus); run<DPP_MOV>(dbuf, cus); run<AND>(dbuf, cus); run<EXP_FMA_MIX>(dbuf, cus); run<EXP_PKFMA_MIX>(dbuf, cus);
    run<PERM>(dbuf, cus); run<FRACT>(dbuf, cus); run<CVT_I32>(dbuf, cus); run<MFMA_EXP_MIX>(dbuf, cus);
    run<CMP16_CND>(dbuf, cus); run<CMPSDWA_CND>(dbuf, cus); run<MOV_B64>(dbuf, cus); run<DROP_FIN>(dbuf, cus); run<SUB>(dbuf, cus); run<MFMA_ONLY>(dbuf, cus);
    run<PK_LSHL16>(dbuf, cus); run<PK_ASHR16>(dbuf, cus); run<PERMSWAP>(dbuf, cus); run<DOT2C_BF16>(dbuf, cus); run<BFE_I32>(dbuf, cus); run<MOV_B32>(dbuf, cus);
    run<PK_MUL_F16>(dbuf, cus); run<MFMA16_ONLY>(dbuf, cus); run<PK_MASK3>(dbuf, cus); run<BFE_AND2>(dbuf, cus); run<MFMA_C_OTHER>(dbuf, cus);
    return 0;
}
