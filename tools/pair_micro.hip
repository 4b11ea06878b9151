// Cross-wave interference probe for gfx950: one 512-thread workgroup per CU = two waves per SIMD.  Waves 0..3 ("V") run a pure vector
// stream, waves 4..7 ("M") a pure matrix (or LDS, or nothing) stream — the complementary phases of attn_bwd_pair.h in isolation.
// Printed per combination: cycles per V instruction and per M instruction as each wave sees it, against the same stream run with an
// idle partner.  Question: does a wave's MFMA take vector issue cycles from the OTHER wave of its SIMD beyond the 8 cycles it holds the
// issue port, and what do the partner's LDS instructions cost?
//   hipcc --offload-arch=gfx950 -O2 -o tools/bin/pair_micro tools/pair_micro.hip && tools/bin/pair_micro
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4_ __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
enum VOp { V_NONE, V_FMA, V_EXP, V_BFE, V_CVT, V_PKFMA, V_MIX, NV };
enum MOp { M_NONE, M_MFMA32_CHAIN, M_MFMA32_4ACC, M_MFMA16_4ACC, M_LDS_TR, M_LDS_W128, M_MFMA32_LDS, NM };
static const char* vname[NV] = {"(idle)", "v_fma_f32", "v_exp_f32", "v_bfe_i32", "v_cvt_pk_bf16_f32", "v_pk_fma_f32", "bwd X mix (exp, bfe, and, mul, fma, cvt)"};
static const int vper[NV] = {1, 128, 128, 128, 128, 128, 16 * 6};
static const char* mname[NM] = {"(idle)", "mfma32x32x16, one chain", "mfma32x32x16, 4 accumulators", "mfma16x16x32, 4 accumulators", "ds_read_b64_tr_b16 x16", "ds_write_b128 x8", "4 mfma32 + 8 ds_read_tr + 2 ds_write_b128"};
static const int mper[NM] = {1, 8, 8, 8, 16, 8, 4};

template <int VO, int MO>
__global__ __launch_bounds__(512) void probe(unsigned long long* out, int iters, int prio) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = 0.001f * (float)(threadIdx.x + 7 * i + 1);
    float k = 1.0001f, z = 0.5f;
    unsigned w = 0x5A5A1234u ^ threadIdx.x;
    f32x16 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    f32x4_ a4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    bf16x8 fa, fb;
#pragma unroll
    for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(0.01f * (float)(threadIdx.x & 15)); fb[i] = (__bf16)0.5f; }
    for (int i = threadIdx.x; i < 8192; i += 512) reinterpret_cast<float*>(lds)[i] = 0.f;
    asm volatile("" : "+v"(k), "+v"(z), "+v"(w));
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool mrole = wave >= 4;
    char* my = lds + wave * 4096 + lane * 16;
    __syncthreads();
    if (mrole && prio) __builtin_amdgcn_s_setprio(3);
    unsigned long long t0, t1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (!mrole) {
        for (int it = 0; it < iters; ++it) {
#define R16(STMT) _Pragma("unroll") for (int i = 0; i < 16; ++i) { STMT; }
#define R128(STMT) _Pragma("unroll") for (int u = 0; u < 8; ++u) { R16(STMT) }
            if (VO == V_FMA) R128(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(k), "v"(z)))
            else if (VO == V_EXP) R128(asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])))
            else if (VO == V_BFE) R128(asm volatile("v_bfe_i32 %0, %0, 3, 1" : "+v"(r[i])))
            else if (VO == V_CVT) R128(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(k)))
            else if (VO == V_PKFMA) {
#pragma unroll
                for (int u = 0; u < 16; ++u)
#pragma unroll
                    for (int i = 0; i < 16; i += 2)
                        asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*reinterpret_cast<double*>(&r[i])) : "v"(*reinterpret_cast<double*>(&r[(i + 2) & 15])));
            } else if (VO == V_MIX) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float m, t;
                    asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
                    asm volatile("v_bfe_i32 %0, %1, 3, 1" : "=v"(m) : "v"(w));
                    asm volatile("s_nop 0\n\tv_mul_f32 %0, %1, %2" : "=v"(t) : "v"(r[i]), "v"(z));
                    asm volatile("v_and_b32 %0, %0, %1" : "+v"(m) : "v"(r[i]));
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r[i]) : "v"(m), "v"(k), "v"(t));
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(t) : "v"(m), "v"(r[i]));
                }
            }
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            if (MO == M_MFMA32_CHAIN) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[0], 0, 0, 0);
            } else if (MO == M_MFMA32_4ACC) {
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[u & 3], 0, 0, 0);
            } else if (MO == M_MFMA16_4ACC) {
#pragma unroll
                for (int u = 0; u < 8; ++u) a4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, a4[u & 3], 0, 0, 0);
            } else if (MO == M_LDS_TR) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(my + (u & 1) * 8));
                    asm volatile("" :: "v"(v));
                }
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (MO == M_LDS_W128) {
#pragma unroll
                for (int u = 0; u < 8; ++u) *reinterpret_cast<volatile bf16x8*>(my) = fa;
                asm volatile("s_waitcnt lgkmcnt(0)");
            } else if (MO == M_MFMA32_LDS) {
#pragma unroll
                for (int u = 0; u < 2; ++u) *reinterpret_cast<volatile bf16x8*>(my) = fa;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(my + (u & 1) * 8));
                    asm volatile("" :: "v"(v));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[u & 3], 0, 0, 0);
                asm volatile("s_waitcnt lgkmcnt(0)");
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = k + z + (float)w;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += r[i] + acc[0][i] + acc[1][i] + acc[2][i] + acc[3][i];
#pragma unroll
    for (int a = 0; a < 4; ++a) s += a4[a][0] + a4[a][1] + a4[a][2] + a4[a][3];
    const int gw = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (lane == 0) out[gw] = t1 - t0;
    if (s == 123.456f) out[0] = 0;
}

template <int VO, int MO>
static void run(unsigned long long* dbuf, int cus, int iv, int im, int prio) {
    // iters: the V role runs iv iterations, the M role im; both loops live in one kernel, so the shorter role simply ends first — the
    // figures are taken from runs where the measured role is the SHORTER one (its partner is active throughout)
    auto once = [&](int iters_v, int iters_m, double& cv, double& cm) {
        // the kernel takes one iteration count: run twice, once sized for each role
        (void)iters_m;
        hipLaunchKernelGGL((probe<VO, MO>), dim3(cus), dim3(512), 0, 0, dbuf, iters_v, prio);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(cus * 8);
        CHECK(hipMemcpy(h.data(), dbuf, h.size() * 8, hipMemcpyDeviceToHost));
        std::vector<double> v, m;
        for (int b = 0; b < cus; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? v : m).push_back((double)h[b * 8 + w]);
        std::sort(v.begin(), v.end()); std::sort(m.begin(), m.end());
        cv = v[v.size() / 2] / ((double)iters_v * vper[VO]);
        cm = m[m.size() / 2] / ((double)iters_v * mper[MO]);
    };
    double cv, cm;
    once(iv, im, cv, cm);
    once(iv, im, cv, cm);
    printf("V: %-44s M: %-44s prio %d | %7.2f cycles per V instruction, %7.2f per M instruction (each as its own wave sees it; M iterations = V iterations)\n",
           vname[VO], mname[MO], prio, cv, cm);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned long long* dbuf;
    CHECK(hipMalloc(&dbuf, (size_t)cus * 8 * 8));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL((probe<V_FMA, M_MFMA32_4ACC>), dim3(cus), dim3(512), 0, 0, dbuf, 2000, 0);
    CHECK(hipDeviceSynchronize());
    const int it = 2000;
    printf("# per-iteration work: V 128 instructions (mix: 96), M 8 MFMAs / 16 reads / 8 writes / (4 MFMA + 8 reads + 2 writes).  With equal iteration counts the roles end at\n"
           "# different times; the role that ends LAST saw an idle partner for part of its life, so read each line's SHORTER-lived role.\n");
    run<V_FMA, M_NONE>(dbuf, cus, it, it, 0); run<V_EXP, M_NONE>(dbuf, cus, it, it, 0); run<V_BFE, M_NONE>(dbuf, cus, it, it, 0);
    run<V_CVT, M_NONE>(dbuf, cus, it, it, 0); run<V_PKFMA, M_NONE>(dbuf, cus, it, it, 0); run<V_MIX, M_NONE>(dbuf, cus, it, it, 0);
    run<V_NONE, M_MFMA32_CHAIN>(dbuf, cus, it, it, 0); run<V_NONE, M_MFMA32_4ACC>(dbuf, cus, it, it, 0); run<V_NONE, M_MFMA16_4ACC>(dbuf, cus, it, it, 0);
    run<V_NONE, M_LDS_TR>(dbuf, cus, it, it, 0); run<V_NONE, M_LDS_W128>(dbuf, cus, it, it, 0); run<V_NONE, M_MFMA32_LDS>(dbuf, cus, it, it, 0);
    for (int prio = 0; prio < 2; ++prio) {
        run<V_FMA, M_MFMA32_CHAIN>(dbuf, cus, it, it, prio); run<V_FMA, M_MFMA32_4ACC>(dbuf, cus, it, it, prio); run<V_FMA, M_MFMA16_4ACC>(dbuf, cus, it, it, prio);
        run<V_EXP, M_MFMA32_4ACC>(dbuf, cus, it, it, prio); run<V_BFE, M_MFMA32_4ACC>(dbuf, cus, it, it, prio); run<V_CVT, M_MFMA32_4ACC>(dbuf, cus, it, it, prio);
        run<V_PKFMA, M_MFMA32_4ACC>(dbuf, cus, it, it, prio); run<V_MIX, M_MFMA32_4ACC>(dbuf, cus, it, it, prio);
        run<V_MIX, M_LDS_TR>(dbuf, cus, it, it, prio); run<V_MIX, M_LDS_W128>(dbuf, cus, it, it, prio); run<V_MIX, M_MFMA32_LDS>(dbuf, cus, it, it, prio);
        run<V_FMA, M_MFMA32_LDS>(dbuf, cus, it, it, prio);
    }
    return 0;
}
