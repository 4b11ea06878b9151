// LSTM scans for hidden sizes 129..256 (the reference's default embed_dim = 256 decoder,
// transformer/SFT/multiTransformer.py:423,444) at small batches (one sequence per workgroup, B <= 256).
//
// W_rec is 4H x H = 512 KB of bf16 at H = 256: exactly the size of a CU's whole register file, so it cannot be resident
// the way it is for H <= 128 (scan.h).  The generic kernel re-streamed all of it from L2 every step (7-14 us/step).
// Here a wave owns 32 hidden units (two MFMA row tiles); HALF of its weight fragments (the first RES k-blocks of every
// gate/tile) stay in registers for the whole scan, the other half is streamed from L2 in 4-fragment groups through a
// 2-deep register ring that runs ahead across the step boundary (the weights do not depend on h, so the first groups of
// step t+1 are in flight while step t does its gate math and barrier).  L2 traffic per step halves and its latency hides
// behind the dependent part of the step.  Inputs come through the cooperative LDS ring and the gate math is "dense"
// (one lane = one hidden unit) exactly as in scan.h's COOP form.
#pragma once
#include "scan.h"

#define S256_KP 256
#define S256_LDH (S256_KP + 8)

// grid = B (one sequence per workgroup); block = 64 * ceil(HP16/32) <= 512.
template <int RES, int PF>
__global__ __launch_bounds__(512) void lstm_scan_fwd256_kernel(const float* __restrict__ gx, const bf16* __restrict__ Wf,
                                                               const float* __restrict__ h0, const float* __restrict__ c0,
                                                               float* __restrict__ h_all, float* __restrict__ c_all,
                                                               float* __restrict__ acts, int T, int B, int H, int HP16) {
    constexpr int KS = S256_KP / 32, NG = 2 * (KS - RES);                 // streamed groups per step: (k-block, tile)
    __shared__ __attribute__((aligned(16))) bf16 hbuf[2 * 16 * S256_LDH];
    __shared__ __attribute__((aligned(16))) float gslot[2 * 4 * S256_KP];  // [slot][4H] of the one sequence
    __shared__ __attribute__((aligned(16))) float xch[8 * 4 * 32];         // per wave [gate][32 units]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x;
    const int ubase = 32 * w;

    // weight fragment addresses: gate q, tile ut, k-block ks: row q*HP16 + ubase + 16*ut + l15.  One base pointer and uniform
    // offsets (a ragged last tile reads the rows that follow it in the workspace — valid memory, dead results)
    const bf16* wbase = Wf + (size_t)(ubase + l15) * S256_KP + 8 * lq;       // (tile 1 of a ragged last wave reads rows past HP16: finite weights of the next gate / of Wb)
    const size_t gs = (size_t)HP16 * S256_KP;                              // gate stride
    bf16x8 a[4][2][RES];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int ks = 0; ks < RES; ++ks) a[q][ut][ks] = *reinterpret_cast<const bf16x8*>(wbase + q * gs + ut * 16 * S256_KP + ks * 32);
    bf16x8 sw[2][4];                                                       // streamed ring: group g = (ks = RES + g/2, ut = g&1)
    auto load_group = [&](bf16x8 (&dst)[4], int g) {
        const int ks = RES + (g >> 1), ut = g & 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *reinterpret_cast<const bf16x8*>(wbase + q * gs + ut * 16 * S256_KP + ks * 32);
    };
    load_group(sw[0], 0);
    load_group(sw[1], 1);

    for (int i = tid; i < 2 * 16 * S256_LDH; i += blockDim.x) hbuf[i] = (bf16)0.f;
    __syncthreads();
    // dense role: lanes 0..31 of a wave own the wave's 32 units of THE sequence
    const int du = lane & 31, ud = ubase + du;
    const bool lived = lane < 32 && ud < H;
    const int udc = ud < H ? ud : H - 1;
    float cd = 0.f;
    if (lived) {
        if (c0) cd = c0[(size_t)b * H + ud];
        if (h0) hbuf[ud] = (bf16)h0[(size_t)b * H + ud];
    }
    // cooperative input ring: thread i fetches chunk i of the 4H floats of a step
    const bool ld_on = tid < H;
    const float* gxl = gx + (size_t)b * 4 * H + 4 * (ld_on ? tid : 0);
    const size_t gstep = (size_t)B * 4 * H;
    f32x4 ring[PF];
    auto fetch = [&](f32x4& r, int t) { r = *reinterpret_cast<const f32x4*>(gxl + (size_t)(t < T ? t : T - 1) * gstep); };
    {
        f32x4 first;
        fetch(first, 0);
        if (ld_on) *reinterpret_cast<f32x4*>(gslot + 4 * tid) = first;
#pragma unroll
        for (int d = 0; d < PF; ++d) fetch(ring[d], d + 1);
    }
    __syncthreads();

    int cur = 0;
    float* xw = xch + w * 128;
    auto step = [&](int t, f32x4& in) {
        float gin[4];
        const float* sl = gslot + (t & 1) * 4 * S256_KP + udc;
#pragma unroll
        for (int q = 0; q < 4; ++q) gin[q] = sl[q * H];
        if (ld_on) *reinterpret_cast<f32x4*>(gslot + ((t + 1) & 1) * 4 * S256_KP + 4 * tid) = in;
        fetch(in, t + 1 + PF);

        f32x4 acc[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) acc[q][ut] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16* hb = hbuf + cur * 16 * S256_LDH + l15 * S256_LDH + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < RES; ++ks) {
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(hb + ks * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ut = 0; ut < 2; ++ut) acc[q][ut] = mfma16(a[q][ut][ks], bf, acc[q][ut]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int ks = RES + (g >> 1), ut = g & 1;
            const bf16x8 bf = *reinterpret_cast<const bf16x8*>(hb + ks * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q][ut] = mfma16(sw[g & 1][q], bf, acc[q][ut]);
            load_group(sw[g & 1], (g + 2) % NG);                           // two groups ahead, wrapping into the next step
            __builtin_amdgcn_sched_barrier(0);
        }
        // re-deal the live MFMA column (sequence 0 = lanes with l15 == 0) to the dense lanes
        if (l15 == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int ut = 0; ut < 2; ++ut) *reinterpret_cast<f32x4*>(xw + q * 32 + 16 * ut + 4 * lq) = acc[q][ut];
        }
        const float ig = sigmoid_f(xw[du] + gin[0]), fg = sigmoid_f(xw[32 + du] + gin[1]);
        const float gg = tanh_f(xw[64 + du] + gin[2]), og = sigmoid_f(xw[96 + du] + gin[3]);
        cd = fg * cd + ig * gg;
        const float hn = lived ? og * tanh_f(cd) : 0.f;
        if (lane < 32) hbuf[(cur ^ 1) * 16 * S256_LDH + ud] = (bf16)hn;
        lds_barrier();
        if (lived) {
            const size_t o = ((size_t)t * B + b) * H + ud;
            h_all[o] = hn;
            c_all[o] = cd;
            float* ap = acts + ((size_t)t * B + b) * 4 * H + ud;
            ap[0] = ig; ap[H] = fg; ap[2 * H] = gg; ap[3 * H] = og;
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

// Backward.  Wb [HP16][4*256]: row j = hidden unit whose dh is formed, k = gate*256 + unit.  A wave owns 32 rows (two tiles);
// k-blocks [0, RESB) of both tiles stay in registers, [RESB, 32) are streamed in groups of 4 fragments (tile, 4 k-blocks).
template <int RESB, int PF>
__global__ __launch_bounds__(512) void lstm_scan_bwd256_kernel(const float* __restrict__ dh_ext, const float* __restrict__ dc_ext,
                                                               const bf16* __restrict__ Wb, const float* __restrict__ c0,
                                                               const float* __restrict__ c_all, const float* __restrict__ acts,
                                                               float* __restrict__ dG, float* __restrict__ dh0, float* __restrict__ dc0,
                                                               int T, int B, int H, int HP16) {
    constexpr int KP4 = 4 * S256_KP, KS4 = KP4 / 32, ldg = KP4 + 8, NG = 2 * ((KS4 - RESB) / 4);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* gbuf = reinterpret_cast<bf16*>(smem);                            // [2][16][ldg]
    float* gslot = reinterpret_cast<float*>(smem + (size_t)2 * 16 * ldg * sizeof(bf16));   // [2][8H <= 2048]
    float* xch = gslot + 2 * 8 * S256_KP;                                  // per wave [32 units]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x, ubase = 32 * w;

    const bf16* wbase = Wb + (size_t)(ubase + l15) * KP4 + 8 * lq;          // tile ut: + ut*16*KP4 (ragged last tile: see forward)
    bf16x8 a[2][RESB];
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
#pragma unroll
        for (int kb = 0; kb < RESB; ++kb) a[ut][kb] = *reinterpret_cast<const bf16x8*>(wbase + ut * 16 * KP4 + kb * 32);
    bf16x8 sw[2][4];                                                       // group g = (k-blocks RESB + 4*(g>>1) .. +3, tile g&1)
    auto load_group = [&](bf16x8 (&dst)[4], int g) {
        const int kb0 = RESB + 4 * (g >> 1), ut = g & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) dst[i] = *reinterpret_cast<const bf16x8*>(wbase + ut * 16 * KP4 + (kb0 + i) * 32);
    };
    load_group(sw[0], 0);
    load_group(sw[1], 1);
    for (int i = tid; i < 2 * 16 * ldg; i += blockDim.x) gbuf[i] = (bf16)0.f;
    __syncthreads();

    const int du = lane & 31, ud = ubase + du;
    const bool lived = lane < 32 && ud < H;
    const int udc = ud < H ? ud : H - 1;
    // cooperative loader: thread i owns chunk i of the 8H floats [ i f g o | c_t | c_{t-1} | dh_ext | dc_ext ] of a step
    const size_t ostep = (size_t)B * H;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const bool ld_on = tid < 2 * H;
    const int cof = ld_on ? 4 * tid : 0, seg = cof / H, so = cof - seg * H;
    const float* cbase = seg < 4 ? acts + (size_t)b * 4 * H + cof
                       : seg < 6 ? c_all + (size_t)b * H + so
                       : seg == 6 ? (dh_ext ? dh_ext + (size_t)b * H + so : c_all) : (dc_ext ? dc_ext + (size_t)b * H + so : c_all);
    const size_t cstride = seg < 4 ? ostep * 4 : ostep;
    const int cshift = seg == 5 ? 1 : 0;
    const bool czero = (seg == 6 && !dh_ext) || (seg == 7 && !dc_ext);
    f32x4 cfirst = zero4;
    if (ld_on && seg == 5 && c0) cfirst = *reinterpret_cast<const f32x4*>(c0 + (size_t)b * H + so);
    f32x4 cring[PF];
    auto cfetch = [&](f32x4& r, int t) {
        const int tt = (t > 0 ? t : 0) - cshift;
        f32x4 v = *reinterpret_cast<const f32x4*>(cbase + (size_t)(tt > 0 ? tt : 0) * cstride);
        if (tt < 0) v = cfirst;
        r = czero ? zero4 : v;
    };
    {
        f32x4 first;
        cfetch(first, T - 1);
        if (ld_on) *reinterpret_cast<f32x4*>(gslot + cof) = first;
#pragma unroll
        for (int d = 0; d < PF; ++d) cfetch(cring[d], T - 2 - d);
    }
    __syncthreads();

    int cur = 0;
    float* xw = xch + w * 32;
    float dhd = 0.f, dcd = 0.f;
    auto step = [&](int t, f32x4& cslot) {
        const int it = T - 1 - t;
        const float* sl = gslot + (it & 1) * 8 * S256_KP + udc;
        const float ig = sl[0], fg = sl[H], gg = sl[2 * H], og = sl[3 * H], ct = sl[4 * H], cp = sl[5 * H];
        const float dhe = sl[6 * H], dce = sl[7 * H];
        if (ld_on) *reinterpret_cast<f32x4*>(gslot + ((it + 1) & 1) * 8 * S256_KP + cof) = cslot;
        cfetch(cslot, t - 1 - PF);
        const float dh = dhd + dhe;
        const float th = tanh_f(ct);
        const float dct = dcd + dce + dh * og * (1.f - th * th);
        const float dgo = lived ? dh * th * og * (1.f - og) : 0.f;       // selects: a dead lane's dh may be anything
        const float dgi = lived ? dct * gg * ig * (1.f - ig) : 0.f;
        const float dgf = lived ? dct * cp * fg * (1.f - fg) : 0.f;
        const float dgg = lived ? dct * ig * (1.f - gg * gg) : 0.f;
        dcd = lived ? dct * fg : 0.f;
        if (lane < 32) {
            bf16* gw = gbuf + cur * 16 * ldg + ud;                         // row 0 = the sequence
            gw[0] = (bf16)dgi; gw[S256_KP] = (bf16)dgf; gw[2 * S256_KP] = (bf16)dgg; gw[3 * S256_KP] = (bf16)dgo;
        }
        lds_barrier();
        if (lived) {
            float* gp = dG + ((size_t)t * B + b) * 4 * H + ud;
            gp[0] = dgi; gp[H] = dgf; gp[2 * H] = dgg; gp[3 * H] = dgo;
        }
        const bf16* gb = gbuf + cur * 16 * ldg + 8 * lq;                   // only MFMA column 0 is live
        f32x4 acc[2][2] = {{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}};
#pragma unroll
        for (int kb = 0; kb < RESB; ++kb) {
            bf16x8 bfr;
            if (l15 == 0) bfr = *reinterpret_cast<const bf16x8*>(gb + kb * 32);
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) acc[ut][kb & 1] = mfma16(a[ut][kb], bfr, acc[ut][kb & 1]);
        }
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int kb0 = RESB + 4 * (g >> 1), ut = g & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                bf16x8 bfr;
                if (l15 == 0) bfr = *reinterpret_cast<const bf16x8*>(gb + (kb0 + i) * 32);
                acc[ut][i & 1] = mfma16(sw[g & 1][i], bfr, acc[ut][i & 1]);
            }
            load_group(sw[g & 1], (g + 2) % NG);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (l15 == 0) {
#pragma unroll
            for (int ut = 0; ut < 2; ++ut) *reinterpret_cast<f32x4*>(xw + 16 * ut + 4 * lq) = acc[ut][0] + acc[ut][1];
        }
        dhd = xw[du];
        cur ^= 1;
    };
    int tb = T - 1;
    for (; tb - PF + 1 >= 0; tb -= PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) step(tb - d, cring[d]);
    }
#pragma unroll
    for (int d = 0; d < PF; ++d) if (tb - d >= 0) step(tb - d, cring[d]);
    if (lived) {
        if (dh0) dh0[(size_t)b * H + ud] = dhd;
        if (dc0) dc0[(size_t)b * H + ud] = dcd;
    }
}
