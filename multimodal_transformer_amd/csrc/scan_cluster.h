// LSTM scans for hidden sizes 129..256 and batches of up to 32 sequences: FOUR workgroups (CUs) per sequence.
//
// W_rec at H = 256 is 512 KB of bf16 — a CU's whole register file — so one CU cannot keep it resident (scan256.h keeps
// half and streams half: 5.5 / 7.5 us per step).  Here the hidden units of a sequence are split over 4 workgroups of
// 4 waves; each keeps ITS 64 units' rows (128 KB) as MFMA A fragments in registers for the whole scan, and the only
// per-step traffic is the exchange of the new h (forward: 64 bf16 per workgroup) or of the gate gradients (backward:
// 256 bf16 per workgroup) between the four CUs through 8-byte {tag, value} granules written with agent-scope atomic
// stores and polled with agent-scope atomic loads — the data is the flag, no fences (cdna_hip_programming.md, Guideline
// 16, form R2; one hop costs ~1 us, MI355X_MICROARCH.md "handoff-1to1").  tag = step + 1; every granule is zeroed by a
// memset node before each launch (also under hipGraph replay), spins are bounded.
//
// Residency: the 4 workgroups of a sequence must run at the same time.  The host only takes this path for
// 4*B <= 128 workgroups and only when the occupancy query says that twice that many fit the device at once (api.hip:
// cl4_resident), launched as ONE kernel whose block index is decoded so that the four parts of a sequence sit on the same
// XCD.  Nothing can guarantee residency against other streams or processes holding CUs: every wait is bounded and a
// time-out raises a device error word (cl_wait_granule) instead of silently consuming a stale granule.
#pragma once
#include "scan.h"

#define CL_NP 4                         // workgroups per sequence
#define CL_UW 64                        // hidden units per workgroup
#define CL_KP 256                       // padded hidden size
#define CL_SPIN_MAX (1u << 20)               // ~1 s of polling before a wait gives up

typedef unsigned long long cl_u64;

__device__ __forceinline__ void cl_store_granule(cl_u64* g, unsigned tag, unsigned value) {
    __hip_atomic_store(g, ((cl_u64)tag << 32) | value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// One lane polls one granule until its tag matches.  The spin is bounded: a partner workgroup that is not resident (CUs held by
// another stream or process) must not hang the kernel.  On exhaustion the lane raises the device error word `err` (surfaced by
// the callers of mmt_lstm_scan_* at their next synchronisation: the scan's results are then invalid) and stops waiting for the
// rest of the scan (`dead`), so a failed launch drains in about one time-out instead of T of them.
__device__ __forceinline__ unsigned cl_wait_granule(cl_u64* g, unsigned tag, unsigned* err, bool& dead) {
    cl_u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (dead) return (unsigned)x;
    unsigned spins = 0;
    while ((unsigned)(x >> 32) != tag) {
        if (++spins > CL_SPIN_MAX) {
            dead = true;
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return (unsigned)x;
}
__device__ __forceinline__ unsigned cl_pack2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// block index -> (sequence, part): blocks b, b+8, b+16, b+24 of a group of 32 are the four parts of one sequence (same XCD)
__device__ __forceinline__ void cl_decode(int& seq, int& part) {
    const int grp = blockIdx.x >> 5, rem = blockIdx.x & 31;
    part = rem >> 3;
    seq = grp * 8 + (rem & 7);
}

// xb: [B][2 slots][CL_NP parts][32 granules]
template <int PF>
__global__ __launch_bounds__(256) void lstm_scan_fwd_cl4_kernel(const float* __restrict__ gx, const bf16* __restrict__ Wf,
                                                                const float* __restrict__ h0, const float* __restrict__ c0,
                                                                float* __restrict__ h_all, float* __restrict__ c_all,
                                                                float* __restrict__ acts, cl_u64* xb, unsigned* err, int T, int B, int H, int HP16) {
    constexpr int KS = CL_KP / 32, LDH = CL_KP + 8;
    __shared__ __attribute__((aligned(16))) bf16 hbuf[2 * LDH];            // the sequence's h, bf16, double buffered
    __shared__ __attribute__((aligned(16))) float gslot[2 * 4 * CL_UW];    // [slot][gate][own unit]
    int b, part;
    cl_decode(b, part);
    if (b >= B) return;                                                    // whole workgroup
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int ubase = CL_UW * part + 16 * w;                               // first unit of this wave

    const int wr_ = (ubase + l15) < HP16 ? ubase + l15 : HP16 - 1;          // units past the hidden size: any valid row (results unused)
    const bf16* wbase = Wf + (size_t)wr_ * CL_KP + 8 * lq;
    const size_t gs = (size_t)HP16 * CL_KP;
    bf16x8 a[4][KS];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[q][ks] = *reinterpret_cast<const bf16x8*>(wbase + q * gs + ks * 32);

    for (int i = tid; i < 2 * LDH; i += 256) hbuf[i] = (bf16)0.f;
    __syncthreads();
    const int du = lane & 15, ud = ubase + du;
    const bool lived = lane < 16 && ud < H;
    float cd = 0.f;
    if (h0) { for (int i = tid; i < H; i += 256) hbuf[i] = (bf16)h0[(size_t)b * H + i]; }
    if (lived && c0) cd = c0[(size_t)b * H + ud];
    // cooperative input ring: thread i < 64 fetches (gate i>>4, units 4*(i&15)..+3 of this workgroup)
    const int cq = (tid >> 4) & 3, cu = CL_UW * part + 4 * (tid & 15);
    const bool ld_on = tid < 64 && cu < H;
    const float* gxl = gx + (size_t)b * 4 * H + (ld_on ? cq * H + cu : 0);
    const size_t gstep = (size_t)B * 4 * H;
    const int ld_dst = cq * CL_UW + 4 * (tid & 15);
    f32x4 ring[PF];
    auto fetch = [&](f32x4& r, int t) { r = *reinterpret_cast<const f32x4*>(gxl + (size_t)(t < T ? t : T - 1) * gstep); };
    {
        f32x4 first;
        fetch(first, 0);
        if (ld_on) *reinterpret_cast<f32x4*>(gslot + ld_dst) = first;
#pragma unroll
        for (int d = 0; d < PF; ++d) fetch(ring[d], d + 1);
    }
    __syncthreads();

    cl_u64* xseq = xb + (size_t)b * 2 * CL_NP * 32;
    int cur = 0;
    bool dead = false;
    auto step = [&](int t, f32x4& in) {
        float gin[4];
        const float* sl = gslot + (t & 1) * 4 * CL_UW + 16 * w + du;
#pragma unroll
        for (int q = 0; q < 4; ++q) gin[q] = sl[q * CL_UW];
        if (ld_on) *reinterpret_cast<f32x4*>(gslot + ((t + 1) & 1) * 4 * CL_UW + ld_dst) = in;
        fetch(in, t + 1 + PF);

        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        // h W^T (scan.h, cooperative form): A = the h row — only row 0, the sequence, is live —, B = this wave's rows of W_rec, so
        // register 0 of lanes 0..15 holds the pre-activation of the lane's own unit: no re-deal through LDS
        const bf16* hb = hbuf + cur * LDH + 8 * lq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            bf16x8 bf;
            if (l15 == 0) bf = *reinterpret_cast<const bf16x8*>(hb + ks * 32);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = mfma16(bf, a[q][ks], acc[q]);
        }
        const float ig = sigmoid_f(acc[0][0] + gin[0]), fg = sigmoid_f(acc[1][0] + gin[1]);
        const float gg = tanh_f(acc[2][0] + gin[2]), og = sigmoid_f(acc[3][0] + gin[3]);
        cd = fg * cd + ig * gg;
        const float hn = lived ? og * tanh_f(cd) : 0.f;
        // publish this wave's 16 new h values as 8 granules, then collect the other three workgroups' 96
        const float hnb = __shfl_down(hn, 1);
        cl_u64* mine = xseq + ((t & 1) * CL_NP + part) * 32;
        if (lane < 16) {
            hbuf[(cur ^ 1) * LDH + ud] = (bf16)hn;
            if (!(lane & 1)) cl_store_granule(mine + 8 * w + (du >> 1), (unsigned)(t + 1), cl_pack2(hn, hnb));
        }
        if (tid < 96) {
            const int sp = (part + 1 + (tid >> 5)) & 3, g = tid & 31;
            const unsigned v = cl_wait_granule(xseq + ((t & 1) * CL_NP + sp) * 32 + g, (unsigned)(t + 1), err, dead);
            *reinterpret_cast<unsigned*>(hbuf + (cur ^ 1) * LDH + CL_UW * sp + 2 * g) = v;
        }
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

// Backward.  Each workgroup forms dh for its 64 units (rows of Wb) from ALL 4*256 gate gradients of the sequence: its own
// 4*64 come from its dense lanes, the other 3*256 through the exchange.  xb: [B][2 slots][CL_NP parts][128 granules]
// (granule q*32 + u/2 of a part = gate q, units u, u+1 of that part).
template <int PF>
__global__ __launch_bounds__(256) void lstm_scan_bwd_cl4_kernel(const float* __restrict__ dh_ext, const float* __restrict__ dc_ext,
                                                                const bf16* __restrict__ Wb, const float* __restrict__ c0,
                                                                const float* __restrict__ c_all, const float* __restrict__ acts,
                                                                float* __restrict__ dG, float* __restrict__ dh0, float* __restrict__ dc0,
                                                                cl_u64* xb, unsigned* err, int T, int B, int H, int HP16) {
    constexpr int KP4 = 4 * CL_KP, KS4 = KP4 / 32, LDG = KP4 + 8;
    __shared__ __attribute__((aligned(16))) bf16 gbuf[2 * LDG];            // the sequence's gate gradients, k = gate*256 + unit
    __shared__ __attribute__((aligned(16))) float gslot[2 * 8 * CL_UW];    // [slot][segment][own unit]
    int b, part;
    cl_decode(b, part);
    if (b >= B) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, l15 = lane & 15, lq = lane >> 4;
    const int ubase = CL_UW * part + 16 * w;

    const int wr_ = (ubase + l15) < HP16 ? ubase + l15 : HP16 - 1;
    const bf16* wbase = Wb + (size_t)wr_ * KP4 + 8 * lq;
    bf16x8 a[KS4];
#pragma unroll
    for (int kb = 0; kb < KS4; ++kb) a[kb] = *reinterpret_cast<const bf16x8*>(wbase + kb * 32);
    for (int i = tid; i < 2 * LDG; i += 256) gbuf[i] = (bf16)0.f;
    __syncthreads();

    const int du = lane & 15, ud = ubase + du;
    const bool lived = lane < 16 && ud < H;
    // cooperative loader: thread i < 128 owns (segment i>>4, units 4*(i&15)..+3 of this workgroup);
    // segments: 0..3 gate activations i f g o | 4 c_t | 5 c_{t-1} | 6 dh_ext | 7 dc_ext
    const size_t ostep = (size_t)B * H;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int seg = (tid >> 4) & 7, cu = CL_UW * part + 4 * (tid & 15);
    const bool ld_on = tid < 128 && cu < H;
    const int cuc = ld_on ? cu : 0;
    const float* cbase = seg < 4 ? acts + (size_t)b * 4 * H + seg * H + cuc
                       : seg < 6 ? c_all + (size_t)b * H + cuc
                       : seg == 6 ? (dh_ext ? dh_ext + (size_t)b * H + cuc : c_all) : (dc_ext ? dc_ext + (size_t)b * H + cuc : c_all);
    const size_t cstride = seg < 4 ? ostep * 4 : ostep;
    const int cshift = seg == 5 ? 1 : 0;
    const bool czero = (seg == 6 && !dh_ext) || (seg == 7 && !dc_ext);
    f32x4 cfirst = zero4;
    if (ld_on && seg == 5 && c0) cfirst = *reinterpret_cast<const f32x4*>(c0 + (size_t)b * H + cu);
    const int ld_dst = seg * CL_UW + 4 * (tid & 15);
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
        if (ld_on) *reinterpret_cast<f32x4*>(gslot + ld_dst) = first;
#pragma unroll
        for (int d = 0; d < PF; ++d) cfetch(cring[d], T - 2 - d);
    }
    __syncthreads();

    cl_u64* xseq = xb + (size_t)b * 2 * CL_NP * 128;
    int cur = 0;
    bool dead = false;
    float dhd = 0.f, dcd = 0.f;
    auto step = [&](int t, f32x4& cslot) {
        const int it = T - 1 - t;
        const float* sl = gslot + (it & 1) * 8 * CL_UW + 16 * w + du;
        const float ig = sl[0], fg = sl[CL_UW], gg = sl[2 * CL_UW], og = sl[3 * CL_UW], ct = sl[4 * CL_UW], cp = sl[5 * CL_UW];
        const float dhe = sl[6 * CL_UW], dce = sl[7 * CL_UW];
        if (ld_on) *reinterpret_cast<f32x4*>(gslot + ((it + 1) & 1) * 8 * CL_UW + ld_dst) = cslot;
        cfetch(cslot, t - 1 - PF);
        const float dh = dhd + dhe;
        const float th = tanh_f(ct);
        const float dct = dcd + dce + dh * og * (1.f - th * th);
        // selects, not products with 0: a dead lane's dh comes from rows that are not weights and may hold anything
        const float dgo = lived ? dh * th * og * (1.f - og) : 0.f;
        const float dgi = lived ? dct * gg * ig * (1.f - ig) : 0.f;
        const float dgf = lived ? dct * cp * fg * (1.f - fg) : 0.f;
        const float dgg = lived ? dct * ig * (1.f - gg * gg) : 0.f;
        dcd = lived ? dct * fg : 0.f;
        // own gate gradients into the LDS vector and out to the other three workgroups (4 granules per lane pair)
        const float ni = __shfl_down(dgi, 1), nf = __shfl_down(dgf, 1), ng = __shfl_down(dgg, 1), no = __shfl_down(dgo, 1);
        cl_u64* mine = xseq + ((it & 1) * CL_NP + part) * 128;
        bf16* gw = gbuf + cur * LDG;
        if (lane < 16) {
            gw[ud] = (bf16)dgi; gw[CL_KP + ud] = (bf16)dgf; gw[2 * CL_KP + ud] = (bf16)dgg; gw[3 * CL_KP + ud] = (bf16)dgo;
            if (!(lane & 1)) {
                const int g = 8 * w + (du >> 1);
                const unsigned tag = (unsigned)(it + 1);
                cl_store_granule(mine + g, tag, cl_pack2(dgi, ni));
                cl_store_granule(mine + 32 + g, tag, cl_pack2(dgf, nf));
                cl_store_granule(mine + 64 + g, tag, cl_pack2(dgg, ng));
                cl_store_granule(mine + 96 + g, tag, cl_pack2(dgo, no));
            }
        }
#pragma unroll
        for (int rnd = 0; rnd < 2; ++rnd) {
            const int i = tid + 256 * rnd;                                 // 384 granules to collect
            if (i < 384) {
                const int sp = (part + 1 + (i >> 7)) & 3, g = i & 127, q = g >> 5, u2 = g & 31;
                const unsigned v = cl_wait_granule(xseq + ((it & 1) * CL_NP + sp) * 128 + g, (unsigned)(it + 1), err, dead);
                *reinterpret_cast<unsigned*>(gw + q * CL_KP + CL_UW * sp + 2 * u2) = v;
            }
        }
        lds_barrier();
        if (lived) {
            float* gp = dG + ((size_t)t * B + b) * 4 * H + ud;
            gp[0] = dgi; gp[H] = dgf; gp[2 * H] = dgg; gp[3 * H] = dgo;
        }
        const bf16* gb = gw + 8 * lq;
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kb = 0; kb < KS4; ++kb) {
            bf16x8 bfr;
            if (l15 == 0) bfr = *reinterpret_cast<const bf16x8*>(gb + kb * 32);
            acc[kb & 3] = mfma16(bfr, a[kb], acc[kb & 3]);      // dG W: row 0 = the sequence, column = this lane's unit
        }
        dhd = ((acc[0] + acc[1]) + (acc[2] + acc[3]))[0];      // (lanes 0..15; the others hold rows that are not the sequence)
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
