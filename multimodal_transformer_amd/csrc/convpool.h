// Window encoder for gfx950: Conv1d(D -> F, kernel 2, bias) over the W positions of every window followed by a
// global max-pool over the conv positions (transformer/SFT/models.py:57-79; caller :118-132 loops over the batch —
// windows are independent, so the kernels see N = B*T windows).
//
// The conv is a GEMM whose A matrix is an overlapping view of the raw window: conv row p is the contiguous span
// x[p*D .. p*D + 2D) (tap 0 = row p, tap 1 = row p+1), so one staged copy of rows p..p+32 serves both taps:
//     S[p][f] = sum_d x[p][d] w[f][d][0] + x[p+1][d] w[f][d][1]        out[f] = max_p S[p][f] + b[f]
// This is the one MFMA-bound kernel of the model (2*(2D)*F FLOP per position: 29.7 MFLOP per 30-frame vision window).
//
// Forward  (convpool_fwd_kernel): workgroup = 8 windows x 256 channels, 8 waves; wave = 2 windows x 128 channels
//   (8 accumulator tiles of mfma_f32_32x32x16_bf16: rows = conv positions, columns = channels).  Per 32-column chunk
//   of D the raw rows (fp32 -> bf16) and both taps' weights are staged in LDS, double buffered, one barrier per chunk;
//   every A fragment is used by 4 MFMAs and every B fragment by 2.  The max-pool is the epilogue: a column of the
//   accumulator tile is one channel, its rows the positions of one window.
// Backward (convpool_bwd_kernel): inputs are data, so only dW, db exist:
//     dW[f][d][j] = sum_n dy[n][f] * x[n][arg(n,f) + j][d]
//   run as a dense MFMA GEMM over k = (window, position) with the one-hot matrix dy*[p == arg] built in registers
//   as the A operand and the window's rows, transposed to position-major bf16 while staging, as the B operand.
//   Workgroup = 256 channels x 128 raw features x both taps for a slice of the windows; slabs summed afterwards.
#pragma once
#include "common.h"

#define CP_KC 32                       // D columns per forward stage
#define CP_LDX (CP_KC + 8)             // LDS row stride (bf16 elements): 80 B, conflict-free 16-byte fragment reads
#define CP_ROWS 33                     // raw rows staged per 32-position row tile
#define CP_WIN 8                       // windows per forward workgroup
#define CP_FB 256                      // channels per workgroup
#define CP_DB 128                      // raw features per backward workgroup
#define CP_PS 40                       // backward: positions per transposed row (32 + pad)

__host__ __device__ inline size_t convpool_fwd_lds_bytes(int ct) {      // ct = 32-channel tiles per wave: workgroup = 64*ct channels
    return (size_t)2 * (CP_WIN * CP_ROWS * CP_LDX + 2 * 64 * ct * CP_LDX) * sizeof(bf16);
}
__host__ __device__ inline size_t convpool_bwd_lds_bytes() {
    return (size_t)2 * 2 * 2 * CP_DB * CP_PS * sizeof(bf16);      // [buffer][window of the pair][tap][d][position]
}

// weight (F, D, 2) fp32 (nn.Conv1d layout) -> Wp bf16 [2][FPAD][DP], zero padded
__global__ void convpool_prep_kernel(const float* __restrict__ w, bf16* __restrict__ Wp, int F, int D, int FPAD, int DP) {
    const size_t n = (size_t)2 * FPAD * DP;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(idx % DP), f = (int)((idx / DP) % FPAD), tap = (int)(idx / ((size_t)DP * FPAD));
        Wp[idx] = (bf16)((f < F && d < D) ? w[((size_t)f * D + d) * 2 + tap] : 0.f);
    }
}

// CT = channel tiles (32 wide) per wave: the workgroup covers CFB = 64*CT channels starting at c_first + blockIdx.y*CFB.
// CT = 4 for the bulk; a narrower instance finishes channel counts that are not a multiple of 256 (F = 300: 256 + 64).
template <int CT>
__global__ __launch_bounds__(512) void convpool_fwd_kernel(const float* __restrict__ X, const bf16* __restrict__ Wp,
                                                           const float* __restrict__ bias, float* __restrict__ out,
                                                           int* __restrict__ arg, int N, int W, int D, int DP, int F, int FPAD,
                                                           int c_first) {
    constexpr int CFB = 64 * CT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* Xs = reinterpret_cast<bf16*>(smem);                              // [2][CP_WIN][CP_ROWS][CP_LDX]
    bf16* Bs = Xs + 2 * CP_WIN * CP_ROWS * CP_LDX;                         // [2][2][CFB][CP_LDX]
    constexpr int XS_STAGE = CP_WIN * CP_ROWS * CP_LDX, BS_STAGE = 2 * CFB * CP_LDX;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int wpair = wave >> 1, chalf = wave & 1;
    const int n0 = blockIdx.x * CP_WIN, cblk = c_first + blockIdx.y * CFB;
    const int npos = W - 1, nrt = (npos + 31) / 32, nchunk = DP / CP_KC;

    // staging tasks of this thread: X: items tid + 512*i over CP_WIN*CP_ROWS*8 float4;  W: 4 items of 8 bf16
    constexpr int XITEMS = CP_WIN * CP_ROWS * (CP_KC / 4), XPER = (XITEMS + 511) / 512;
    f32x4 xr[XPER];
    bf16x8 wr[CT];

    float best[2][CT];
    int bestp[2][CT];
#pragma unroll
    for (int wi = 0; wi < 2; ++wi)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { best[wi][ct] = -INFINITY; bestp[wi][ct] = 0; }

    for (int rt = 0; rt < nrt; ++rt) {
        auto load_chunk = [&](int ch) {
            const int kc = ch * CP_KC;
#pragma unroll
            for (int i = 0; i < XPER; ++i) {
                const int item = tid + 512 * i;
                const int win = item / (CP_ROWS * 8), rem = item - win * (CP_ROWS * 8), row = rem >> 3, c4 = rem & 7;
                const int n = n0 + win, p = rt * 32 + row, c = kc + 4 * c4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (item < XITEMS && n < N && p < W && c < D) v = *reinterpret_cast<const f32x4*>(X + ((size_t)n * W + p) * D + c);
                xr[i] = v;
            }
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int item = tid + 512 * i;                            // 2 taps * CFB channels * 4 pieces = 512 * CT
                const int tap = item / (CFB * 4), chn = (item >> 2) % CFB, c8 = item & 3;
                wr[i] = *reinterpret_cast<const bf16x8*>(Wp + ((size_t)tap * FPAD + cblk + chn) * DP + kc + 8 * c8);
            }
        };
        auto store_chunk = [&](int stage) {
            bf16* xs = Xs + stage * XS_STAGE;
            bf16* bs = Bs + stage * BS_STAGE;
#pragma unroll
            for (int i = 0; i < XPER; ++i) {
                const int item = tid + 512 * i;
                if (item < XITEMS) {
                    const int win = item / (CP_ROWS * 8), rem = item - win * (CP_ROWS * 8), row = rem >> 3, c4 = rem & 7;
                    const f32x2 lo = {xr[i][0], xr[i][1]}, hi = {xr[i][2], xr[i][3]};
                    uint2 pk;
                    pk.x = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf16x2));
                    pk.y = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, bf16x2));
                    *reinterpret_cast<uint2*>(xs + (win * CP_ROWS + row) * CP_LDX + 4 * c4) = pk;
                }
            }
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                const int item = tid + 512 * i;
                const int tap = item / (CFB * 4), chn = (item >> 2) % CFB, c8 = item & 3;
                *reinterpret_cast<bf16x8*>(bs + (tap * CFB + chn) * CP_LDX + 8 * c8) = wr[i];
            }
        };

        f32x16 acc[2][CT];
#pragma unroll
        for (int wi = 0; wi < 2; ++wi)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[wi][ct][i] = 0.f;

        __syncthreads();                                                  // previous row tile's readers are done
        load_chunk(0);
        store_chunk(0);
        __syncthreads();
        for (int ch = 0; ch < nchunk; ++ch) {
            const int stage = ch & 1;
            // global loads of the next chunk in flight behind this chunk's MFMAs.  Branch-free on purpose (the last iteration
            // re-fetches its own chunk into the idle stage): no skippable block may sit between MFMAs and their readers
            load_chunk(ch + 1 < nchunk ? ch + 1 : ch);
            const bf16* xs = Xs + stage * XS_STAGE + (2 * wpair * CP_ROWS + r) * CP_LDX + 8 * hh;
            const bf16* bs = Bs + stage * BS_STAGE + (chalf * 32 * CT + r) * CP_LDX + 8 * hh;
#pragma unroll
            for (int tap = 0; tap < 2; ++tap) {
#pragma unroll
                for (int ks = 0; ks < CP_KC / 16; ++ks) {
                    bf16x8 a[2];
#pragma unroll
                    for (int wi = 0; wi < 2; ++wi)
                        a[wi] = *reinterpret_cast<const bf16x8*>(xs + (wi * CP_ROWS + tap) * CP_LDX + ks * 16);
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const bf16x8 b = *reinterpret_cast<const bf16x8*>(bs + (tap * CFB + ct * 32) * CP_LDX + ks * 16);
#pragma unroll
                        for (int wi = 0; wi < 2; ++wi) acc[wi][ct] = mfma32(a[wi], b, acc[wi][ct]);
                    }
                }
            }
            store_chunk(stage ^ 1);
            __syncthreads();
        }

        // ---- max-pool epilogue of this row tile: column r of a tile = channel, rows = positions rt*32 + row
        const int plim = npos - rt * 32;                                   // positions >= plim are padding
#pragma unroll
        for (int wi = 0; wi < 2; ++wi)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                float bv = -INFINITY;
                int bp = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int row = acc32_row(i, hh);
                    const float v = row < plim ? acc[wi][ct][i] : -INFINITY;
                    if (v > bv) { bv = v; bp = row; }                     // rows ascend with i inside a lane: first maximum wins
                }
                const float ov = __shfl_xor(bv, 32);
                const int op = __shfl_xor(bp, 32);
                if (ov > bv || (ov == bv && op < bp)) { bv = ov; bp = op; }
                bp += rt * 32;
                if (bv > best[wi][ct]) { best[wi][ct] = bv; bestp[wi][ct] = bp; }
            }
    }

    if (hh == 0) {
#pragma unroll
        for (int wi = 0; wi < 2; ++wi) {
            const int n = n0 + 2 * wpair + wi;
            if (n >= N) continue;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const int c = cblk + chalf * 32 * CT + ct * 32 + r;
                if (c < F) {
                    out[(size_t)n * F + c] = best[wi][ct] + bias[c];
                    arg[(size_t)n * F + c] = bestp[wi][ct];
                }
            }
        }
    }
}

// (Measured and rejected for the forward kernel: a one-wave-per-SIMD variant with 16 accumulator tiles in AGPRs (spills, 1.4x
// slower), a shift-only staging index map and branch-free clamped loads (both 1.25x slower on the vision shape than the
// predicated loads below, same box).)
// grid (ceil(D/128), nsplit, FPAD/256); block 512.  slab layout [split][tap][FPAD][DPB] with DPB = 128*gridDim.x.
// ONE_RT: windows of at most 33 rows (every reference shape): item = window pair, no index divisions in the loop
template <bool ONE_RT>
__global__ __launch_bounds__(512) void convpool_bwd_kernel(const float* __restrict__ X, const float* __restrict__ dy,
                                                           const int* __restrict__ arg, float* __restrict__ slab,
                                                           float* __restrict__ dbpart,
                                                           int N, int W, int D, int F, int FPAD, int wins_per_split) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* XT = reinterpret_cast<bf16*>(smem);                              // [2 buffers][2 windows][2 taps][CP_DB][CP_PS]
    constexpr int XT_TAP = CP_DB * CP_PS, XT_WIN = 2 * XT_TAP, XT_BUF = 2 * XT_WIN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, hh = lane >> 5;
    const int fq = wave >> 1, dhalf = wave & 1;                            // wave tile: f tiles 2fq, 2fq+1; d tiles 2dhalf, 2dhalf+1
    const int d0 = blockIdx.x * CP_DB, cblk = blockIdx.z * CP_FB;
    const int nbeg = blockIdx.y * wins_per_split, nend = min(N, nbeg + wins_per_split);
    const int DPB = CP_DB * gridDim.x;
    const int npos = W - 1, nrt = (npos + 31) / 32;

    // staging role: 4 waves per window slot; lane = g + 8*dd: rows 4g..4g+4, float4 column dd of this wave's 32 columns
    const int swin = wave >> 2, sw4 = wave & 3, g = lane & 7, dd = lane >> 3;
    const int sd = sw4 * 32 + 4 * dd;                                      // first of 4 raw features (within the 128 block)
    f32x4 xr[5];

    f32x16 acc[2][2][2];                                                   // [tap][f tile][d tile]
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[t][a][b][i] = 0.f;

    // work items: (window pair, row tile); each item stages two windows and runs 2 x 16 MFMAs per wave
    const int npairs = (nend - nbeg + 1) / 2;
    const int nitems = npairs > 0 ? npairs * nrt : 0;
    float dyn[2][2];                                                       // next item: dy and (arg - row tile base) of this lane's
    int avn[2][2];                                                         // channels, [window of the pair][f tile]
    auto load_rows = [&](int it) {
        const int pr = ONE_RT ? it : it / nrt, rt = ONE_RT ? 0 : it - pr * nrt;
        const int n = nbeg + 2 * pr + swin;
        const bool ok = n < nend && (d0 + sd) < D;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int p = rt * 32 + 4 * g + i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok && p < W) v = *reinterpret_cast<const f32x4*>(X + ((size_t)n * W + p) * D + d0 + sd);
            xr[i] = v;
        }
    };
    auto load_dy = [&](int it) {
        const int pr = ONE_RT ? it : it / nrt, rt = ONE_RT ? 0 : it - pr * nrt;
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int nn = nbeg + 2 * pr + w2, f = cblk + (2 * fq + a) * 32 + r;
                const bool okf = nn < nend && f < F;
                dyn[w2][a] = okf ? dy[(size_t)nn * F + f] : 0.f;
                avn[w2][a] = okf ? arg[(size_t)nn * F + f] - rt * 32 : -1;
            }
    };
    auto store_item = [&](int buf) {
        bf16* base = XT + buf * XT_BUF + swin * XT_WIN;
#pragma unroll
        for (int c = 0; c < 4; ++c) {                                      // raw feature sd + c: positions 4g..4g+3 of both taps
            const f32x2 a0 = {xr[0][c], xr[1][c]}, a1 = {xr[2][c], xr[3][c]};
            const f32x2 b0 = {xr[1][c], xr[2][c]}, b1 = {xr[3][c], xr[4][c]};
            uint2 t0, t1;
            t0.x = __builtin_bit_cast(unsigned, __builtin_convertvector(a0, bf16x2));
            t0.y = __builtin_bit_cast(unsigned, __builtin_convertvector(a1, bf16x2));
            t1.x = __builtin_bit_cast(unsigned, __builtin_convertvector(b0, bf16x2));
            t1.y = __builtin_bit_cast(unsigned, __builtin_convertvector(b1, bf16x2));
            *reinterpret_cast<uint2*>(base + (sd + c) * CP_PS + 4 * g) = t0;
            *reinterpret_cast<uint2*>(base + XT_TAP + (sd + c) * CP_PS + 4 * g) = t1;
        }
    };

    // Pipeline: rows are fetched two items ahead of the MFMAs (registers), transposed into the idle LDS buffer one item
    // ahead; dy / argmax one item ahead.  One barrier per item; the loop body has no skippable blocks (clamped re-fetches).
    float dyc[2][2];
    int avc[2][2];
    float dbacc[2] = {0.f, 0.f};
    if (nitems > 0) {
        load_rows(0);
        load_dy(0);
        store_item(0);
        load_rows(nitems > 1 ? 1 : 0);
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
            for (int a = 0; a < 2; ++a) { dyc[w2][a] = dyn[w2][a]; avc[w2][a] = avn[w2][a]; }
    }
    __syncthreads();
    for (int it = 0; it < nitems; ++it) {
        store_item((it + 1) & 1);                                          // rows of item it+1 (readers of that buffer: item it-1, done)
        load_rows(it + 2 < nitems ? it + 2 : nitems - 1);
        load_dy(it + 1 < nitems ? it + 1 : it);
        const bf16* XTb = XT + (it & 1) * XT_BUF;
        const float first_rt = (ONE_RT || it % nrt == 0) ? 1.f : 0.f;      // bias gradient: every (window, channel) once
#pragma unroll
        for (int a = 0; a < 2; ++a) dbacc[a] += first_rt * (dyc[0][a] + dyc[1][a]);
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2) {
            // one-hot A fragments: lane (r, hh) holds channel f, positions rt*32 + ks*16 + 8hh + j
            const float* dyv = dyc[w2];
            const int* av = avc[w2];
            const bf16* xt = XTb + w2 * XT_WIN + (dhalf * 64 + r) * CP_PS + 8 * hh;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 afr[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    const int j = av[a] - ks * 16 - 8 * hh;                // element index inside this fragment, or out of range
                    const unsigned bits = (unsigned)__builtin_bit_cast(unsigned short, (bf16)dyv[a]);
                    const unsigned val = bits << ((j & 1) << 4);           // the value in its half of a dword
                    const int e0 = j >> 1;                                 // dword index (arithmetic shift: out of range stays so)
                    u32x4_t q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = (e0 == e) ? val : 0u;
                    afr[a] = __builtin_bit_cast(bf16x8, q);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(xt + t * XT_TAP + b * 32 * CP_PS + ks * 16);
#pragma unroll
                        for (int a = 0; a < 2; ++a) acc[t][a][b] = mfma32(afr[a], bfr, acc[t][a][b]);
                    }
            }
        }
#pragma unroll
        for (int w2 = 0; w2 < 2; ++w2)
#pragma unroll
            for (int a = 0; a < 2; ++a) { dyc[w2][a] = dyn[w2][a]; avc[w2][a] = avn[w2][a]; }
        __syncthreads();
    }

    if (blockIdx.x == 0 && dhalf == 0 && hh == 0) {                        // one wave per f-tile pair owns the bias partials
#pragma unroll
        for (int a = 0; a < 2; ++a) dbpart[(size_t)blockIdx.y * FPAD + cblk + (2 * fq + a) * 32 + r] = dbacc[a];
    }
    // D tile: column = lane r = raw feature, rows = channels
    float* sl = slab + (size_t)blockIdx.y * 2 * FPAD * DPB;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int f = cblk + (2 * fq + a) * 32 + acc32_row(i, hh);
                    const int d = d0 + dhalf * 64 + b * 32 + r;
                    sl[((size_t)t * FPAD + f) * DPB + d] = acc[t][a][b][i];
                }
}

// dweight (F, D, 2) = sum over splits of slab[split][tap][f][d];  dbias[f] = sum over splits of dbpart[split][f]
__global__ void convpool_finish_kernel(const float* __restrict__ slab, const float* __restrict__ dbpart, float* __restrict__ dweight,
                                       float* __restrict__ dbias, int nsplit, int D, int F, int FPAD, int DPB) {
    const size_t nw = (size_t)F * D * 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < nw + F; idx += (size_t)gridDim.x * blockDim.x) {
        if (idx < nw) {
            const int tap = (int)(idx & 1), d = (int)((idx >> 1) % D), f = (int)((idx >> 1) / D);
            float s = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) s += slab[(((size_t)sp * 2 + tap) * FPAD + f) * DPB + d];
            dweight[idx] = s;
        } else {
            const int f = (int)(idx - nw);
            float s = 0.f;
            for (int sp = 0; sp < nsplit; ++sp) s += dbpart[(size_t)sp * FPAD + f];
            dbias[f] = s;
        }
    }
}
