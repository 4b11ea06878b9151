// Data movement and the small element-wise steps between the GEMMs and the scans of the sequence models, so that a model's forward and
// backward issue no library (ATen) kernel:
//   * copy2d_kernel   — up to MMT_COPY_MAX_SEGS strided 2-D copies per launch: column concatenation and its split (the MFN's cStar
//                       = [c_{t-1} of every modality ; c_t of every modality] and [h ; mem], transformer/MFT/multiTransformer.py:212-217,241-243),
//                       the one-step time shift of c, batch-major <-> time-major row permutation ((B,T,d) <-> (T,B,d), :300), column slices
//                       of weights (gamma fc1 = [attended part | memory part], :221-223), row broadcasts, sums of two sources, row scaling
//                       by the window mask (:310), zero fill, accumulation into the destination;
//   * softmax_mul     — attended = softmax(logits) * cStar over the feature axis and its backward (:218-219);
//   * colsum          — column sums of a few rows (gradient of a row that was broadcast over the batch).
// All HBM-bound one-pass kernels: 16-byte accesses when every pointer / stride of a segment allows, else 4-byte.
#pragma once
#include "common.h"

#define MMT_COPY_MAX_SEGS 24
// perm: 0 none; 1: the source is batch-major (row r = b*T + t), the destination time-major (row t*B + b); 2: the reverse.
// rowscale (optional) is indexed by the BATCH-major row in both permuted modes, by the row otherwise.
// src == nullptr: zeros.  src2 (optional): added to src.  src_ld / src2_ld == 0: one row broadcast over all rows.  accumulate: dst += value.
struct CopySeg {
    const float* src; const float* src2; float* dst; const float* rowscale;
    int rows, cols, src_ld, src2_ld, dst_ld, perm, pB, pT, accumulate, vec4;
};
struct CopySegs { CopySeg s[MMT_COPY_MAX_SEGS]; int n; };

__global__ __launch_bounds__(256) void copy2d_kernel(const CopySegs segs) {
    const CopySeg& g = segs.s[blockIdx.y];
    const int cpr = g.vec4 ? (g.cols >> 2) : g.cols;                   // work items per row
    const size_t total = (size_t)g.rows * cpr;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int r = (int)(idx / cpr), c = (int)(idx - (size_t)r * cpr) * (g.vec4 ? 4 : 1);
        int rd = r, rb = r;                                             // destination row, batch-major row (for rowscale)
        if (g.perm == 1) { const int b = r / g.pT, t = r - b * g.pT; rd = t * g.pB + b; }
        else if (g.perm == 2) { const int t = r / g.pB, b = r - t * g.pB; rd = b * g.pT + t; rb = rd; }
        const float rs = g.rowscale ? g.rowscale[rb] : 1.f;
        float* d = g.dst + (size_t)rd * g.dst_ld + c;
        if (g.vec4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (g.src) v = *reinterpret_cast<const f32x4*>(g.src + (size_t)r * g.src_ld + c);
            if (g.src2) v += *reinterpret_cast<const f32x4*>(g.src2 + (size_t)r * g.src2_ld + c);
            v *= rs;
            if (g.accumulate) v += *reinterpret_cast<const f32x4*>(d);
            *reinterpret_cast<f32x4*>(d) = v;
        } else {
            float v = g.src ? g.src[(size_t)r * g.src_ld + c] : 0.f;
            if (g.src2) v += g.src2[(size_t)r * g.src2_ld + c];
            v *= rs;
            if (g.accumulate) v += *d;
            *d = v;
        }
    }
}

// attended[m][n] = softmax_n(logits[m][:])[n] * v[m][n]; att is kept for the backward.  One wave per row, 4 rows per workgroup.
__global__ __launch_bounds__(256) void softmax_mul_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ v,
                                                              float* __restrict__ att, float* __restrict__ out, int M, int N) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* lr = logits + (size_t)m * N;
    float mx = -INFINITY;
    for (int n = lane; n < N; n += 64) mx = fmaxf(mx, lr[n]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int n = lane; n < N; n += 64) sum += __expf(lr[n] - mx);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
    for (int n = lane; n < N; n += 64) {
        const float a = __expf(lr[n] - mx) * inv;
        att[(size_t)m * N + n] = a;
        out[(size_t)m * N + n] = a * v[(size_t)m * N + n];
    }
}
// g = dout * v;  dlogits = att * (g - sum_n g att);  dv = dout * att
__global__ __launch_bounds__(256) void softmax_mul_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ att,
                                                              const float* __restrict__ v, float* __restrict__ dlogits,
                                                              float* __restrict__ dv, int M, int N) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const size_t o0 = (size_t)m * N;
    float dot = 0.f;
    for (int n = lane; n < N; n += 64) dot += dout[o0 + n] * v[o0 + n] * att[o0 + n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    for (int n = lane; n < N; n += 64) {
        const float a = att[o0 + n], d = dout[o0 + n];
        dlogits[o0 + n] = a * (d * v[o0 + n] - dot);
        dv[o0 + n] = d * a;
    }
}

// Highway combine of the window encoder + the Dropout(0.3) behind it (transformer/SFT/models.py:47-55, 132-134):
//   out = drop(gate * proj + (1 - gate) * x)          gate = sigmoid(linear_gate(x)), proj = linear_projection(x): the row GEMMs' outputs
// and its backward (the mask is regenerated from the same counters: stream MMT_HIGHWAY_STREAM, index = element; nothing is stored):
//   g = drop'(dout);  dx = g (1 - gate);  dproj = g gate;  dgate = g (proj - x)
// One index PAIR per thread and step (one hash word decides both elements, common.h); an odd n leaves one single element.
#define MMT_HIGHWAY_STREAM 3000u
__global__ __launch_bounds__(256) void highway_fwd_kernel(const float* __restrict__ x, const float* __restrict__ proj, const float* __restrict__ gate,
                                                          float* __restrict__ out, size_t n, DropCfg c, const uint64_t* __restrict__ seedblock) {
    c = drop_resolve(c, seedblock);
    const size_t npair = (n + 1) >> 1;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < npair; q += (size_t)gridDim.x * 256) {
        const size_t i = 2 * q;
        const bool two = i + 1 < n;
        const float x0 = x[i], x1 = two ? x[i + 1] : 0.f, g0 = gate[i], g1 = two ? gate[i + 1] : 0.f;
        float y0 = fmaf(g0, proj[i] - x0, x0), y1 = two ? fmaf(g1, proj[i + 1] - x1, x1) : 0.f;
        if (c.thr16) { const uint32_t w = drop_pair(c, i); y0 = drop_lo(c, w, y0); y1 = drop_hi(c, w, y1); }
        out[i] = y0;
        if (two) out[i + 1] = y1;
    }
}
__global__ __launch_bounds__(256) void highway_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ x, const float* __restrict__ proj,
                                                          const float* __restrict__ gate, float* __restrict__ dx, float* __restrict__ dproj,
                                                          float* __restrict__ dgate, size_t n, DropCfg c, const uint64_t* __restrict__ seedblock) {
    c = drop_resolve(c, seedblock);
    const size_t npair = (n + 1) >> 1;
    for (size_t q = (size_t)blockIdx.x * 256 + threadIdx.x; q < npair; q += (size_t)gridDim.x * 256) {
        const size_t i = 2 * q;
        const bool two = i + 1 < n;
        float d0 = dout[i], d1 = two ? dout[i + 1] : 0.f;
        if (c.thr16) { const uint32_t w = drop_pair(c, i); d0 = drop_lo(c, w, d0); d1 = drop_hi(c, w, d1); }
        const float g0 = gate[i], x0 = x[i];
        dx[i] = d0 * (1.f - g0); dproj[i] = d0 * g0; dgate[i] = d0 * (proj[i] - x0);
        if (two) { const float g1 = gate[i + 1], x1 = x[i + 1]; dx[i + 1] = d1 * (1.f - g1); dproj[i + 1] = d1 * g1; dgate[i + 1] = d1 * (proj[i + 1] - x1); }
    }
}

// out[c] = sum_r x[r][c] in a fixed order (rows is small: a batch)
__global__ void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, int rows, int cols, int ld) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += x[(size_t)r * ld + c];
    out[c] = s;
}

// accum[0] |= word[0]: a device error word (e.g. the four-CU LSTM scans' time-out flag, first word of their workspace) folded into a word
// that outlives the workspace, as a kernel of the launch sequence — so it is ordered like the kernels around it in a captured hipGraph
// too (a device-to-host copy node recorded behind the scan was seen to run as a root node of the replayed graph).
__global__ void error_accumulate_kernel(const unsigned* __restrict__ word, unsigned* __restrict__ accum) {
    if (word[0]) atomicOr(accum, word[0]);
}

// p[0 .. n) = 0 (dwords).  Used instead of hipMemsetAsync: under hipGraph replay on ROCm 7.2 a captured memset node was seen to leave
// other bytes than the ones asked for from the second replay on (tools note in DESIGN.md), and the four-CU scans rely on their exchange
// granules and error word being cleared before every launch.
__global__ void zero_fill_kernel(unsigned* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
