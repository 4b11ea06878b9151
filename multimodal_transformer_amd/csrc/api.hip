// Host side of libmmt_hip.so: workspace carving, kernel sequencing, and the extern "C" boundary
// declared in include/mmt_hip.h.  Nothing here allocates or synchronises; everything is enqueued on
// the caller's stream so a whole training step can be captured into one hipGraph.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>

#include "../../include/mmt_hip.h"
#include "common.h"
#include "rowgemm.h"
#include "attn_mask.h"
#include "attn.h"
#include "attn_bwd_diag.h"
#include "misc_kernels.h"
#include "scan.h"
#include "scan_units.h"
#include "scan256.h"
#include "scan_cluster.h"
#include "convpool.h"
#include "glue.h"

// ------------------------------------------------------------------------------------ error plumbing
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}
#define HIP_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) \
    return fail(MMT_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)
#define LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) \
    return fail(MMT_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); } while (0)

// ------------------------------------------------------------------------------------ per-kernel timing
// Optional HIP-event bracket around every launch site (eager mode only), so bench.py can report the
// dominant kernel's average duration on the stream the kernels really run on.
enum Site { S_PREP = 0, S_LN1_QKV, S_ATTN_FWD, S_OUTPROJ, S_LN2_FFN1, S_FFN2, S_LN_FWD, S_LN_BWD, S_BWD_FFN2, S_BWD_FFN1_LN2,
            S_BWD_OUTPROJ, S_ATTN_BWD, S_ATTN_BWD_DQ, S_BWD_QKV_LN1, S_WGRAD, S_FINALIZE, S_OTHER,
            S_LINEAR_FWD, S_LINEAR_BWD_DX, S_LINEAR_WGRAD, S_LSTM_FWD, S_LSTM_BWD, S_MEM_FWD, S_MEM_BWD, S_CONV_FWD, S_CONV_BWD, S_CHAIN4_FWD, S_ATTN_BWD_FUSED, S_MASK_GEN, S_BWD_BOUNDARY, S_COUNT };
static const char* const g_site_names[S_COUNT] = {
    "encoder_prep_kernel", "rowgemm<FRAG,LN>:ln1+qkv", "attn_fwd_kernel", "chain:outproj+res>ln2+ffn1>ffn2+res",
    "rowgemm<PLAIN,LN>:ln2+ffn1+relu", "rowgemm<PLAIN>:ffn2+res", "layernorm_fwd_kernel", "layernorm_bwd_kernel",
    "chain:bwd_ffn2>bwd_ffn1+ln2>bwd_outproj->dO", "rowgemm<LNBWD>:bwd_ffn1+ln2", "rowgemm<FRAG>:bwd_outproj->dO", "attn_bwd_dkv_kernel",
    "attn_bwd_dq_kernel", "rowgemm<LNBWD>:bwd_qkv+ln1", "wgrad_kernel", "finalize_kernels", "other",
    "rowgemm<PLAIN>:linear_fwd", "rowgemm<PLAIN>:linear_bwd_dx", "wgrad_kernel:linear", "lstm_scan_fwd_kernel", "lstm_scan_bwd_kernel",
    "mfn_mem_scan_fwd_kernel", "mfn_mem_scan_bwd_kernel", "convpool_fwd_kernel", "convpool_bwd_kernel",
    "chain:outproj+res>ln2+ffn1>ffn2+res>ln1+qkv(next)", "attn_bwd_diag16_kernel", "attn_mask_gen_kernel",
    "chain:bwd_qkv+ln1>bwd_ffn2(below)>bwd_ffn1+ln2>bwd_outproj->dO"};
struct ProfRec { int site; hipEvent_t a, b; };
static bool g_prof = false;
static ProfRec* g_recs = nullptr;
static int g_nrec = 0, g_cap = 0;
struct ProfScope {
    hipStream_t st; int idx;
    ProfScope(int site, hipStream_t s) : st(s), idx(-1) {
        if (!g_prof) return;
        if (g_nrec == g_cap) {
            const int ncap = g_cap ? g_cap * 2 : 1024;
            ProfRec* n = static_cast<ProfRec*>(realloc(g_recs, sizeof(ProfRec) * ncap));
            if (!n) return;
            for (int i = g_cap; i < ncap; ++i) { n[i].a = nullptr; n[i].b = nullptr; }
            g_recs = n; g_cap = ncap;
        }
        ProfRec& r = g_recs[g_nrec];
        if (!r.a && (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess)) return;
        r.site = site; idx = g_nrec++;
        hipEventRecord(r.a, st);
    }
    ~ProfScope() { if (idx >= 0) hipEventRecord(g_recs[idx].b, st); }
};

extern "C" int mmt_profile_enable(int on) { g_prof = on != 0; return MMT_OK; }
extern "C" int mmt_profile_reset(void) { g_nrec = 0; return MMT_OK; }
extern "C" int mmt_profile_num_sites(void) { return S_COUNT; }
extern "C" const char* mmt_profile_site_name(int site) { return (site >= 0 && site < S_COUNT) ? g_site_names[site] : ""; }
extern "C" int mmt_profile_collect(float* total_ms, int* launches) {
    if (!total_ms || !launches) return fail(MMT_EINVAL, "null pointer argument");
    for (int i = 0; i < S_COUNT; ++i) { total_ms[i] = 0.f; launches[i] = 0; }
    for (int i = 0; i < g_nrec; ++i) {
        HIP_TRY(hipEventSynchronize(g_recs[i].b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_recs[i].a, g_recs[i].b));
        total_ms[g_recs[i].site] += ms; launches[g_recs[i].site] += 1;
    }
    return MMT_OK;
}

extern "C" int mmt_abi_version(void) { return 1; }
extern "C" const char* mmt_last_error(void) { return g_err; }

// ------------------------------------------------------------------------------------ workspace carving
struct Carver {
    char* base; size_t off;
    explicit Carver(void* b) : base(static_cast<char*>(b)), off(0) {}
    template <typename T> T* take(size_t n) {
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += (n * sizeof(T) + 255) / 256 * 256;
        return p;
    }
};

static constexpr int MAX_LAYERS = 16;

struct EncDims {
    int B, T, d, h, f, N;
    int M, MP, Tp, nt, G, GR, nsplit, mchunk, M16;      // G: 32-window tiles (LayerNorm kernels), GR: row-kernel tiles of MMT_ROWS
    LayerLayout L;
};

static int make_dims(EncDims& D, int B, int T, int d, int h, int f, int N) {
    if (B <= 0 || T <= 0 || d <= 0 || h <= 0 || f <= 0 || N < 0) return fail(MMT_EINVAL, "non-positive dimension");
    if (N > MAX_LAYERS) return fail(MMT_EUNSUPPORTED, "n_layers %d > %d", N, MAX_LAYERS);
    if (d % h) return fail(MMT_EINVAL, "d_model %d not divisible by h %d", d, h);   // multiTransformer.py:39
    if (d % 4 || f % 4) return fail(MMT_EUNSUPPORTED, "d_model and d_ff must be multiples of 4 (got %d, %d)", d, f);
    if (d / h > 64) return fail(MMT_EUNSUPPORTED, "d_k = %d > 64 not supported", d / h);
    if (d < 2) return fail(MMT_EINVAL, "d_model < 2 (unbiased std undefined)");
    D.B = B; D.T = T; D.d = d; D.h = h; D.f = f; D.N = N;
    D.M = B * T; D.MP = round_up(D.M, 64); D.M16 = round_up(D.M, 16);
    D.Tp = round_up(T, 32); D.nt = D.Tp / 32; D.G = (D.M + 31) / 32; D.GR = (D.M + MMT_ROWS - 1) / MMT_ROWS;
    D.L = make_layout(d, f, h);
    {   // the row kernels keep a whole window tile in LDS: refuse here what their launches would refuse later (launch_rowchain /
        // launch_rowgemm), so a shape fails at the workspace query and never half-way through a training step
        const LayerLayout& L = D.L;
        const int kmax = std::max(std::max(L.DP, L.FP), L.HDP), fw = std::max(128, L.DP);
        const bool no_gs = L.DP > 128;
        const size_t bwd_chain = (size_t)MMT_ROWS * ((kmax + 8) * 2 + (fw + 4) * 4 * (no_gs ? 1 : 2) + (L.FP + 8) * 2);
        const size_t fwd_chain = (size_t)MMT_ROWS * ((kmax + 8) * 2 + (128 + 4) * 4 + (L.DP + 4) * 4 + (L.FP + 8) * 2);
        const size_t bwd_qkv = rowgemm_lds_bytes(EPI_LNBWD, false, L.NQ, L.DP, L.NQ > 512 ? 512 : 0, no_gs || L.NQ > 512);
        const int bk = std::max(std::max(L.NQ > 512 ? 512 : L.NQ, L.DP), L.FP);
        const size_t boundary = (size_t)MMT_ROWS * ((bk + 8) * 2 + (fw + 4) * 4 * ((no_gs || L.NQ > 512) ? 1 : 2) + (L.FP + 8) * 2);
        const size_t worst = std::max(std::max(std::max(bwd_chain, fwd_chain), bwd_qkv), boundary);
        if (worst > 160 * 1024)
            return fail(MMT_EUNSUPPORTED, "d_model %d / d_ff %d: a %d-window tile of the row kernels needs %zu B of LDS (160 KB per CU)",
                        d, f, MMT_ROWS, worst);
    }
    // weight-gradient split over windows: ONE launch covers every layer.  The launch deals (layer, split) units of `tpl` tiles
    // round-robin to the 8 XCDs (wgrad_kernel), and an XCD holds 32 CUs x 3 workgroups (48 KB of LDS each) at a time: pick the
    // split count that minimises (dispatch rounds on the fullest XCD) x (64-window chunks per workgroup + overhead).
    const LayerLayout& L = D.L;
    const int tpl = (L.NQ / 64) * (L.DP / 64) + (L.DP / 64) * (L.HDP / 64) + 2 * (L.FP / 64) * (L.DP / 64);
    const int nl = N > 0 ? N : 1;
    int s = 1; long best = -1;
    for (int s2 = 1; s2 <= 32; ++s2) {
        const int chunks = (round_up((D.MP + s2 - 1) / s2, 64)) / 64;
        const int per_xcd = (nl * s2 + 7) / 8 * tpl;
        // per workgroup ~4 chunk-times of prologue/slab store; per extra split ~0.7 chunk-times in the slab sums (measured, C4/C3e)
        const long cost = (long)((per_xcd + 95) / 96) * (chunks + 4) * 10 + 7 * s2;
        if (best < 0 || cost < best) { best = cost; s = s2; }
    }
    D.mchunk = round_up((D.MP + s - 1) / s, 64);
    D.nsplit = (D.MP + D.mchunk - 1) / D.mchunk;
    return MMT_OK;
}

// Which attention backward runs for this shape: the one-kernel form reads every operand in the R layout only; the two-kernel form
// also needs the T (transposed) fragment layouts of Q, K and dO, which are allocated and written only then.
static bool use_fused_bwd(const EncDims& D) {
    static const bool allowed = getenv("MMT_NO_FUSED_ATTN_BWD") == nullptr;
    return allowed && attn_bwd_fused_ok(D.L.DKP, D.nt);
}

struct LayerWs {
    float *xout, *x1, *stats1, *stats2, *lse;
    bf16 *xn1, *xn2, *QR, *KR, *VR, *ctx, *hid;      // xn1, xn2, ctx, hid: row-major [MP][pad], operands of wgrad
    // backward operands of the weight-gradient GEMMs (row-major bf16 [MP][pad]), kept per layer so ONE batched launch forms every layer's dW
    bf16 *dx2, *dh, *dx1, *dqkv;
    float *lnpart1, *lnpart2;
    uint16_t *maskQ, *maskK;                // attention-dropout lane words of this layer (attn_mask.h)
};
struct EncWs {
    uint64_t* seedword;                      // device-resident dropout seed of the forward that last used this workspace (256-byte slot)
    bf16* wprep; float* bprep; float* statsf;
    LayerWs lw[MAX_LAYERS];
    // backward scratch (shared by all layers; single stream)
    float *dxa, *dxb, *delta, *lnpartf;
    bf16 *dOR;
    float *sWqkv, *sbqkv, *sWo, *sbo, *sW1, *sb1, *sW2, *sb2;   // slab sets of layer 0; layer l at + l * slab_stride
    size_t slab_stride;
    size_t bytes;
};

// `train`: with the attention-dropout bit masks (2 orientations x N layers x B*h x Tp^2/4 bytes: 1.9 GB at T=2500, B=25).  They are the
// LAST region, so an eval-mode workspace is a prefix of a train-mode one and a long-context evaluation never pays for them.
static void carve_encoder(EncWs& W, const EncDims& D, void* base, bool train = true) {
    Carver c(base);
    const LayerLayout& L = D.L;
    const size_t M = D.M, MP = D.MP, BH = (size_t)D.B * D.h;
    W.seedword = c.take<uint64_t>(MMT_SEED_BLOCK_WORDS);
    W.wprep = c.take<bf16>(L.pstride() * (size_t)(D.N > 0 ? D.N : 1));
    W.bprep = c.take<float>(L.qstride() * (size_t)(D.N > 0 ? D.N : 1));
    W.statsf = c.take<float>(2 * M);
    for (int l = 0; l < D.N; ++l) {
        LayerWs& w = W.lw[l];
        w.xout = c.take<float>(M * D.d); w.x1 = c.take<float>(M * D.d);
        w.stats1 = c.take<float>(2 * M); w.stats2 = c.take<float>(2 * M);
        w.lse = c.take<float>(BH * D.Tp);
        w.xn1 = c.take<bf16>(MP * L.DP); w.xn2 = c.take<bf16>(MP * L.DP);          // MP rows: rows >= M stay zero for wgrad
        w.QR = c.take<bf16>(BH * fragR_elems(D.Tp, L.DKP)); w.KR = c.take<bf16>(BH * fragR_elems(D.Tp, L.DKP));
        w.VR = c.take<bf16>(BH * fragR_elems(D.Tp, L.DKP));
        w.ctx = c.take<bf16>(MP * L.HDP); w.hid = c.take<bf16>(MP * L.FP);
        w.dx2 = c.take<bf16>(MP * L.DP); w.dh = c.take<bf16>(MP * L.FP);
        w.dx1 = c.take<bf16>(MP * L.DP); w.dqkv = c.take<bf16>(MP * L.NQ);
    }
    for (int l = 0; l < D.N; ++l) {        // contiguous [layer][2 norms][G][2][DP] so one launch reduces them all
        W.lw[l].lnpart1 = c.take<float>((size_t)D.GR * 2 * L.DP); W.lw[l].lnpart2 = c.take<float>((size_t)D.GR * 2 * L.DP);
    }
    W.dxa = c.take<float>(M * D.d); W.dxb = c.take<float>(M * D.d);
    W.delta = c.take<float>(BH * D.Tp);
    W.lnpartf = c.take<float>((size_t)D.G * 2 * L.DP);
    W.dOR = c.take<bf16>(BH * fragR_elems(D.Tp, L.DKP));
    const size_t S = D.nsplit;
    {   // one slab set per layer, identical sizes: layer l's set lives at + l * slab_stride floats
        const size_t before = c.off;
        W.sWqkv = c.take<float>(S * L.NQ * L.DP); W.sbqkv = c.take<float>(S * L.NQ);
        W.sWo = c.take<float>(S * L.DP * L.HDP); W.sbo = c.take<float>(S * L.DP);
        W.sW1 = c.take<float>(S * L.FP * L.DP); W.sb1 = c.take<float>(S * L.FP);
        W.sW2 = c.take<float>(S * L.DP * L.FP); W.sb2 = c.take<float>(S * L.DP);
        const size_t set_bytes = c.off - before;
        W.slab_stride = set_bytes / sizeof(float);
        for (int l = 1; l < D.N; ++l) c.take<char>(set_bytes);
    }
    for (int l = 0; l < D.N; ++l) { W.lw[l].maskQ = nullptr; W.lw[l].maskK = nullptr; }
    if (train) {   // attention-dropout bit masks: [layer][bh][tile][tile][32 words], both orientations, written by ONE generator launch
        const size_t lw = attn_mask_layer_words(D.B * D.h, D.nt);
        uint16_t* mq = c.take<uint16_t>(lw * (size_t)(D.N > 0 ? D.N : 1));
        uint16_t* mk = c.take<uint16_t>(lw * (size_t)(D.N > 0 ? D.N : 1));
        for (int l = 0; l < D.N; ++l) { W.lw[l].maskQ = mq ? mq + lw * l : nullptr; W.lw[l].maskK = mk ? mk + lw * l : nullptr; }
    }
    W.bytes = c.off;
}

// ------------------------------------------------------------------------------------ launch helpers
template <int EPI, bool LN, bool WIDE = false, bool DEVSEED = false>
static int launch_rowgemm(const RowGemmParams& p, hipStream_t st, int site = S_OTHER) {
    if (EPI == EPI_LNBWD && !WIDE && p.no_gs) return launch_rowgemm<EPI_LNBWD, false, true>(p, st, site);      // d_model > 128: its own instance
    if (EPI == EPI_PLAIN && !DEVSEED && p.seedword) return launch_rowgemm<EPI_PLAIN, LN, WIDE, true>(p, st, site);   // device-resident seed: its own instance
    if (!WIDE && (p.no_gs || p.kchunk)) return fail(MMT_EINVAL, "K-chunked staging / single-tile column sums exist for the LayerNorm-backward kernel only");
    const size_t lds = rowgemm_lds_bytes(EPI, LN, p.KP, p.NP, p.kchunk, p.no_gs != 0);
    if (lds > 160 * 1024) return fail(MMT_EUNSUPPORTED, "row GEMM tile needs %zu B of LDS (K=%d, N=%d)", lds, p.K, p.N);
    if (WIDE && p.kchunk && (!p.a_bf16 || (p.kchunk & (p.kchunk - 1)) || p.kchunk < 64 || p.A_out))
        return fail(MMT_EINVAL, "K-chunked staging needs a bf16 A matrix, a power-of-two chunk >= 64 and no A copy");
    static size_t configured = 0;           // per instantiation
    if (lds > configured) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&rowgemm_kernel<EPI, LN, WIDE, DEVSEED>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = 160 * 1024;
    }
    const int grid = (p.M + MMT_ROWS - 1) / MMT_ROWS;
    ProfScope prof(site, st);
    hipLaunchKernelGGL((rowgemm_kernel<EPI, LN, WIDE, DEVSEED>), dim3(grid), dim3(MMT_RTHREADS), lds, st, p);
    LAUNCH_CHECK("rowgemm_kernel");
    return MMT_OK;
}

// layer 0's single-stage launches, fixed-shape instances (rowgemm.h): same LDS bytes as the generic kernel computes for these shapes
template <typename K>
static int launch_rowgemm_fixed(K kernel, const RowGemmParams& p, size_t lds, hipStream_t st, int site, const char* name) {
    static const void* configured[4] = {};
    const void* kp = reinterpret_cast<const void*>(kernel);
    bool seen = false;
    for (int i = 0; i < 4; ++i) seen = seen || configured[i] == kp;
    if (!seen) {
        HIP_TRY(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int i = 0; i < 4; ++i) if (!configured[i]) { configured[i] = kp; break; }
    }
    ProfScope prof(site, st);
    hipLaunchKernelGGL(kernel, dim3((p.M + MMT_ROWS - 1) / MMT_ROWS), dim3(MMT_RTHREADS), lds, st, p);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return fail(MMT_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e_));
    return MMT_OK;
}

static int chain_extra_kp(const RowChain3&) { return 0; }
static int chain_extra_kp(const RowChain4& ch) { return ch.d.KP; }

template <typename K, typename CH>
static int launch_rowchain(K kernel, CH& ch, bool lnbwd, int site, const char* name, hipStream_t st, int lnb_np = 0, int shape = 0) {
    // lnbwd: the chain holds LayerNorm-backward stages of output width lnb_np (default: stage b's)
    // LDS geometry shared by the stages
    const int akp = (ch.a.kchunk > 0 && ch.a.kchunk < ch.a.KP) ? ch.a.kchunk : ch.a.KP;     // K-chunked first stage: one chunk in LDS
    int kmax = akp > ch.b.KP ? akp : ch.b.KP;                     // stages reading sm.As: a (global) and whichever of b/c stages via Xs
    if (ch.c.KP > kmax) kmax = ch.c.KP;
    if (chain_extra_kp(ch) > kmax) kmax = chain_extra_kp(ch);
    ch.lda_max = kmax + 8;
    int fw = 128;
    if (lnb_np == 0) lnb_np = ch.b.NP;
    if (lnbwd && lnb_np > fw) fw = lnb_np;                     // LayerNorm-backward epilogue needs the full row
    ch.ldf = fw + 4;
    const bool with_g = lnbwd && !(ch.a.no_gs || ch.b.no_gs || ch.c.no_gs);      // ... and a second fp32 tile unless the column sums recompute x-hat
    const size_t lds = rowchain_lds_bytes(ch, with_g);
    if (lds > 160 * 1024) return fail(MMT_EUNSUPPORTED, "fused row chain needs %zu B of LDS", lds);
    if (shape == 128) {          // the fixed-shape instance carves its LDS from constants (rowgemm.h): they must be this geometry
        const bool ok = ch.ldf == MMT_FIX128_LDF && ch.lda2 == MMT_FIX128_LDA2 && (ch.lda_max == MMT_FIX128_LDA_FWD || ch.lda_max == MMT_FIX128_LDA_BND)
                        && with_g == lnbwd && ch.ldx == (lnbwd ? 0 : 132);
        if (!ok) return fail(MMT_EHIP, "internal: LDS geometry of %s (%d, %d, %d, %d) is not the fixed instance's", name, ch.lda_max, ch.ldf, ch.ldx, ch.lda2);
    }
    if (shape == 256) {
        const bool ok = ch.lda2 == MMT_FIX256_LDA2 && !with_g &&
                        (lnbwd ? (ch.ldf == MMT_FIX256_LDF_BWD && ch.ldx == 0 && (ch.lda_max == MMT_FIX256_LDA || ch.lda_max == MMT_FIX256_LDA_BND))
                               : (ch.ldf == MMT_FIX256_LDF_FWD && ch.ldx == MMT_FIX256_LDX && ch.lda_max == MMT_FIX256_LDA));
        if (!ok) return fail(MMT_EHIP, "internal: LDS geometry of %s (%d, %d, %d, %d) is not the fixed instance's", name, ch.lda_max, ch.ldf, ch.ldx, ch.lda2);
    }
    static const void* configured[16] = {};       // every chain kernel of one argument type shares this instantiation
    const void* kp = reinterpret_cast<const void*>(kernel);
    bool seen = false;
    for (int i = 0; i < 16; ++i) seen = seen || configured[i] == kp;
    if (!seen) {
        HIP_TRY(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        for (int i = 0; i < 16; ++i) if (!configured[i]) { configured[i] = kp; break; }
    }
    ProfScope prof(site, st);
    hipLaunchKernelGGL(kernel, dim3((ch.a.M + MMT_ROWS - 1) / MMT_ROWS), dim3(MMT_RTHREADS), lds, st, ch);
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return fail(MMT_EHIP, "launch of %s failed: %s", name, hipGetErrorString(e_));
    return MMT_OK;
}

// Fixed-shape instances of the chain kernels (rowgemm.h): 128 = d_model = d_ff = h d_k = 128 with 8 heads of 16 (configs[3]); 0 = generic
static int chain_shape(int d, int f, int h, const LayerLayout& L) {
    static const bool off = getenv("MMT_NO_FIXED_SHAPES") != nullptr;
    if (off) return 0;
    if (d == 128 && L.DP == 128 && f == 128 && L.FP == 128 && L.HDP == 128 && L.NQ == 384 && h == 8 && L.DKP == 16) return 128;
    if (d == 256 && L.DP == 256 && f == 128 && L.FP == 128 && L.HDP == 256 && L.NQ == 768 && h == 8 && L.DKP == 32) return 256;   // MFT stacks
    return 0;
}
static RowGemmParams rg_zero() { RowGemmParams p; memset(&p, 0, sizeof(p)); p.mask_scale = 1.0f; return p; }

static DropCfg no_drop() { return make_drop(0.f, 0, 0); }
// Dropout configuration of one stream: keys mixed on the host from a by-value seed, or — device-resident seed (`devseed`) — the stream id
// in s0 for the kernels to resolve against the workspace's seed word (common.h drop_resolve)
#define MMT_ATTN_DROP_BITS 12         // resolution of the attention-probability dropout (common.h make_drop)
// `first`: the stream whose keys sit in slot 0 of the workspace's seed block (seed_advance_kernel)
static DropCfg stream_drop(float p, uint64_t seed, uint32_t stream, bool devseed, int bits = 16, uint32_t first = 0) {
    DropCfg c = make_drop(p, seed, stream, bits);
    if (devseed) { c.s0 = stream - first; c.s1 = 0; }
    return c;
}
static int launch_seed_advance(uint64_t* state, uint64_t* block, uint32_t first, int n, hipStream_t st) {
    if (n + 1 > MMT_SEED_BLOCK_WORDS) return fail(MMT_EUNSUPPORTED, "%d dropout streams do not fit the seed block", n);
    hipLaunchKernelGGL(seed_advance_kernel, dim3(1), dim3(128), 0, st, state, block, first, n);
    LAUNCH_CHECK("seed_advance_kernel");
    return MMT_OK;
}

// Attention-probability dropout of `nlayers` layers: ONE launch draws every decision (stream 4l+0 of layer l) into the lane-mask
// arrays mq / mk (layer l at + l * attn_mask_layer_words).  attn_mask.h.
static int fill_mask_gen(MaskGenParams& P, uint16_t* mq, uint16_t* mk, const EncDims& D, int nlayers, float p, uint64_t seed,
                         const uint64_t* seedword = nullptr) {
    if (nlayers > 16) return fail(MMT_EUNSUPPORTED, "mask generator: %d layers > 16", nlayers);
    memset(&P, 0, sizeof(P));
    P.lq = mq; P.lk = mk; P.nbh = D.B * D.h; P.nt = D.nt; P.nlayers = nlayers;
    P.layer_words = attn_mask_layer_words(P.nbh, P.nt);
    P.seedword = seedword;
    for (int l = 0; l < nlayers; ++l) { const DropCfg c = stream_drop(p, seed, 4 * l + 0, seedword != nullptr, MMT_ATTN_DROP_BITS); P.thr16 = c.thr16; P.s0[l] = c.s0; P.s1[l] = c.s1; }
    return MMT_OK;
}
static int launch_mask_gen(uint16_t* mq, uint16_t* mk, const EncDims& D, int nlayers, float p, uint64_t seed, hipStream_t st) {
    MaskGenParams P;
    int rc;
    if ((rc = fill_mask_gen(P, mq, mk, D, nlayers, p, seed))) return rc;
    const size_t blocks = (size_t)P.nbh * P.nt * P.nt;
    ProfScope prof(S_MASK_GEN, st);
    hipLaunchKernelGGL(attn_mask_gen_kernel, dim3((unsigned)((blocks + 255) / 256), nlayers), dim3(256), 0, st, P);
    LAUNCH_CHECK("attn_mask_gen_kernel");
    return MMT_OK;
}

// `drop`: the layer's attention dropout (thr16 == 0: off); maskQ: its lane words, written by launch_mask_gen
static int launch_attn_fwd(int DKP, const bf16* QR, const bf16* KR, const bf16* VR, bf16* ctx, float* lse,
                           const EncDims& D, hipStream_t st, DropCfg drop = no_drop(), const uint16_t* maskQ = nullptr) {
    dim3 grid(attn_grid((D.nt + 3) / 4, D.B * D.h));
    if (drop.thr16 && !maskQ) return fail(MMT_EINVAL, "attention dropout without a mask buffer");
    ProfScope prof(S_ATTN_FWD, st);
#define MMT_FWD(dkp, dr) for (int fb = 0; fb < (dkp + 31) / 32; ++fb) \
        hipLaunchKernelGGL((attn_fwd_kernel<dkp, dr>), grid, dim3(MMT_THREADS), 0, st, QR, KR, VR, ctx, lse, \
                           D.h, D.T, D.nt, D.B * D.h, D.L.HDP, maskQ, drop.scale, fb)
#ifdef MMT_ABLATIONS
    static const int abl = getenv("MMT_ABL") ? atoi(getenv("MMT_ABL")) : 0;
#define MMT_FWD_A(a) hipLaunchKernelGGL((attn_fwd_kernel<16, true, a>), grid, dim3(MMT_THREADS), 0, st, QR, KR, VR, ctx, lse, \
                                        D.h, D.T, D.nt, D.B * D.h, D.L.HDP, maskQ, drop.scale, 0)
    if (abl && DKP == 16 && drop.thr16) {
        switch (abl) { case 1: MMT_FWD_A(1); break; case 2: MMT_FWD_A(2); break; case 3: MMT_FWD_A(3); break;
                       case 5: MMT_FWD_A(5); break; default: MMT_FWD_A(6); }
        LAUNCH_CHECK("attn_fwd_kernel"); return MMT_OK;
    }
#endif
    if (DKP == 16) { if (drop.thr16) MMT_FWD(16, true); else MMT_FWD(16, false); }
    else if (DKP == 32) { if (drop.thr16) MMT_FWD(32, true); else MMT_FWD(32, false); }
    else { if (drop.thr16) MMT_FWD(64, true); else MMT_FWD(64, false); }
#undef MMT_FWD
    LAUNCH_CHECK("attn_fwd_kernel");
    return MMT_OK;
}

// dQ, dK, dV -> bf16 row-major dqkv [M][NQ] (columns: dQ | dK | dV, heads padded)
static int launch_attn_bwd(int DKP, const bf16* QR, const bf16* KR, const bf16* VR,
                           const bf16* dOR, const float* lse, const float* delta, const float* rowmask,
                           bf16* dqkv, const EncDims& D, hipStream_t st, DropCfg drop = no_drop(),
                           const uint16_t* maskQ = nullptr, const uint16_t* maskK = nullptr) {
    dim3 grid(attn_grid((D.nt + 3) / 4, D.B * D.h));
    const float scale = 1.0f / sqrtf((float)D.L.dk);
    if (drop.thr16 && (!maskQ || !maskK)) return fail(MMT_EINVAL, "attention dropout without mask buffers");
    if (use_fused_bwd(D)) {                             // one evaluation of P and dS per score, diagonal sweep: attn_bwd_diag.h
        static bool configured = false;
        if (!configured) {
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_diag16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_diag16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            configured = true;
        }
        ProfScope prof(S_ATTN_BWD_FUSED, st);
#define MMT_DIAG(dr) hipLaunchKernelGGL((attn_bwd_diag16_kernel<dr>), dim3(D.B * D.h), dim3(MMT_DIAG_THREADS), MMT_DIAG_LDS_BYTES, st, \
                                         QR, KR, VR, dOR, lse, delta, rowmask, scale, dqkv, D.L.NQ, D.h, D.T, D.nt, maskK, drop.scale)
#ifdef MMT_ABLATIONS
        static const bool stamp = getenv("MMT_ABL") && atoi(getenv("MMT_ABL")) == 7;
        if (stamp && drop.thr16) {
            static bool c2 = false;
            if (!c2) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_diag16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); c2 = true; }
            hipLaunchKernelGGL((attn_bwd_diag16_kernel<true, true>), dim3(D.B * D.h), dim3(MMT_DIAG_THREADS), MMT_DIAG_LDS_BYTES, st,
                               QR, KR, VR, dOR, lse, delta, rowmask, scale, dqkv, D.L.NQ, D.h, D.T, D.nt, maskK, drop.scale);
            LAUNCH_CHECK("attn_bwd_diag16_kernel");
            return MMT_OK;
        }
        static const int babl = getenv("MMT_BABL") ? atoi(getenv("MMT_BABL")) : 0;      // timing-only ablation mask (attn_bwd_diag.h)
        if (babl && drop.thr16) {
#define MMT_DIAG_A(a) case a: { static bool c3 = false; \
            if (!c3) { HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_diag16_kernel<true, false, a>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); c3 = true; } \
            hipLaunchKernelGGL((attn_bwd_diag16_kernel<true, false, a>), dim3(D.B * D.h), dim3(MMT_DIAG_THREADS), MMT_DIAG_LDS_BYTES, st, \
                               QR, KR, VR, dOR, lse, delta, rowmask, scale, dqkv, D.L.NQ, D.h, D.T, D.nt, maskK, drop.scale); } break;
            switch (babl) {
                MMT_DIAG_A(1) MMT_DIAG_A(2) MMT_DIAG_A(3) MMT_DIAG_A(4) MMT_DIAG_A(8) MMT_DIAG_A(16) MMT_DIAG_A(32) MMT_DIAG_A(64) MMT_DIAG_A(128)
                MMT_DIAG_A(192) MMT_DIAG_A(7) MMT_DIAG_A(24) MMT_DIAG_A(28) MMT_DIAG_A(60) MMT_DIAG_A(63) MMT_DIAG_A(255) MMT_DIAG_A(56) MMT_DIAG_A(59)
                default: return fail(MMT_EINVAL, "MMT_BABL: no such ablation instance");
            }
#undef MMT_DIAG_A
            LAUNCH_CHECK("attn_bwd_diag16_kernel");
            return MMT_OK;
        }
#endif
        if (drop.thr16) MMT_DIAG(true); else MMT_DIAG(false);
#undef MMT_DIAG
        LAUNCH_CHECK("attn_bwd_diag16_kernel");
        return MMT_OK;
    }
    {
        ProfScope prof(S_ATTN_BWD, st);
#define MMT_DKV(dkp, dr) for (int fb = 0; fb < (dkp + 31) / 32; ++fb) \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<dkp, dr>), grid, dim3(MMT_THREADS), 0, st, QR, KR, VR, dOR, lse, delta, \
                           dqkv, D.L.NQ, D.h, D.T, D.nt, D.B * D.h, maskK, drop.scale, fb)
        if (DKP == 16) { if (drop.thr16) MMT_DKV(16, true); else MMT_DKV(16, false); }
        else if (DKP == 32) { if (drop.thr16) MMT_DKV(32, true); else MMT_DKV(32, false); }
        else { if (drop.thr16) MMT_DKV(64, true); else MMT_DKV(64, false); }
#undef MMT_DKV
    }
    LAUNCH_CHECK("attn_bwd_dkv_kernel");
    {
        ProfScope prof(S_ATTN_BWD_DQ, st);
#define MMT_DQ(dkp, dr) for (int fb = 0; fb < (dkp + 31) / 32; ++fb) \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<dkp, dr>), grid, dim3(MMT_THREADS), 0, st, QR, KR, VR, dOR, lse, delta, rowmask, \
                           scale, dqkv, D.L.NQ, D.h, D.T, D.nt, D.B * D.h, maskQ, drop.scale, fb)
        if (DKP == 16) { if (drop.thr16) MMT_DQ(16, true); else MMT_DQ(16, false); }
        else if (DKP == 32) { if (drop.thr16) MMT_DQ(32, true); else MMT_DQ(32, false); }
        else { if (drop.thr16) MMT_DQ(64, true); else MMT_DQ(64, false); }
#undef MMT_DQ
    }
    LAUNCH_CHECK("attn_bwd_dq_kernel");
    return MMT_OK;
}

static int grid_for(size_t n, int block = 256) {
    size_t g = (n + block - 1) / block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

// ------------------------------------------------------------------------------------ encoder stack
extern "C" size_t mmt_encoder_param_count(int d, int f, int n_layers) {
    LayerLayout L = make_layout(d, f, 1);
    return L.stride() * (size_t)n_layers + 2 * (size_t)d;
}

extern "C" size_t mmt_encoder_workspace_bytes(int B, int T, int d, int h, int f, int n_layers) {
    EncDims D;
    if (make_dims(D, B, T, d, h, f, n_layers) != MMT_OK) return 0;
    EncWs W; carve_encoder(W, D, nullptr, true);
    return W.bytes;
}
extern "C" size_t mmt_encoder_workspace_bytes_eval(int B, int T, int d, int h, int f, int n_layers) {
    EncDims D;
    if (make_dims(D, B, T, d, h, f, n_layers) != MMT_OK) return 0;
    EncWs W; carve_encoder(W, D, nullptr, false);
    return W.bytes;
}

static const float LOG2E = 1.4426950408889634f;

// seed_state != nullptr: device-resident seed (see common.h): the seed is read from, and advanced in, device memory by the first kernel
static int encoder_forward_impl(const float* x, const float* mask, const float* params, float* y,
                                void* workspace, size_t workspace_bytes,
                                int B, int T, int d, int h, int f, int n_layers, float eps,
                                float dropout_p, uint64_t seed, uint64_t* seed_state, mmt_stream_t stream) {
    EncDims D;
    int rc = make_dims(D, B, T, d, h, f, n_layers);
    if (rc) return rc;
    if (!x || !mask || !params || !y || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(MMT_EINVAL, "dropout_p %g not in [0,1)", dropout_p);
    if (dropout_p > 0.f && D.Tp > 4096) return fail(MMT_EUNSUPPORTED, "train-mode dropout supports T <= 4096 (got %d)", T);
    EncWs W; carve_encoder(W, D, workspace, dropout_p > 0.f);
    if (workspace_bytes < W.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, W.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const LayerLayout& L = D.L;
    const bool devseed = seed_state != nullptr && dropout_p > 0.f;
    const uint64_t* seedword = devseed ? W.seedword : nullptr;
    if (devseed && (rc = launch_seed_advance(seed_state, W.seedword, 0, 4 * D.N + 4, st))) return rc;
    auto mkdrop = [&](int stream_id) { return stream_drop(dropout_p, seed, (uint32_t)stream_id, devseed, (stream_id & 3) == 0 ? MMT_ATTN_DROP_BITS : 16); };

    if (D.N > 0 && dropout_p > 0.f) {
        // weight preparation + every attention-dropout decision of this forward pass (and of its backward), all layers: one launch
        MaskGenParams P;
        if ((rc = fill_mask_gen(P, W.lw[0].maskQ, W.lw[0].maskK, D, D.N, dropout_p, seed, seedword))) return rc;
        const size_t blocks = (size_t)P.nbh * P.nt * P.nt;
        const int gen_blocks = (int)((blocks + 255) / 256), prep_blocks = std::min(grid_for(L.pstride() + L.qstride()), 64);
        ProfScope prof(S_MASK_GEN, st);
        hipLaunchKernelGGL(encoder_prep_maskgen_kernel, dim3((unsigned)(gen_blocks + prep_blocks), D.N), dim3(256), 0, st,
                           params, W.wprep, W.bprep, L, gen_blocks, P);
        LAUNCH_CHECK("encoder_prep_maskgen_kernel");
    } else if (D.N > 0) {
        ProfScope prof(S_PREP, st);
        hipLaunchKernelGGL(encoder_prep_kernel, dim3(grid_for(L.pstride() + L.qstride()), D.N), dim3(256), 0, st,
                           params, W.wprep, W.bprep, L);
        LAUNCH_CHECK("encoder_prep_kernel");
    }
    const float* xin = x;
    const int shape = chain_shape(d, f, h, L);
    static const bool fuse_next_qkv = getenv("MMT_NO_CHAIN4") == nullptr;
    for (int l = 0; l < D.N; ++l) {
        const LayerWs& w = W.lw[l];
        const float* P = params + (size_t)l * L.stride();
        const bf16* wp = W.wprep + (size_t)l * L.pstride();
        const float* bp = W.bprep + (size_t)l * L.qstride();
        // LayerNorm 1 + fused Q/K/V projection of layer `ll` -> attention operand fragments.  Layer 0 runs it as its own
        // kernel; for the later layers it is the fourth stage of the previous layer's post-attention chain.
        auto qkv_params = [&](int ll, const float* src) {
            const LayerWs& wl = W.lw[ll];
            const float* Pl = params + (size_t)ll * L.stride();
            RowGemmParams p = rg_zero();
            p.M = D.M; p.K = d; p.KP = L.DP; p.N = 3 * L.HD; p.NP = L.NQ;
            p.A = src; p.lda = d; p.A_out = wl.xn1; p.lda_out = L.DP;
            p.ln_a = Pl + L.oln(0); p.ln_b = Pl + L.oln(1); p.eps = eps; p.stats = wl.stats1;
            p.W = W.wprep + (size_t)ll * L.pstride() + L.pWqkv(); p.bias = W.bprep + (size_t)ll * L.qstride() + L.qbqkv();
            p.fragR[0] = wl.QR; p.fragR[1] = wl.KR; p.fragR[2] = wl.VR;
            p.T = T; p.Tp = D.Tp; p.h = h; p.DKP = L.DKP; p.nwhich = 3;
            p.rowmask = mask; p.qscale = LOG2E / sqrtf((float)L.dk); p.scale_first = 1;
            return p;
        };
        if (l == 0) {
            RowGemmParams p = qkv_params(0, xin);
            if (shape == 128) rc = launch_rowgemm_fixed(encoder_ln1_qkv128_kernel, p, rowgemm_lds_bytes(EPI_FRAG, true, 128, 384), st, S_LN1_QKV, "encoder_ln1_qkv128_kernel");
            else rc = launch_rowgemm<EPI_FRAG, true>(p, st, S_LN1_QKV);
            if (rc) return rc;
        }
        // dropout streams of layer l: 4l+0 attention probabilities (:33), 4l+1 / 4l+3 sublayer outputs (:104), 4l+2 FFN hidden (:20)
        if ((rc = launch_attn_fwd(L.DKP, w.QR, w.KR, w.VR, w.ctx, w.lse, D, st, mkdrop(4 * l + 0), w.maskQ))) return rc;
        {   // out-proj + residual -> LN2 + FFN1 + ReLU -> FFN2 + residual, one kernel, x1 and hid stay in LDS
            RowChain3 ch; memset(&ch, 0, sizeof(ch));
            {   RowGemmParams& p = ch.a; p = rg_zero();
                p.M = D.M; p.K = L.HDP; p.KP = L.HDP; p.N = d; p.NP = L.DP;
                p.A = w.ctx; p.a_bf16 = 1; p.lda = L.HDP;
                p.W = wp + L.pWo(); p.bias = bp + L.qbo();
                p.residual = xin; p.ldr = d; p.out_f32 = w.x1; p.ldo = d;
                p.drop = mkdrop(4 * l + 1); p.seedword = seedword; }
            {   RowGemmParams& p = ch.b; p = rg_zero();
                p.M = D.M; p.K = d; p.KP = L.DP; p.N = f; p.NP = L.FP;
                p.A_out = w.xn2; p.lda_out = L.DP;
                p.ln_a = P + L.oln(2); p.ln_b = P + L.oln(3); p.eps = eps; p.stats = w.stats2;
                p.W = wp + L.pW1(); p.bias = bp + L.qb1(); p.act = 1;
                p.out_bf16 = w.hid; p.ldo16 = L.FP; p.n_store16 = L.FP;
                p.drop = mkdrop(4 * l + 2); p.seedword = seedword; }
            {   RowGemmParams& p = ch.c; p = rg_zero();
                p.M = D.M; p.K = L.FP; p.KP = L.FP; p.N = d; p.NP = L.DP;
                p.W = wp + L.pW2(); p.bias = bp + L.qb2();
                p.out_f32 = w.xout; p.ldo = d;
                p.drop = mkdrop(4 * l + 3); p.seedword = seedword; }
            ch.ldx = L.DP + 4; ch.lda2 = L.FP + 8;
            if (l + 1 < D.N && fuse_next_qkv) {
                RowChain4 c4; memset(&c4, 0, sizeof(c4));
                c4.a = ch.a; c4.b = ch.b; c4.c = ch.c; c4.ldx = ch.ldx; c4.lda2 = ch.lda2;
                c4.d = qkv_params(l + 1, nullptr);                 // A operand: the x2 tile in LDS
                auto k4 = shape == 128 ? (devseed ? encoder_post_attn_fwd4_kernel<true, 128> : encoder_post_attn_fwd4_kernel<false, 128>)
                        : shape == 256 ? (devseed ? encoder_post_attn_fwd4_kernel<true, 256> : encoder_post_attn_fwd4_kernel<false, 256>)
                                       : (devseed ? encoder_post_attn_fwd4_kernel<true, 0> : encoder_post_attn_fwd4_kernel<false, 0>);
                if ((rc = launch_rowchain(k4, c4, false, S_CHAIN4_FWD, "encoder_post_attn_fwd4_kernel", st, 0, shape))) return rc;
            } else {
                if (l + 1 == D.N) {        // last layer: the stack's final LayerNorm runs on the output tile while it is in LDS
                    const float* Pf = params + (size_t)D.N * L.stride();
                    ch.ln.a = Pf; ch.ln.b = Pf + d; ch.ln.eps = eps; ch.ln.y = y; ch.ln.stats = W.statsf; ch.ln.d = d;
                }
                auto k3 = shape == 128 ? (devseed ? encoder_post_attn_fwd_kernel<true, 128> : encoder_post_attn_fwd_kernel<false, 128>)
                        : shape == 256 ? (devseed ? encoder_post_attn_fwd_kernel<true, 256> : encoder_post_attn_fwd_kernel<false, 256>)
                                       : (devseed ? encoder_post_attn_fwd_kernel<true, 0> : encoder_post_attn_fwd_kernel<false, 0>);
                if ((rc = launch_rowchain(k3, ch, false, S_OUTPROJ, "encoder_post_attn_fwd_kernel", st, 0, shape))) return rc;
                if (l + 1 < D.N) {
                    RowGemmParams p = qkv_params(l + 1, w.xout);
                    if ((rc = launch_rowgemm<EPI_FRAG, true>(p, st, S_LN1_QKV))) return rc;
                }
            }
        }
        xin = w.xout;
    }
    if (D.N == 0) {                    // no layers: the final LayerNorm alone (otherwise it is the tail of the last layer's chain)
        const float* Pf = params;
        ProfScope prof(S_LN_FWD, st);
        hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(D.G), dim3(MMT_THREADS), 0, st, xin, Pf, Pf + d, eps, y, W.statsf, D.M, d);
        LAUNCH_CHECK("layernorm_fwd_kernel");
    }
    return MMT_OK;
}

extern "C" int mmt_encoder_forward(const float* x, const float* mask, const float* params, float* y,
                                   void* workspace, size_t workspace_bytes,
                                   int B, int T, int d, int h, int f, int n_layers, float eps,
                                   float dropout_p, uint64_t seed, mmt_stream_t stream) {
    return encoder_forward_impl(x, mask, params, y, workspace, workspace_bytes, B, T, d, h, f, n_layers, eps, dropout_p, seed, nullptr, stream);
}
extern "C" int mmt_encoder_forward_devseed(const float* x, const float* mask, const float* params, float* y,
                                           void* workspace, size_t workspace_bytes,
                                           int B, int T, int d, int h, int f, int n_layers, float eps,
                                           float dropout_p, uint64_t* seed_state, mmt_stream_t stream) {
    if (!seed_state) return fail(MMT_EINVAL, "null seed state");
    return encoder_forward_impl(x, mask, params, y, workspace, workspace_bytes, B, T, d, h, f, n_layers, eps, dropout_p, 0, seed_state, stream);
}

static int launch_ln_bwd(const float* dy, const float* x, const float* a, const float* stats, float eps, float* dx,
                         float* colpart, int M, int d, int DP, hipStream_t st) {
    const size_t lds = (size_t)2 * 32 * (DP + 4) * 4;
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&layernorm_bwd_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    if (lds > 160 * 1024) return fail(MMT_EUNSUPPORTED, "LayerNorm width %d too large", d);
    ProfScope prof(S_LN_BWD, st);
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((M + 31) / 32), dim3(MMT_THREADS), lds, st, dy, x, a, stats, eps, dx, colpart, M, d, DP);
    LAUNCH_CHECK("layernorm_bwd_kernel");
    return MMT_OK;
}

// devseed: the forward was device-seeded; its seed sits in the workspace's seed word
static int encoder_backward_impl(const float* dy, const float* x, const float* mask, const float* params,
                                 float* dx, float* dparams,
                                 void* workspace, size_t workspace_bytes,
                                 int B, int T, int d, int h, int f, int n_layers, float eps,
                                 float dropout_p, uint64_t seed, bool devseed_in, mmt_stream_t stream) {
    EncDims D;
    int rc = make_dims(D, B, T, d, h, f, n_layers);
    if (rc) return rc;
    if (!dy || !x || !mask || !params || !dx || !dparams || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(MMT_EINVAL, "dropout_p %g not in [0,1)", dropout_p);
    EncWs W; carve_encoder(W, D, workspace, dropout_p > 0.f);
    if (workspace_bytes < W.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, W.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const LayerLayout& L = D.L;
    const bool devseed = devseed_in && dropout_p > 0.f;
    const uint64_t* seedword = devseed ? W.seedword : nullptr;
    auto mkdrop = [&](int stream_id) { return stream_drop(dropout_p, seed, (uint32_t)stream_id, devseed, (stream_id & 3) == 0 ? MMT_ATTN_DROP_BITS : 16); };

    // final LayerNorm
    const float* Pf = params + (size_t)D.N * L.stride();
    float* gPf = dparams + (size_t)D.N * L.stride();
    const float* x_last = (D.N > 0) ? W.lw[D.N - 1].xout : x;
    float* cur = (D.N > 0) ? W.dxa : dx;
    if ((rc = launch_ln_bwd(dy, x_last, Pf, W.statsf, eps, cur, W.lnpartf, D.M, d, L.DP, st))) return rc;
    // (the LayerNorm parameter gradients of the whole stack are reduced by the finalize launch at the end)

    WgradJobs J; memset(&J, 0, sizeof(J));
    J.MP = D.MP; J.M16 = D.M16; J.mchunk = D.mchunk;
    int t0 = 0;
    auto add_job = [&](const bf16* A, int lda, const bf16* B_, int ldb, float* out, float* bout, int NPj, int KPj) {
        WgradJob& j = J.j[J.njobs++];
        j.A = A; j.lda = lda; j.B = B_; j.ldb = ldb; j.out = out; j.bias_out = bout;
        j.NPj = NPj; j.KPj = KPj; j.tile0 = t0; j.tiles_k = KPj / 64;
        t0 += (NPj / 64) * (KPj / 64);
    };

    float* other = W.dxb;
    // dx2 -> dh -> dx1 -> dO fragments of layer l (reads `cur` = dx2, writes `other` = dx1); dh and drop'(dx1) stay in LDS
    auto build_chain = [&](int l, RowChain3& ch) {
        const LayerWs& w = W.lw[l];
        const float* P = params + (size_t)l * L.stride();
        const bf16* wp = W.wprep + (size_t)l * L.pstride();
        memset(&ch, 0, sizeof(ch));
        {   RowGemmParams& p = ch.a; p = rg_zero();           // dh = (drop'(dx2) W2) * relu'(hid)   [emits dx2, dh as bf16 rows]
            p.M = D.M; p.K = d; p.KP = L.DP; p.N = f; p.NP = L.FP;
            p.A = cur; p.lda = d; p.A_out = w.dx2; p.lda_out = L.DP;
            p.W = wp + L.pW2T();
            p.relu_mask = w.hid; p.ldm = L.FP;          // hid > 0  <=>  ReLU passed AND the unit was kept
            p.mask_scale = make_drop(dropout_p, seed, 4 * l + 2).scale;
            p.a_drop = mkdrop(4 * l + 3); p.seedword = seedword;   // gradient of the dropped sublayer-1 output
            p.out_bf16 = w.dh; p.ldo16 = L.FP; p.n_store16 = L.FP; }
        {   RowGemmParams& p = ch.b; p = rg_zero();           // dx1 = dx2 + LN2bwd(dh W1)
            p.M = D.M; p.K = L.FP; p.KP = L.FP; p.N = d; p.NP = L.DP;
            p.W = wp + L.pW1T();
            p.x = w.x1; p.ldx = d; p.st = w.stats2; p.ln_a = P + L.oln(2); p.eps = eps; p.d_real = d;
            p.dres = cur; p.lddres = d; p.out_f32 = other; p.ldo = d; p.colpart = w.lnpart2;
            p.no_gs = L.DP > 128;                               // d_model > 128: a second fp32 tile would leave one workgroup per CU
            p.next_drop = mkdrop(4 * l + 1); p.seedword = seedword;   // gradient of the dropped sublayer-0 output, applied to the next A tile
            p.next_lda = L.DP + 8; }
        {   RowGemmParams& p = ch.c; p = rg_zero();           // dO = drop'(dx1) Wo -> fragments + delta   [emits dx1 as bf16 rows]
            p.M = D.M; p.K = d; p.KP = L.DP; p.N = L.HD; p.NP = L.HDP;
            p.A_out = w.dx1; p.lda_out = L.DP;                 // (its A tile, drop'(dx1) in bf16, was left in LDS by the stage before)
            p.W = wp + L.pWoT();
            p.fragR[0] = W.dOR;
            p.T = T; p.Tp = D.Tp; p.h = h; p.DKP = L.DKP; p.nwhich = 1;
            p.ctx = w.ctx; p.ldctx = L.HDP; p.delta = W.delta; }
        ch.ldx = 0; ch.lda2 = L.FP + 8;                        // no fp32 tile kept between the stages
    };
    // dx = dx1 + LN1bwd(dQKV Wqkv) of layer l (reads `other` = dx1, writes `out`)
    auto build_qkv = [&](int l, float* out) {
        const LayerWs& w = W.lw[l];
        const float* P = params + (size_t)l * L.stride();
        const bf16* wp = W.wprep + (size_t)l * L.pstride();
        RowGemmParams p = rg_zero();
        p.M = D.M; p.K = L.NQ; p.KP = L.NQ; p.N = d; p.NP = L.DP;
        p.A = w.dqkv; p.a_bf16 = 1; p.lda = L.NQ;
        p.W = wp + L.pWqkvT();
        p.x = (l > 0) ? W.lw[l - 1].xout : x; p.ldx = d; p.st = w.stats1; p.ln_a = P + L.oln(0); p.eps = eps; p.d_real = d;
        p.dres = other; p.lddres = d; p.out_f32 = out; p.ldo = d; p.colpart = w.lnpart1;
        p.kchunk = (L.NQ > 512) ? 512 : 0;                     // d_model = 256: K = 768; half of the A tile in LDS at a time
        p.no_gs = (L.DP > 128) || p.kchunk;                    // (both live in the WIDE instance of the kernel)
        return p;
    };
    // Layers l >= 1 close (bwd_qkv + LayerNorm-1 backward) inside the kernel that opens layer l-1 (encoder_bwd_boundary_kernel) unless
    // MMT_NO_BWD_BOUNDARY=1
    const int shape = chain_shape(d, f, h, L);
    static const bool fuse_boundary = getenv("MMT_NO_BWD_BOUNDARY") == nullptr;
    const bool boundary = fuse_boundary;
    for (int l = D.N - 1; l >= 0; --l) {
        const LayerWs& w = W.lw[l];
        if (l == D.N - 1 || !boundary) {
            RowChain3 ch; build_chain(l, ch);
            const int sh = ch.b.no_gs ? (shape == 256 ? 256 : 0) : (shape == 128 ? 128 : 0);
            auto k3 = sh == 256 ? (devseed ? encoder_pre_attn_bwd_kernel<true, true, 256> : encoder_pre_attn_bwd_kernel<true, false, 256>)
                    : ch.b.no_gs ? (devseed ? encoder_pre_attn_bwd_kernel<true, true> : encoder_pre_attn_bwd_kernel<true, false>)
                    : sh == 128 ? (devseed ? encoder_pre_attn_bwd_kernel<false, true, 128> : encoder_pre_attn_bwd_kernel<false, false, 128>)
                                : (devseed ? encoder_pre_attn_bwd_kernel<false, true> : encoder_pre_attn_bwd_kernel<false, false>);
            rc = launch_rowchain(k3, ch, true, S_BWD_FFN2, "encoder_pre_attn_bwd_kernel", st, 0, sh);
            if (rc) return rc;
        }
        if ((rc = launch_attn_bwd(L.DKP, w.QR, w.KR, w.VR, W.dOR, w.lse, W.delta, mask,
                                  w.dqkv, D, st, mkdrop(4 * l + 0), w.maskQ, w.maskK))) return rc;
        float* dxin = (l > 0) ? cur : dx;
        if (l > 0 && boundary) {
            RowChain3 below; build_chain(l - 1, below);        // (its stages read `cur` = the dx this kernel's first stage writes)
            RowChain4 c4; memset(&c4, 0, sizeof(c4));
            c4.a = build_qkv(l, dxin);
            c4.a.next_drop = below.a.a_drop; c4.a.next_lda = below.a.KP + 8;      // next A tile = bf16(drop'(dx)), left in LDS
            c4.a.seedword = seedword;
            c4.b = below.a; c4.c = below.b; c4.d = below.c; c4.ldx = 0; c4.lda2 = below.lda2;
            if (c4.a.no_gs != c4.c.no_gs) { c4.a.no_gs = c4.c.no_gs = 1; }        // (one WIDE flag per kernel: K-chunking alone implies it)
            const int sh = (c4.a.no_gs || c4.a.kchunk) ? ((shape == 256 && c4.a.no_gs && c4.a.kchunk == 512) ? 256 : 0) : (shape == 128 ? 128 : 0);
            auto k4 = sh == 256 ? (devseed ? encoder_bwd_boundary_kernel<true, true, 256> : encoder_bwd_boundary_kernel<true, false, 256>)
                    : c4.a.no_gs ? (devseed ? encoder_bwd_boundary_kernel<true, true> : encoder_bwd_boundary_kernel<true, false>)
                    : sh == 128 ? (devseed ? encoder_bwd_boundary_kernel<false, true, 128> : encoder_bwd_boundary_kernel<false, false, 128>)
                                : (devseed ? encoder_bwd_boundary_kernel<false, true> : encoder_bwd_boundary_kernel<false, false>);
            rc = launch_rowchain(k4, c4, true, S_BWD_BOUNDARY, "encoder_bwd_boundary_kernel", st, L.DP, sh);
            if (rc) return rc;
        } else {
            RowGemmParams p = build_qkv(l, dxin);
            if (shape == 128 && !p.no_gs && !p.kchunk) {
                p.seedword = seedword;
                rc = launch_rowgemm_fixed(devseed ? encoder_bwd_qkv_ln1_128_kernel<true> : encoder_bwd_qkv_ln1_128_kernel<false>, p,
                                          rowgemm_lds_bytes(EPI_LNBWD, false, 384, 128), st, S_BWD_QKV_LN1, "encoder_bwd_qkv_ln1_128_kernel");
            } else rc = launch_rowgemm<EPI_LNBWD, false>(p, st, S_BWD_QKV_LN1);
            if (rc) return rc;
        }
        // weight-gradient jobs of this layer (run later, all layers in one launch)
        const size_t so = (size_t)l * W.slab_stride;
        add_job(w.dqkv, L.NQ, w.xn1, L.DP, W.sWqkv + so, W.sbqkv + so, L.NQ, L.DP);
        add_job(w.dx1, L.DP, w.ctx, L.HDP, W.sWo + so, W.sbo + so, L.DP, L.HDP);
        add_job(w.dh, L.FP, w.xn2, L.DP, W.sW1 + so, W.sb1 + so, L.FP, L.DP);
        add_job(w.dx2, L.DP, w.hid, L.FP, W.sW2 + so, W.sb2 + so, L.DP, L.FP);
        // cur now holds dx of this layer (= dx2 of the layer below); `other` is free again
    }
    if (D.N > 0) {
        {   // every layer's weight and bias gradients: one launch
            ProfScope prof(S_WGRAD, st);
            J.tiles_per_layer = t0 / D.N; J.nlayers = D.N; J.nsplit = D.nsplit;         // XCD-aware 1-D grid (wgrad_kernel)
            const int units = D.N * D.nsplit;
            hipLaunchKernelGGL(wgrad_kernel, dim3(8 * ((units + 7) / 8) * J.tiles_per_layer), dim3(MMT_THREADS), 0, st, J);
        }
        LAUNCH_CHECK("wgrad_kernel");
        ProfScope prof(S_FINALIZE, st);
        LayerSlabs S;
        S.dWqkv = W.sWqkv; S.dbqkv = W.sbqkv; S.dWo = W.sWo; S.dbo = W.sbo;
        S.dW1 = W.sW1; S.db1 = W.sb1; S.dW2 = W.sW2; S.db2 = W.sb2;
        S.ln1part = nullptr; S.ln2part = nullptr; S.nsplit = D.nsplit; S.G = D.G; S.slab_stride = W.slab_stride;
        // LayerNorm jobs: the final norm (32-row partials of layernorm_bwd_kernel) and the 2N sublayer norms (row-tile partials of the
        // LayerNorm-backward epilogues); outputs (a_2, b_2) are adjacent in the flat gradient
        LnJobs LJ; memset(&LJ, 0, sizeof(LJ));
        LJ.DP = L.DP; LJ.d = d; LJ.lnblocks = 2 * ((d + 31) / 32);
        LJ.wblocks = (int)std::min<size_t>(64, (L.oln(0) + 4095) / 4096);
        LJ.part[0] = W.lnpartf; LJ.out_a[0] = gPf; LJ.G[0] = D.G; LJ.n = 1;
        for (int l = 0; l < D.N; ++l)
            for (int k = 0; k < 2; ++k) {
                LJ.part[LJ.n] = k == 0 ? W.lw[l].lnpart1 : W.lw[l].lnpart2;
                LJ.out_a[LJ.n] = dparams + (size_t)l * L.stride() + L.oln(2 * k);
                LJ.G[LJ.n] = D.GR; ++LJ.n;
            }
        hipLaunchKernelGGL(encoder_finalize_kernel, dim3(LJ.n * LJ.lnblocks + D.N * LJ.wblocks), dim3(1024), 0, st, S, L, dparams, LJ);
        LAUNCH_CHECK("encoder_finalize_kernel");
    }
    else {      // no layers: only the final norm
        ProfScope prof(S_FINALIZE, st);
        hipLaunchKernelGGL(ln_param_finalize_kernel, dim3((d + 31) / 32, 2, 1), dim3(1024), 0, st, W.lnpartf, D.G, L.DP, d, gPf, gPf + d,
                           (size_t)0, (size_t)0);
        LAUNCH_CHECK("ln_param_finalize_kernel");
    }
    return MMT_OK;
}

extern "C" int mmt_encoder_backward(const float* dy, const float* x, const float* mask, const float* params,
                                    float* dx, float* dparams,
                                    void* workspace, size_t workspace_bytes,
                                    int B, int T, int d, int h, int f, int n_layers, float eps,
                                    float dropout_p, uint64_t seed, mmt_stream_t stream) {
    return encoder_backward_impl(dy, x, mask, params, dx, dparams, workspace, workspace_bytes, B, T, d, h, f, n_layers, eps, dropout_p, seed, false, stream);
}
extern "C" int mmt_encoder_backward_devseed(const float* dy, const float* x, const float* mask, const float* params,
                                            float* dx, float* dparams,
                                            void* workspace, size_t workspace_bytes,
                                            int B, int T, int d, int h, int f, int n_layers, float eps,
                                            float dropout_p, mmt_stream_t stream) {
    return encoder_backward_impl(dy, x, mask, params, dx, dparams, workspace, workspace_bytes, B, T, d, h, f, n_layers, eps, dropout_p, 0, true, stream);
}

// ------------------------------------------------------------------------------------ LayerNorm alone
extern "C" size_t mmt_layernorm_scratch_floats(int M, int d) {
    return (size_t)((M + 31) / 32) * 2 * round_up(d, 64);
}

extern "C" int mmt_layernorm_forward(const float* x, const float* a_2, const float* b_2, float* y, float* stats,
                                     int M, int d, float eps, mmt_stream_t stream) {
    if (!x || !a_2 || !b_2 || !y) return fail(MMT_EINVAL, "null pointer argument");
    if (M <= 0 || d < 2) return fail(MMT_EINVAL, "bad shape M=%d d=%d", M, d);
    if (d % 4) return fail(MMT_EUNSUPPORTED, "feature count %d must be a multiple of 4", d);
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((M + 31) / 32), dim3(MMT_THREADS), 0, static_cast<hipStream_t>(stream),
                       x, a_2, b_2, eps, y, stats, M, d);
    LAUNCH_CHECK("layernorm_fwd_kernel");
    return MMT_OK;
}

extern "C" int mmt_layernorm_backward(const float* dy, const float* x, const float* a_2, const float* stats,
                                      float* dx, float* da_2, float* db_2, float* scratch,
                                      int M, int d, float eps, mmt_stream_t stream) {
    if (!dy || !x || !a_2 || !stats || !dx || !da_2 || !db_2 || !scratch) return fail(MMT_EINVAL, "null pointer argument");
    if (M <= 0 || d < 2) return fail(MMT_EINVAL, "bad shape M=%d d=%d", M, d);
    if (d % 4) return fail(MMT_EUNSUPPORTED, "feature count %d must be a multiple of 4", d);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int DP = round_up(d, 64), G = (M + 31) / 32;
    int rc = launch_ln_bwd(dy, x, a_2, stats, eps, dx, scratch, M, d, DP, st);
    if (rc) return rc;
    hipLaunchKernelGGL(ln_param_finalize_kernel, dim3((d + 31) / 32, 2, 1), dim3(1024), 0, st, scratch, G, DP, d, da_2, db_2, (size_t)0, (size_t)0);
    LAUNCH_CHECK("ln_param_finalize_kernel");
    return MMT_OK;
}

// ------------------------------------------------------------------------------------ attention core alone
// pack (B,T,d) fp32 head-major-column tensors into fragment layouts; unpack head-padded bf16 back.
__global__ void pack_frag_kernel(const float* __restrict__ src, bf16* __restrict__ fr,
                                 const float* __restrict__ rowmask, float scale, int use_mask,
                                 const bf16* __restrict__ ctx, int ldctx, float* __restrict__ delta,
                                 int M, int T, int Tp, int h, int dk, int DKP, int d) {
    const size_t total = (size_t)M * h;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / h), head = (int)(idx % h), b = m / T, t = m - b * T;
        const size_t bh = (size_t)b * h + head;
        float sc = scale;
        if (use_mask && rowmask && rowmask[m] == 0.0f) sc = 0.f;
        float part = 0.f;
        for (int e = 0; e < dk; ++e) {
            const bf16 v = (bf16)(src[(size_t)m * d + head * dk + e] * sc);
            fr[bh * fragR_elems(Tp, DKP) + fragR_index(t, e, DKP)] = v;
            if (delta) part += (float)v * (float)ctx[(size_t)m * ldctx + head * DKP + e];
        }
        if (delta) delta[bh * Tp + t] = -part;       // stored negated, like the fused path
    }
}

__global__ void unpad_heads_kernel(const bf16* __restrict__ src, int ld, int col0, float* __restrict__ dst,
                                   int M, int h, int dk, int DKP, int d) {
    const size_t total = (size_t)M * d;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / d), c = (int)(idx % d), head = c / dk, e = c - head * dk;
        dst[idx] = (float)src[(size_t)m * ld + col0 + head * DKP + e];
    }
}

struct SdpaWs { bf16 *QR, *KR, *VR, *dOR, *ctx, *dqkv; float *lse, *delta; uint16_t *maskQ, *maskK; size_t bytes; };
static void carve_sdpa(SdpaWs& W, const EncDims& D, void* base, bool train = true) {
    Carver c(base);
    const LayerLayout& L = D.L;
    const size_t BH = (size_t)D.B * D.h, M = D.M;
    bf16** r[] = {&W.QR, &W.KR, &W.VR, &W.dOR};
    for (int i = 0; i < 4; ++i) *r[i] = c.take<bf16>(BH * fragR_elems(D.Tp, L.DKP));
    W.ctx = c.take<bf16>(M * L.HDP);
    W.dqkv = c.take<bf16>(M * L.NQ);
    W.lse = c.take<float>(BH * D.Tp); W.delta = c.take<float>(BH * D.Tp);
    W.maskQ = W.maskK = nullptr;
    if (train) { W.maskQ = c.take<uint16_t>(attn_mask_layer_words(D.B * D.h, D.nt)); W.maskK = c.take<uint16_t>(attn_mask_layer_words(D.B * D.h, D.nt)); }
    W.bytes = c.off;       // the dropout bit masks are the last region: an eval-mode workspace is a prefix of a train-mode one
}

static int check_drop(float dropout_p, const EncDims& D) {
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(MMT_EINVAL, "dropout_p %g not in [0,1)", dropout_p);
    if (dropout_p > 0.f && D.Tp > 4096) return fail(MMT_EUNSUPPORTED, "train-mode dropout supports T <= 4096 (got %d)", D.T);
    return MMT_OK;
}

extern "C" size_t mmt_sdpa_workspace_bytes(int B, int T, int d, int h) {
    EncDims D;
    if (make_dims(D, B, T, d, h, 4, 0) != MMT_OK) return 0;
    SdpaWs W; carve_sdpa(W, D, nullptr, true);
    return W.bytes;
}
extern "C" size_t mmt_sdpa_workspace_bytes_eval(int B, int T, int d, int h) {
    EncDims D;
    if (make_dims(D, B, T, d, h, 4, 0) != MMT_OK) return 0;
    SdpaWs W; carve_sdpa(W, D, nullptr, false);
    return W.bytes;
}

extern "C" int mmt_sdpa_forward(const float* q, const float* k, const float* v, const float* mask, float* ctx,
                                void* workspace, size_t workspace_bytes, int B, int T, int d, int h,
                                float dropout_p, uint64_t seed, mmt_stream_t stream) {
    EncDims D;
    int rc = make_dims(D, B, T, d, h, 4, 0);
    if (rc) return rc;
    if (!q || !k || !v || !ctx || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    if ((rc = check_drop(dropout_p, D))) return rc;
    SdpaWs W; carve_sdpa(W, D, workspace, dropout_p > 0.f);
    if (workspace_bytes < W.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, W.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const LayerLayout& L = D.L;
    const int g = grid_for((size_t)D.M * h);
    const float qs = LOG2E / sqrtf((float)L.dk);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(g), dim3(256), 0, st, q, W.QR, mask, qs, 1, nullptr, 0, nullptr, D.M, T, D.Tp, h, L.dk, L.DKP, d);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(g), dim3(256), 0, st, k, W.KR, nullptr, 1.f, 0, nullptr, 0, nullptr, D.M, T, D.Tp, h, L.dk, L.DKP, d);
    hipLaunchKernelGGL(pack_frag_kernel, dim3(g), dim3(256), 0, st, v, W.VR, nullptr, 1.f, 0, nullptr, 0, nullptr, D.M, T, D.Tp, h, L.dk, L.DKP, d);
    LAUNCH_CHECK("pack_frag_kernel");
    if (dropout_p > 0.f && (rc = launch_mask_gen(W.maskQ, W.maskK, D, 1, dropout_p, seed, st))) return rc;
    if ((rc = launch_attn_fwd(L.DKP, W.QR, W.KR, W.VR, W.ctx, W.lse, D, st, make_drop(dropout_p, seed, 0, MMT_ATTN_DROP_BITS), W.maskQ))) return rc;
    hipLaunchKernelGGL(unpad_heads_kernel, dim3(grid_for((size_t)D.M * d)), dim3(256), 0, st, W.ctx, L.HDP, 0, ctx, D.M, h, L.dk, L.DKP, d);
    LAUNCH_CHECK("unpad_heads_kernel");
    return MMT_OK;
}

extern "C" int mmt_sdpa_backward(const float* dctx, const float* mask, float* dq, float* dk, float* dv,
                                 void* workspace, size_t workspace_bytes, int B, int T, int d, int h,
                                 float dropout_p, uint64_t seed, mmt_stream_t stream) {
    EncDims D;
    int rc = make_dims(D, B, T, d, h, 4, 0);
    if (rc) return rc;
    if (!dctx || !dq || !dk || !dv || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    if ((rc = check_drop(dropout_p, D))) return rc;
    SdpaWs W; carve_sdpa(W, D, workspace, dropout_p > 0.f);
    if (workspace_bytes < W.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, W.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const LayerLayout& L = D.L;
    hipLaunchKernelGGL(pack_frag_kernel, dim3(grid_for((size_t)D.M * h)), dim3(256), 0, st, dctx, W.dOR, nullptr, 1.f, 0,
                       W.ctx, L.HDP, W.delta, D.M, T, D.Tp, h, L.dk, L.DKP, d);
    LAUNCH_CHECK("pack_frag_kernel");
    if ((rc = launch_attn_bwd(L.DKP, W.QR, W.KR, W.VR, W.dOR, W.lse, W.delta, mask, W.dqkv, D, st,
                              make_drop(dropout_p, seed, 0, MMT_ATTN_DROP_BITS), W.maskQ, W.maskK))) return rc;     // the masks the forward generated
    const int g = grid_for((size_t)D.M * d);
    hipLaunchKernelGGL(unpad_heads_kernel, dim3(g), dim3(256), 0, st, W.dqkv, L.NQ, 0, dq, D.M, h, L.dk, L.DKP, d);
    hipLaunchKernelGGL(unpad_heads_kernel, dim3(g), dim3(256), 0, st, W.dqkv, L.NQ, L.HD, dk, D.M, h, L.dk, L.DKP, d);
    hipLaunchKernelGGL(unpad_heads_kernel, dim3(g), dim3(256), 0, st, W.dqkv, L.NQ, 2 * L.HD, dv, D.M, h, L.dk, L.DKP, d);
    LAUNCH_CHECK("unpad_heads_kernel");
    return MMT_OK;
}

// ------------------------------------------------------------------------------------ fused affine map alone
__global__ void pad_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int np) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < np) dst[i] = (i < n && src) ? src[i] : 0.f;
}

// g = dy * rowscale * act'(y) -> bf16 row-major [M][NP] (pad columns written as zeros): the A operand of the input-gradient GEMM and
// of the weight-gradient kernel
__global__ void grad_prep_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ rowscale,
                                 int act, bf16* __restrict__ g, int M, int N, int NP, float gscale) {      // gscale: 1/(1-p) of an output dropout (y > 0 <=> kept)
    const size_t total = (size_t)M * (NP >> 2);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / (NP >> 2)), n = (int)(idx % (NP >> 2)) * 4;
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = 0.f;
            if (n + i < N) {
                v = dy[(size_t)m * N + n + i];
                if (rowscale) v *= rowscale[m];
                if (act == 1) { v = (y[(size_t)m * N + n + i] > 0.f) ? v * gscale : 0.f; }
                else if (act == 2) { const float yy = y[(size_t)m * N + n + i]; v *= 1.f - yy * yy; }
                else if (act == 3) { const float yy = y[(size_t)m * N + n + i]; v *= yy * (1.f - yy); }
            }
            o[i] = (bf16)v;
        }
        *reinterpret_cast<bf16x4*>(g + (size_t)m * NP + n) = o;
    }
}

// fp32 [M][K] -> bf16 row-major [M][KP] (pad columns written as zeros): the B operand of the weight-gradient kernel
// (`drop_in`: the input dropout of the forward, index m*KP + k, regenerated here)
__global__ void cast_rows_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int M, int K, int KP, DropCfg drop_in,
                                 const uint64_t* __restrict__ seedword) {
    const DropCfg drop = drop_resolve(drop_in, seedword);
    const size_t total = (size_t)M * (KP >> 2);
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / (KP >> 2)), k = (int)(idx % (KP >> 2)) * 4;
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = (k + i < K) ? src[(size_t)m * K + k + i] : 0.f;
        if (drop.thr16) {
#pragma unroll
            for (int i = 0; i < 4; i += 2) {
                const uint32_t w = drop_pair(drop, (uint64_t)m * KP + k + i);
                v[i] = drop_lo(drop, w, v[i]); v[i + 1] = drop_hi(drop, w, v[i + 1]);
            }
        }
        bf16x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] = (bf16)v[i];
        *reinterpret_cast<bf16x4*>(dst + (size_t)m * KP + k) = o;
    }
}

struct LinWs { uint64_t* seedword; bf16 *Wp, *WTp, *g, *xb; float *bp, *sW, *sb; int KP, NP, MP, M16, nsplit, mchunk; size_t bytes; };
static void carve_linear(LinWs& W, int M, int K, int N, void* base) {
    Carver c(base);
    W.seedword = c.take<uint64_t>(MMT_SEED_BLOCK_WORDS);
    W.KP = round_up(K, 64); W.NP = round_up(N, 64); W.MP = round_up(M, 64); W.M16 = round_up(M, 16);
    const int tiles = (W.NP / 64) * (W.KP / 64);
    int s = (512 + tiles - 1) / tiles; if (s < 1) s = 1; if (s > 32) s = 32;
    W.mchunk = round_up((W.MP + s - 1) / s, 64);
    W.nsplit = (W.MP + W.mchunk - 1) / W.mchunk;
    W.Wp = c.take<bf16>((size_t)W.NP * W.KP); W.WTp = c.take<bf16>((size_t)W.KP * W.NP);
    W.bp = c.take<float>(W.NP);
    W.g = c.take<bf16>((size_t)W.MP * W.NP); W.xb = c.take<bf16>((size_t)W.MP * W.KP);      // MP rows: rows >= M stay zero for wgrad
    W.sW = c.take<float>((size_t)W.nsplit * W.NP * W.KP); W.sb = c.take<float>((size_t)W.nsplit * W.NP);
    W.bytes = c.off;
}

extern "C" size_t mmt_linear_workspace_bytes(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    LinWs W; carve_linear(W, M, K, N, nullptr);
    return W.bytes;
}

static int check_linear(int M, int K, int N) {
    if (M <= 0 || K <= 0 || N <= 0) return fail(MMT_EINVAL, "bad shape M=%d K=%d N=%d", M, K, N);
    if (K % 4) return fail(MMT_EUNSUPPORTED, "in_features %d must be a multiple of 4", K);
    return MMT_OK;
}

#define MMT_LINEAR_IN_STREAM 2000      // dropout stream ids of the affine map's input / output dropout
#define MMT_LINEAR_OUT_STREAM 2001
static int check_linear_drop(float in_p, float out_p, int act) {
    if (!(in_p >= 0.f && in_p < 1.f) || !(out_p >= 0.f && out_p < 1.f)) return fail(MMT_EINVAL, "dropout probability not in [0,1)");
    if (out_p > 0.f && act != 1) return fail(MMT_EUNSUPPORTED, "output dropout is implemented behind ReLU only (the mask rides on y > 0)");
    return MMT_OK;
}
static int linear_forward_impl(const float* x, const float* Wt, const float* b, const float* rowscale, float* y,
                               void* workspace, size_t workspace_bytes, int M, int K, int N, int act,
                               float in_p, float out_p, uint64_t seed, uint64_t* seed_state, mmt_stream_t stream) {
    int rc = check_linear(M, K, N);
    if (rc) return rc;
    if ((rc = check_linear_drop(in_p, out_p, act))) return rc;
    if (!x || !Wt || !y || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    LinWs W; carve_linear(W, M, K, N, workspace);
    if (workspace_bytes < W.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, W.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(pad_cast_kernel, dim3(grid_for((size_t)W.NP * W.KP)), dim3(256), 0, st, Wt, W.Wp, N, K, W.NP, W.KP, 0);
    hipLaunchKernelGGL(pad_f32_kernel, dim3((W.NP + 255) / 256), dim3(256), 0, st, b, W.bp, N, W.NP);
    LAUNCH_CHECK("pad_cast_kernel");
    RowGemmParams p = rg_zero();
    p.M = M; p.K = K; p.KP = W.KP; p.N = N; p.NP = W.NP;
    p.A = x; p.lda = K; p.W = W.Wp; p.bias = W.bp; p.act = act; p.rowscale = rowscale;
    p.out_f32 = y; p.ldo = N;
    const bool devseed = seed_state != nullptr && (in_p > 0.f || out_p > 0.f);
    if (devseed) {
        if ((rc = launch_seed_advance(seed_state, W.seedword, MMT_LINEAR_IN_STREAM, 2, st))) return rc;
        p.seedword = W.seedword;
    }
    p.a_drop = stream_drop(in_p, seed, MMT_LINEAR_IN_STREAM, devseed, 16, MMT_LINEAR_IN_STREAM);     // x -> drop(x) as the A tile is staged (index m*KP + k)
    p.drop = stream_drop(out_p, seed, MMT_LINEAR_OUT_STREAM, devseed, 16, MMT_LINEAR_IN_STREAM);     // behind the activation (index m*NP + n)
    return launch_rowgemm<EPI_PLAIN, false>(p, st, S_LINEAR_FWD);
}
extern "C" int mmt_linear_forward(const float* x, const float* Wt, const float* b, const float* rowscale, float* y,
                                  void* workspace, size_t workspace_bytes, int M, int K, int N, int act, mmt_stream_t stream) {
    return linear_forward_impl(x, Wt, b, rowscale, y, workspace, workspace_bytes, M, K, N, act, 0.f, 0.f, 0, nullptr, stream);
}
extern "C" int mmt_linear_dropout_forward(const float* x, const float* Wt, const float* b, const float* rowscale, float* y,
                                          void* workspace, size_t workspace_bytes, int M, int K, int N, int act,
                                          float in_dropout_p, float out_dropout_p, uint64_t seed, uint64_t* seed_state, mmt_stream_t stream) {
    return linear_forward_impl(x, Wt, b, rowscale, y, workspace, workspace_bytes, M, K, N, act, in_dropout_p, out_dropout_p, seed, seed_state, stream);
}

static int linear_backward_impl(const float* dy, const float* x, const float* Wt, const float* y, const float* rowscale,
                                float* dx, float* dW, float* db,
                                void* workspace, size_t workspace_bytes, int M, int K, int N, int act,
                                float in_p, float out_p, uint64_t seed, bool devseed_in, mmt_stream_t stream) {
    int rc = check_linear(M, K, N);
    if (rc) return rc;
    if ((rc = check_linear_drop(in_p, out_p, act))) return rc;
    if (!dy || !x || !Wt || !workspace || (act != 0 && !y)) return fail(MMT_EINVAL, "null pointer argument");
    LinWs W; carve_linear(W, M, K, N, workspace);
    if (workspace_bytes < W.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, W.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool devseed = devseed_in && (in_p > 0.f || out_p > 0.f);
    const uint64_t* seedword = devseed ? W.seedword : nullptr;              // left there by the forward
    const DropCfg in_drop = stream_drop(in_p, seed, MMT_LINEAR_IN_STREAM, devseed, 16, MMT_LINEAR_IN_STREAM);
    hipLaunchKernelGGL(grad_prep_kernel, dim3(grid_for((size_t)M * (W.NP / 4))), dim3(256), 0, st, dy, y, rowscale, act, W.g, M, N, W.NP,
                       make_drop(out_p, 0, 0).scale);
    LAUNCH_CHECK("grad_prep_kernel");
    if (dx) {
        hipLaunchKernelGGL(pad_cast_kernel, dim3(grid_for((size_t)W.KP * W.NP)), dim3(256), 0, st, Wt, W.WTp, K, N, W.KP, W.NP, 1);
        LAUNCH_CHECK("pad_cast_kernel");
        RowGemmParams p = rg_zero();
        p.M = M; p.K = W.NP; p.KP = W.NP; p.N = K; p.NP = W.KP;
        p.A = W.g; p.a_bf16 = 1; p.lda = W.NP; p.W = W.WTp;
        p.out_f32 = dx; p.ldo = K;
        p.drop = in_drop; p.seedword = seedword;        // d drop(x) / dx: the forward's input mask, index m*KP + k = this output's m*NP + n
        if ((rc = launch_rowgemm<EPI_PLAIN, false>(p, st, S_LINEAR_BWD_DX))) return rc;
    }
    if (dW || db) {
        hipLaunchKernelGGL(cast_rows_kernel, dim3(grid_for((size_t)M * (W.KP / 4))), dim3(256), 0, st, x, W.xb, M, K, W.KP, in_drop, seedword);
        LAUNCH_CHECK("cast_rows_kernel");
        WgradJobs J; memset(&J, 0, sizeof(J));
        J.njobs = 1; J.MP = W.MP; J.M16 = W.M16; J.mchunk = W.mchunk;
        J.j[0].A = W.g; J.j[0].lda = W.NP; J.j[0].B = W.xb; J.j[0].ldb = W.KP; J.j[0].out = W.sW; J.j[0].bias_out = W.sb;
        J.j[0].NPj = W.NP; J.j[0].KPj = W.KP; J.j[0].tile0 = 0; J.j[0].tiles_k = W.KP / 64;
        {
            ProfScope prof(S_LINEAR_WGRAD, st);
            hipLaunchKernelGGL(wgrad_kernel, dim3((W.NP / 64) * (W.KP / 64), W.nsplit), dim3(MMT_THREADS), 0, st, J);
        }
        LAUNCH_CHECK("wgrad_kernel");
        if (dW) hipLaunchKernelGGL(slab_sum_kernel, dim3(grid_for((size_t)N * K)), dim3(256), 0, st, W.sW, W.nsplit, W.NP, W.KP, dW, N, K);
        if (db) hipLaunchKernelGGL(slab_sum_kernel, dim3(grid_for((size_t)N)), dim3(256), 0, st, W.sb, W.nsplit, 1, W.NP, db, 1, N);
        LAUNCH_CHECK("slab_sum_kernel");
    }
    return MMT_OK;
}

extern "C" int mmt_linear_backward(const float* dy, const float* x, const float* Wt, const float* y, const float* rowscale,
                                   float* dx, float* dW, float* db,
                                   void* workspace, size_t workspace_bytes, int M, int K, int N, int act, mmt_stream_t stream) {
    return linear_backward_impl(dy, x, Wt, y, rowscale, dx, dW, db, workspace, workspace_bytes, M, K, N, act, 0.f, 0.f, 0, false, stream);
}
extern "C" int mmt_linear_dropout_backward(const float* dy, const float* x, const float* Wt, const float* y, const float* rowscale,
                                           float* dx, float* dW, float* db,
                                           void* workspace, size_t workspace_bytes, int M, int K, int N, int act,
                                           float in_dropout_p, float out_dropout_p, uint64_t seed, int device_seeded, mmt_stream_t stream) {
    return linear_backward_impl(dy, x, Wt, y, rowscale, dx, dW, db, workspace, workspace_bytes, M, K, N, act, in_dropout_p, out_dropout_p, seed,
                                device_seeded != 0, stream);
}

// ------------------------------------------------------------------------------------ data movement between the kernels (glue.h)
extern "C" int mmt_copy2d(const mmt_copy_seg* segs, int nsegs, mmt_stream_t stream) {
    if (!segs || nsegs <= 0) return fail(MMT_EINVAL, "copy2d: no segments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int s0 = 0; s0 < nsegs; s0 += MMT_COPY_MAX_SEGS) {
        CopySegs S; memset(&S, 0, sizeof(S));
        S.n = std::min(MMT_COPY_MAX_SEGS, nsegs - s0);
        size_t most = 0;
        for (int i = 0; i < S.n; ++i) {
            const mmt_copy_seg& a = segs[s0 + i];
            CopySeg& g = S.s[i];
            if (!a.dst || a.rows < 0 || a.cols < 0 || a.perm < 0 || a.perm > 2) return fail(MMT_EINVAL, "copy2d: bad segment %d", s0 + i);
            if (a.perm && ((long)a.pB * a.pT != a.rows || a.pB <= 0 || a.pT <= 0)) return fail(MMT_EINVAL, "copy2d: segment %d: rows != B*T", s0 + i);
            g.src = a.src; g.src2 = a.src2; g.dst = a.dst; g.rowscale = a.rowscale;
            g.rows = a.rows; g.cols = a.cols; g.src_ld = a.src_ld; g.src2_ld = a.src2_ld; g.dst_ld = a.dst_ld;
            g.perm = a.perm; g.pB = a.pB; g.pT = a.pT; g.accumulate = a.accumulate;
            auto al = [](const void* q, int ld) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0; };
            g.vec4 = (a.cols & 3) == 0 && al(a.dst, a.dst_ld) && (!a.src || al(a.src, a.src_ld)) && (!a.src2 || al(a.src2, a.src2_ld));
            most = std::max(most, (size_t)a.rows * (g.vec4 ? a.cols / 4 : a.cols));
        }
        if (most == 0) continue;
        hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(most), S.n), dim3(256), 0, st, S);
        LAUNCH_CHECK("copy2d_kernel");
    }
    return MMT_OK;
}
extern "C" int mmt_softmax_mul_forward(const float* logits, const float* v, float* att, float* out, int M, int N, mmt_stream_t stream) {
    if (!logits || !v || !att || !out) return fail(MMT_EINVAL, "null pointer argument");
    if (M <= 0 || N <= 0) return fail(MMT_EINVAL, "bad shape M=%d N=%d", M, N);
    hipLaunchKernelGGL(softmax_mul_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), logits, v, att, out, M, N);
    LAUNCH_CHECK("softmax_mul_fwd_kernel");
    return MMT_OK;
}
extern "C" int mmt_softmax_mul_backward(const float* dout, const float* att, const float* v, float* dlogits, float* dv, int M, int N,
                                        mmt_stream_t stream) {
    if (!dout || !att || !v || !dlogits || !dv) return fail(MMT_EINVAL, "null pointer argument");
    if (M <= 0 || N <= 0) return fail(MMT_EINVAL, "bad shape M=%d N=%d", M, N);
    hipLaunchKernelGGL(softmax_mul_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), dout, att, v, dlogits, dv, M, N);
    LAUNCH_CHECK("softmax_mul_bwd_kernel");
    return MMT_OK;
}
extern "C" int mmt_error_accumulate(const uint32_t* word, uint32_t* accum, mmt_stream_t stream) {
    if (!word || !accum) return fail(MMT_EINVAL, "null pointer argument");
    hipLaunchKernelGGL(error_accumulate_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), word, accum);
    LAUNCH_CHECK("error_accumulate_kernel");
    return MMT_OK;
}
extern "C" int mmt_colsum(const float* x, float* out, int rows, int cols, int ld, mmt_stream_t stream) {
    if (!x || !out) return fail(MMT_EINVAL, "null pointer argument");
    if (rows <= 0 || cols <= 0 || ld < cols) return fail(MMT_EINVAL, "bad shape rows=%d cols=%d ld=%d", rows, cols, ld);
    hipLaunchKernelGGL(colsum_kernel, dim3((cols + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, rows, cols, ld);
    LAUNCH_CHECK("colsum_kernel");
    return MMT_OK;
}

// Highway combine + Dropout(0.3) of the window encoder (glue.h)
extern "C" int mmt_highway_forward(const float* x, const float* proj, const float* gate, float* out, size_t n,
                                   float dropout_p, uint64_t seed, uint64_t* seed_state, uint64_t* seedblock, mmt_stream_t stream) {
    if (!x || !proj || !gate || !out) return fail(MMT_EINVAL, "null pointer argument");
    if (n == 0) return MMT_OK;
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(MMT_EINVAL, "dropout_p %g outside [0, 1)", dropout_p);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool devseed = seed_state != nullptr && dropout_p > 0.f;
    if (devseed) {
        if (!seedblock) return fail(MMT_EINVAL, "a device-resident seed needs the 2-word seed block the backward reads");
        int rc = launch_seed_advance(seed_state, seedblock, MMT_HIGHWAY_STREAM, 1, st);
        if (rc) return rc;
    }
    const DropCfg c = stream_drop(dropout_p, seed, MMT_HIGHWAY_STREAM, devseed, 16, MMT_HIGHWAY_STREAM);
    hipLaunchKernelGGL(highway_fwd_kernel, dim3(grid_for((n + 1) / 2)), dim3(256), 0, st, x, proj, gate, out, n, c, devseed ? seedblock : nullptr);
    LAUNCH_CHECK("highway_fwd_kernel");
    return MMT_OK;
}
extern "C" int mmt_highway_backward(const float* dout, const float* x, const float* proj, const float* gate, float* dx, float* dproj, float* dgate,
                                    size_t n, float dropout_p, uint64_t seed, const uint64_t* seedblock, mmt_stream_t stream) {
    if (!dout || !x || !proj || !gate || !dx || !dproj || !dgate) return fail(MMT_EINVAL, "null pointer argument");
    if (n == 0) return MMT_OK;
    if (!(dropout_p >= 0.f && dropout_p < 1.f)) return fail(MMT_EINVAL, "dropout_p %g outside [0, 1)", dropout_p);
    const bool devseed = seedblock != nullptr && dropout_p > 0.f;
    const DropCfg c = stream_drop(dropout_p, seed, MMT_HIGHWAY_STREAM, devseed, 16, MMT_HIGHWAY_STREAM);
    hipLaunchKernelGGL(highway_bwd_kernel, dim3(grid_for((n + 1) / 2)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dout, x, proj, gate, dx, dproj, dgate, n, c, devseed ? seedblock : nullptr);
    LAUNCH_CHECK("highway_bwd_kernel");
    return MMT_OK;
}

// ------------------------------------------------------------------------------------ LSTM scan
struct LstmWs { unsigned* err; cl_u64* xb; bf16 *Wf, *Wb; size_t xb_bytes; int HP16, HPAD; size_t bytes; };
#define MMT_SCAN_ERR_BYTES 256            // error block at workspace offset 0 (word 0: exchange time-out of the four-CU scans)
static int carve_lstm(LstmWs& W, int H, void* base) {
    if (H <= 0 || H > 256) return fail(MMT_EUNSUPPORTED, "LSTM hidden size %d not in [4,256]", H);
    if (H % 4) return fail(MMT_EUNSUPPORTED, "LSTM hidden size %d must be a multiple of 4", H);
    W.HP16 = round_up(H, 16);
    W.HPAD = W.HP16 <= 64 ? 64 : (W.HP16 <= 128 ? 128 : 256);
    Carver c(base);
    W.err = c.take<unsigned>(MMT_SCAN_ERR_BYTES / sizeof(unsigned));     // FIRST: callers read the error word at offset 0
    W.xb_bytes = W.HPAD == 256 ? (size_t)32 * 2 * CL_NP * 128 * sizeof(cl_u64) : 0;   // exchange granules of the 4-CU scans (<= 32 sequences)
    W.xb = c.take<cl_u64>(W.xb_bytes / sizeof(cl_u64));                  // directly behind it: one memset node clears both
    W.Wf = c.take<bf16>((size_t)4 * W.HP16 * W.HPAD); W.Wb = c.take<bf16>((size_t)(W.HP16 + 16) * 4 * W.HPAD);   // +16 rows: scan256 reads a ragged tile whole
    W.bytes = c.off;
    return MMT_OK;
}
extern "C" size_t mmt_lstm_scan_workspace_bytes(int H) { LstmWs W; return carve_lstm(W, H, nullptr) ? 0 : W.bytes; }

// The four-CU scans (scan_cluster.h) need the four workgroups of every sequence resident together.  Take that path only when the
// occupancy query admits at least TWICE the grid on this device (margin for kernels of other streams); anything that still goes
// wrong at run time (another process holding the CUs) ends in the kernels' bounded waits and the workspace's error word.
template <typename K> static bool cl4_resident(K kernel, int grid) {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kernel), 256, 0) != hipSuccess) return false;
    return (long)per_cu * cus >= 2L * grid;
}

// ... evaluated once per (device, kernel): the answer depends on the device that is current, not on the first one that asked
template <typename K> static int cl4_fits(K kernel, int which) {
    static int cache[2][64];                    // 0: unknown, 1: no, 2: yes
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return cl4_resident(kernel, 128) ? 1 : 0;
    if (!cache[which][dev]) cache[which][dev] = cl4_resident(kernel, 128) ? 2 : 1;     // 128: the largest grid this path launches
    return cache[which][dev] == 2;
}

// sequences per scan workgroup: the smallest power of two that still fits the batch into <= 256 workgroups (one per CU)
static int scan_bt(int B) { int bt = 1; while (bt < 16 && (B + bt - 1) / bt > 256) bt *= 2; return bt; }

template <typename K> static int set_lds_attr(K kernel) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return MMT_OK;
}

extern "C" int mmt_lstm_scan_forward(const float* gx, const float* W_rec, const float* h0, const float* c0,
                                     float* h_all, float* c_all, float* acts, void* workspace, size_t workspace_bytes,
                                     int T, int B, int H, mmt_stream_t stream) {
    if (!gx || !W_rec || !h_all || !c_all || !acts || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    if (T <= 0 || B <= 0) return fail(MMT_EINVAL, "bad shape T=%d B=%d", T, B);
    LstmWs W; int rc = carve_lstm(W, H, workspace);
    if (rc) return rc;
    if (workspace_bytes < W.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, W.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(lstm_prep_kernel, dim3(grid_for((size_t)8 * W.HP16 * W.HPAD)), dim3(256), 0, st, W_rec, W.Wf, W.Wb, H, W.HP16, W.HPAD);
    LAUNCH_CHECK("lstm_prep_kernel");
    const int BT = scan_bt(B);
    const dim3 grid((B + BT - 1) / BT), block(64 * (W.HP16 / 16));
    ProfScope prof(S_LSTM_FWD, st);
#define MMT_LSTM_FWD(KS, NT, WREG, PF) hipLaunchKernelGGL((lstm_scan_fwd_kernel<KS, NT, WREG, PF>), grid, block, 0, st, \
        gx, W.Wf, h0, c0, h_all, c_all, acts, T, B, H, W.HP16, BT)
#define MMT_LSTM_FWD_U(KS, NT, WREG, PF, NR) hipLaunchKernelGGL((lstm_scan_fwd_u_kernel<KS, NT, WREG, PF, NR>), grid, block, 0, st, \
        gx, W.Wf, h0, c0, h_all, c_all, acts, T, B, H, W.HP16)
    static const bool no_cluster = getenv("MMT_NO_CLUSTER_SCAN") != nullptr;
    const int cl_grid = 32 * ((B + 7) / 8);
    const int cl_fit = cl4_fits(&lstm_scan_fwd_cl4_kernel<3>, 0);
    const bool cluster = W.HPAD == 256 && B <= 32 && !no_cluster && cl_fit;
    {   // error word (+ exchange granules) cleared by a kernel of the launch sequence (not hipMemsetAsync: see zero_fill_kernel)
        const size_t nw = (MMT_SCAN_ERR_BYTES + (cluster ? W.xb_bytes : 0)) / sizeof(unsigned);
        hipLaunchKernelGGL(zero_fill_kernel, dim3(grid_for(nw)), dim3(256), 0, st, W.err, nw);
        LAUNCH_CHECK("zero_fill_kernel");
    }
    if (cluster) {       // four CUs per sequence, weights fully register-resident (scan_cluster.h)
        hipLaunchKernelGGL((lstm_scan_fwd_cl4_kernel<3>), dim3(cl_grid), dim3(256), 0, st,
                           gx, W.Wf, h0, c0, h_all, c_all, acts, W.xb, W.err, T, B, H, W.HP16);
    } else if (W.HPAD == 256 && BT == 1) {   // half-resident weights, one sequence per workgroup (scan256.h)
        hipLaunchKernelGGL((lstm_scan_fwd256_kernel<4, 1>), dim3(B), dim3(64 * ((W.HP16 + 31) / 32)), 0, st,
                           gx, W.Wf, h0, c0, h_all, c_all, acts, T, B, H, W.HP16);
    } else if (BT <= 2) {       // one / two sequences per workgroup: units on the lanes (scan_units.h)
        if (BT == 1) {
            if (W.HPAD == 64) MMT_LSTM_FWD_U(2, 256, true, 6, 1);
            else if (W.HPAD == 128) MMT_LSTM_FWD_U(4, 512, true, 6, 1);
            else MMT_LSTM_FWD_U(8, 1024, false, 2, 1);
        } else {
            if (W.HPAD == 64) MMT_LSTM_FWD_U(2, 256, true, 6, 2);
            else if (W.HPAD == 128) MMT_LSTM_FWD_U(4, 512, true, 6, 2);
            else MMT_LSTM_FWD_U(8, 1024, false, 1, 2);
        }
    } else {
        if (W.HPAD == 64) MMT_LSTM_FWD(2, 256, true, 4);
        else if (W.HPAD == 128) MMT_LSTM_FWD(4, 512, true, 2);
        else MMT_LSTM_FWD(8, 1024, false, 1);
    }
#undef MMT_LSTM_FWD
#undef MMT_LSTM_FWD_U
    LAUNCH_CHECK("lstm_scan_fwd_kernel");
    return MMT_OK;
}

extern "C" int mmt_lstm_scan_backward(const float* dh_all, const float* dc_all, const float* W_rec, const float* c0,
                                      const float* c_all, const float* acts, float* dgx, float* dh0, float* dc0,
                                      void* workspace, size_t workspace_bytes, int T, int B, int H, mmt_stream_t stream) {
    if (!W_rec || !c_all || !acts || !dgx || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    if (T <= 0 || B <= 0) return fail(MMT_EINVAL, "bad shape T=%d B=%d", T, B);
    LstmWs W; int rc = carve_lstm(W, H, workspace);
    if (rc) return rc;
    if (workspace_bytes < W.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, W.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(lstm_prep_kernel, dim3(grid_for((size_t)8 * W.HP16 * W.HPAD)), dim3(256), 0, st, W_rec, W.Wf, W.Wb, H, W.HP16, W.HPAD);
    LAUNCH_CHECK("lstm_prep_kernel");
    const int BT = scan_bt(B);
    const dim3 grid((B + BT - 1) / BT), block(64 * (W.HP16 / 16));
    const bool coop = BT <= 2;
    const size_t lds = (size_t)2 * 16 * (4 * W.HPAD + 8) * 2;
    static bool attr = false;
    if (!attr) {
        if ((rc = set_lds_attr(&lstm_scan_bwd_kernel<32, 1024, false, 1>))) return rc;
        if ((rc = set_lds_attr(&lstm_scan_bwd_u_kernel<32, 1024, false, 2, 1>))) return rc;
        if ((rc = set_lds_attr(&lstm_scan_bwd_u_kernel<32, 1024, false, 2, 2>))) return rc;
        attr = true;
    }
    ProfScope prof(S_LSTM_BWD, st);
#define MMT_LSTM_BWD(KS4, NT, WREG, PF) hipLaunchKernelGGL((lstm_scan_bwd_kernel<KS4, NT, WREG, PF>), grid, block, lds, st, \
        dh_all, dc_all, W.Wb, c0, c_all, acts, dgx, dh0, dc0, T, B, H, W.HP16, BT)
#define MMT_LSTM_BWD_U(KS4, NT, WREG, PF, NR) hipLaunchKernelGGL((lstm_scan_bwd_u_kernel<KS4, NT, WREG, PF, NR>), grid, block, lds, st, \
        dh_all, dc_all, W.Wb, c0, c_all, acts, dgx, dh0, dc0, T, B, H, W.HP16)
    static const bool no_cluster = getenv("MMT_NO_CLUSTER_SCAN") != nullptr;
    const int cl_fit = cl4_fits(&lstm_scan_bwd_cl4_kernel<2>, 1);
    const bool cluster = W.HPAD == 256 && B <= 32 && !no_cluster && cl_fit;
    {   // error word (+ exchange granules) cleared by a kernel of the launch sequence (not hipMemsetAsync: see zero_fill_kernel)
        const size_t nw = (MMT_SCAN_ERR_BYTES + (cluster ? W.xb_bytes : 0)) / sizeof(unsigned);
        hipLaunchKernelGGL(zero_fill_kernel, dim3(grid_for(nw)), dim3(256), 0, st, W.err, nw);
        LAUNCH_CHECK("zero_fill_kernel");
    }
    if (cluster) {
        hipLaunchKernelGGL((lstm_scan_bwd_cl4_kernel<2>), dim3(32 * ((B + 7) / 8)), dim3(256), 0, st,
                           dh_all, dc_all, W.Wb, c0, c_all, acts, dgx, dh0, dc0, W.xb, W.err, T, B, H, W.HP16);
    } else if (W.HPAD == 256 && BT == 1) {
        static bool attr256 = false;
        if (!attr256) { if ((rc = set_lds_attr(&lstm_scan_bwd256_kernel<16, 1>))) return rc; attr256 = true; }
        const size_t lds256 = (size_t)2 * 16 * (4 * 256 + 8) * 2 + (size_t)(2 * 8 * 256 + 8 * 32) * sizeof(float);
        hipLaunchKernelGGL((lstm_scan_bwd256_kernel<16, 1>), dim3(B), dim3(64 * ((W.HP16 + 31) / 32)), lds256, st,
                           dh_all, dc_all, W.Wb, c0, c_all, acts, dgx, dh0, dc0, T, B, H, W.HP16);
    } else if (coop) {
        if (BT == 1) {
            if (W.HPAD == 64) MMT_LSTM_BWD_U(8, 256, true, 6, 1);
            else if (W.HPAD == 128) MMT_LSTM_BWD_U(16, 512, true, 6, 1);
            else MMT_LSTM_BWD_U(32, 1024, false, 2, 1);
        } else {
            if (W.HPAD == 64) MMT_LSTM_BWD_U(8, 256, true, 4, 2);
            else if (W.HPAD == 128) MMT_LSTM_BWD_U(16, 512, true, 4, 2);
            else MMT_LSTM_BWD_U(32, 1024, false, 2, 2);
        }
    } else {
        if (W.HPAD == 64) MMT_LSTM_BWD(8, 256, true, 2);
        else if (W.HPAD == 128) MMT_LSTM_BWD(16, 512, true, 2);
        else MMT_LSTM_BWD(32, 1024, false, 1);
    }
#undef MMT_LSTM_BWD
#undef MMT_LSTM_BWD_U
    LAUNCH_CHECK("lstm_scan_bwd_kernel");
    return MMT_OK;
}

#ifdef MMT_ABLATIONS
extern "C" int mmt_debug_set_attn_stamp_buffer(void* buf) {
    unsigned long long* b = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &b, sizeof(b)) == hipSuccess ? 0 : 1;
}
#endif
#ifdef MMT_PHASE_TIMING
extern "C" int mmt_debug_set_phase_buffer(void* buf) {
    unsigned long long* b = static_cast<unsigned long long*>(buf);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_buf), &b, sizeof(b)) == hipSuccess ? 0 : 1;
}
#endif

// ------------------------------------------------------------------------------------ window encoder (conv k=2 + max-pool)
struct ConvWs { bf16* Wp; float* slab; float* dbpart; int FPAD, DP, DPB, nsplit, wins; size_t bytes; };
static int carve_conv(ConvWs& C, int N, int W, int D, int F, void* base) {
    if (N <= 0 || W < 2 || D <= 0 || F <= 0) return fail(MMT_EINVAL, "bad shape N=%d W=%d D=%d F=%d", N, W, D, F);
    if (D % 4) return fail(MMT_EUNSUPPORTED, "window encoder needs a raw feature size divisible by 4 (got %d)", D);
    C.FPAD = round_up(F, CP_FB); C.DP = round_up(D, CP_KC); C.DPB = round_up(D, CP_DB);
    const int blocks = (C.DPB / CP_DB) * (C.FPAD / CP_FB);
    int ns = (512 + blocks - 1) / blocks;                   // ~2 workgroups per CU
    if (ns > (N + 1) / 2) ns = (N + 1) / 2;
    if (ns < 1) ns = 1;
    C.wins = round_up((N + ns - 1) / ns, 2);
    C.nsplit = (N + C.wins - 1) / C.wins;
    Carver c(base);
    C.Wp = c.take<bf16>((size_t)2 * C.FPAD * C.DP);
    C.slab = c.take<float>((size_t)C.nsplit * 2 * C.FPAD * C.DPB);
    C.dbpart = c.take<float>((size_t)C.nsplit * C.FPAD);
    C.bytes = c.off;
    return MMT_OK;
}
extern "C" size_t mmt_convpool_workspace_bytes(int N, int W, int D, int F) {
    ConvWs C;
    return carve_conv(C, N, W, D, F, nullptr) ? 0 : C.bytes;
}

extern "C" int mmt_convpool_forward(const float* x, const float* weight, const float* bias, float* out, int32_t* argmax,
                                    void* workspace, size_t workspace_bytes, int N, int W, int D, int F, mmt_stream_t stream) {
    ConvWs C; int rc = carve_conv(C, N, W, D, F, workspace);
    if (rc) return rc;
    if (!x || !weight || !bias || !out || !argmax || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    if (workspace_bytes < C.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, C.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    static bool attr = false;
    if (!attr) {
        if ((rc = set_lds_attr(&convpool_fwd_kernel<4>)) || (rc = set_lds_attr(&convpool_fwd_kernel<2>)) ||
            (rc = set_lds_attr(&convpool_fwd_kernel<1>))) return rc;
        attr = true;
    }
    hipLaunchKernelGGL(convpool_prep_kernel, dim3(grid_for((size_t)2 * C.FPAD * C.DP)), dim3(256), 0, st, weight, C.Wp, F, D, C.FPAD, C.DP);
    LAUNCH_CHECK("convpool_prep_kernel");
    ProfScope prof(S_CONV_FWD, st);
    const int nwg = (N + CP_WIN - 1) / CP_WIN, nmain = F / CP_FB, rem = F - nmain * CP_FB;
#define MMT_CONV_FWD(CT, blocks, first) hipLaunchKernelGGL((convpool_fwd_kernel<CT>), dim3(nwg, blocks), dim3(512), convpool_fwd_lds_bytes(CT), st, \
        x, C.Wp, bias, out, argmax, N, W, D, C.DP, F, C.FPAD, first)
    if (nmain > 0) MMT_CONV_FWD(4, nmain, 0);
    if (rem > 128) MMT_CONV_FWD(4, 1, nmain * CP_FB);
    else if (rem > 64) MMT_CONV_FWD(2, 1, nmain * CP_FB);
    else if (rem > 0) MMT_CONV_FWD(1, 1, nmain * CP_FB);
#undef MMT_CONV_FWD
    LAUNCH_CHECK("convpool_fwd_kernel");
    return MMT_OK;
}

extern "C" int mmt_convpool_backward(const float* x, const float* dout, const int32_t* argmax, float* dweight, float* dbias,
                                     void* workspace, size_t workspace_bytes, int N, int W, int D, int F, mmt_stream_t stream) {
    ConvWs C; int rc = carve_conv(C, N, W, D, F, workspace);
    if (rc) return rc;
    if (!x || !dout || !argmax || !dweight || !dbias || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    if (workspace_bytes < C.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, C.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    static bool attr = false;
    if (!attr) { if ((rc = set_lds_attr(&convpool_bwd_kernel<true>)) || (rc = set_lds_attr(&convpool_bwd_kernel<false>))) return rc; attr = true; }
    {
        ProfScope prof(S_CONV_BWD, st);
        const dim3 grid(C.DPB / CP_DB, C.nsplit, C.FPAD / CP_FB);
        if (W <= 33) hipLaunchKernelGGL(convpool_bwd_kernel<true>, grid, dim3(512), convpool_bwd_lds_bytes(), st,
                                        x, dout, argmax, C.slab, C.dbpart, N, W, D, F, C.FPAD, C.wins);
        else hipLaunchKernelGGL(convpool_bwd_kernel<false>, grid, dim3(512), convpool_bwd_lds_bytes(), st,
                                x, dout, argmax, C.slab, C.dbpart, N, W, D, F, C.FPAD, C.wins);
    }
    LAUNCH_CHECK("convpool_bwd_kernel");
    hipLaunchKernelGGL(convpool_finish_kernel, dim3(grid_for((size_t)F * D * 2 + F)), dim3(256), 0, st, C.slab, C.dbpart, dweight, dbias,
                       C.nsplit, D, F, C.FPAD, C.DPB);
    LAUNCH_CHECK("convpool_finish_kernel");
    return MMT_OK;
}

// ------------------------------------------------------------------------------------ MFN memory scan
struct MfnWs { uint64_t* seedword; bf16 *WmF, *W2F, *WmB, *W2B; size_t bytes; };
static void carve_mfn(MfnWs& W, void* base) {
    Carver c(base);
    W.seedword = c.take<uint64_t>(MMT_SEED_BLOCK_WORDS);
    W.WmF = c.take<bf16>(MFN_U * MFN_MD); W.W2F = c.take<bf16>(2 * MFN_MD * MFN_HG);
    W.WmB = c.take<bf16>(MFN_MD * MFN_U); W.W2B = c.take<bf16>(MFN_U * MFN_MD);
    W.bytes = c.off;
}
extern "C" size_t mmt_mfn_mem_scan_workspace_bytes(void) { MfnWs W; carve_mfn(W, nullptr); return W.bytes; }

static int check_mfn_dims(int mem_dim, int h_gamma) {
    if (mem_dim != MFN_MD || h_gamma != MFN_HG)
        return fail(MMT_EUNSUPPORTED, "MFN memory scan is built for mem_dim=%d, gamma hidden=%d (got %d, %d)", MFN_MD, MFN_HG, mem_dim, h_gamma);
    return MMT_OK;
}

static int mfn_mem_scan_forward_impl(const float* apre, const float* chat, const float* Wm, const float* W2, const float* b2,
                                     float* mem_all, float* u_all, float* g_all, void* workspace, size_t workspace_bytes,
                                     int T, int B, int mem_dim, int h_gamma, float dropout_p, uint64_t seed, uint64_t* seed_state, mmt_stream_t stream) {
    int rc = check_mfn_dims(mem_dim, h_gamma);
    if (rc) return rc;
    if (!apre || !chat || !Wm || !W2 || !b2 || !mem_all || !u_all || !g_all || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    if (T <= 0 || B <= 0) return fail(MMT_EINVAL, "bad shape T=%d B=%d", T, B);
    MfnWs W; carve_mfn(W, workspace);
    if (workspace_bytes < W.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, W.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mfn_prep_kernel, dim3(64), dim3(256), 0, st, Wm, W2, W.WmF, W.W2F, W.WmB, W.W2B);
    LAUNCH_CHECK("mfn_prep_kernel");
    const bool devseed = seed_state != nullptr && dropout_p > 0.f;
    // (the gamma mask is not regenerated by the backward — u_all keeps the dropped values — so nothing reads the block after this launch)
    if (devseed && (rc = launch_seed_advance(seed_state, W.seedword, 1000, 1, st))) return rc;
    ProfScope prof(S_MEM_FWD, st);
    const int BT = scan_bt(B);
    const DropCfg dcfg = stream_drop(dropout_p, seed, 1000, devseed, 16, 1000);
    const uint64_t* sw = devseed ? W.seedword : nullptr;
    if (BT == 1)        // one / two sequences per workgroup: units on the lanes (scan.h)
        hipLaunchKernelGGL(mfn_mem_scan_fwd_sw_kernel<1>, dim3(B), dim3(512), 0, st, apre, chat, W.WmF, W.W2F, b2, mem_all, u_all, g_all, T, B, dcfg, sw);
    else if (BT == 2)
        hipLaunchKernelGGL(mfn_mem_scan_fwd_sw_kernel<2>, dim3((B + 1) / 2), dim3(512), 0, st, apre, chat, W.WmF, W.W2F, b2, mem_all, u_all, g_all, T, B, dcfg, sw);
    else
        hipLaunchKernelGGL(mfn_mem_scan_fwd_kernel, dim3((B + BT - 1) / BT), dim3(512), 0, st, apre, chat, W.WmF, W.W2F, b2, mem_all, u_all, g_all, T, B, BT, dcfg, sw);
    LAUNCH_CHECK("mfn_mem_scan_fwd_kernel");
    return MMT_OK;
}
extern "C" int mmt_mfn_mem_scan_forward(const float* apre, const float* chat, const float* Wm, const float* W2, const float* b2,
                                        float* mem_all, float* u_all, float* g_all, void* workspace, size_t workspace_bytes,
                                        int T, int B, int mem_dim, int h_gamma, float dropout_p, uint64_t seed, mmt_stream_t stream) {
    return mfn_mem_scan_forward_impl(apre, chat, Wm, W2, b2, mem_all, u_all, g_all, workspace, workspace_bytes, T, B, mem_dim, h_gamma, dropout_p, seed, nullptr, stream);
}
extern "C" int mmt_mfn_mem_scan_forward_devseed(const float* apre, const float* chat, const float* Wm, const float* W2, const float* b2,
                                                float* mem_all, float* u_all, float* g_all, void* workspace, size_t workspace_bytes,
                                                int T, int B, int mem_dim, int h_gamma, float dropout_p, uint64_t* seed_state, mmt_stream_t stream) {
    if (!seed_state) return fail(MMT_EINVAL, "null seed state");
    return mfn_mem_scan_forward_impl(apre, chat, Wm, W2, b2, mem_all, u_all, g_all, workspace, workspace_bytes, T, B, mem_dim, h_gamma, dropout_p, 0, seed_state, stream);
}

extern "C" int mmt_mfn_mem_scan_backward(const float* dmem_all, const float* chat, const float* mem_all, const float* u_all,
                                         const float* g_all, const float* Wm, const float* W2,
                                         float* dchat, float* dapre, float* dz_all, void* workspace, size_t workspace_bytes,
                                         int T, int B, int mem_dim, int h_gamma, float dropout_p, mmt_stream_t stream) {
    int rc = check_mfn_dims(mem_dim, h_gamma);
    if (rc) return rc;
    if (!chat || !mem_all || !u_all || !g_all || !Wm || !W2 || !dchat || !dapre || !dz_all || !workspace) return fail(MMT_EINVAL, "null pointer argument");
    if (T <= 0 || B <= 0) return fail(MMT_EINVAL, "bad shape T=%d B=%d", T, B);
    MfnWs W; carve_mfn(W, workspace);
    if (workspace_bytes < W.bytes) return fail(MMT_EWORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, W.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mfn_prep_kernel, dim3(64), dim3(256), 0, st, Wm, W2, W.WmF, W.W2F, W.WmB, W.W2B);
    LAUNCH_CHECK("mfn_prep_kernel");
    ProfScope prof(S_MEM_BWD, st);
    const int BT = scan_bt(B);
    const float dscale = make_drop(dropout_p, 0, 0).scale;
    if (BT == 1)
        hipLaunchKernelGGL(mfn_mem_scan_bwd_sw_kernel<1>, dim3(B), dim3(512), 0, st, dmem_all, chat, mem_all, u_all, g_all, W.WmB, W.W2B, dchat, dapre, dz_all, T, B, dscale);
    else if (BT == 2)
        hipLaunchKernelGGL(mfn_mem_scan_bwd_sw_kernel<2>, dim3((B + 1) / 2), dim3(512), 0, st, dmem_all, chat, mem_all, u_all, g_all, W.WmB, W.W2B, dchat, dapre, dz_all, T, B, dscale);
    else
        hipLaunchKernelGGL(mfn_mem_scan_bwd_kernel, dim3((B + BT - 1) / BT), dim3(512), 0, st, dmem_all, chat, mem_all, u_all, g_all, W.WmB, W.W2B,
                           dchat, dapre, dz_all, T, B, BT, dscale);
    LAUNCH_CHECK("mfn_mem_scan_bwd_kernel");
    return MMT_OK;
}


// ------------------------------------------------------------------------------------ loss
extern "C" size_t mmt_mse_sum_scratch_doubles(size_t n) { return (size_t)grid_for(n, 1024); }

extern "C" int mmt_mse_sum_forward(const float* pred, const float* target, float inv_denom, float* loss, float* dpred, double* scratch,
                                   size_t n, mmt_stream_t stream) {
    if (!pred || !target || !loss || !dpred || !scratch) return fail(MMT_EINVAL, "null pointer argument");
    if (n == 0) return fail(MMT_EINVAL, "empty loss");
    if ((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(dpred)) & 15)
        return fail(MMT_EINVAL, "mse_sum: pred, target and dpred must be 16-byte aligned (the kernel moves four floats per lane)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int g = grid_for(n, 1024);
    hipLaunchKernelGGL(mse_sum_partial_kernel, dim3(g), dim3(256), 0, st, pred, target, inv_denom, dpred, scratch, n);
    hipLaunchKernelGGL(mse_sum_final_kernel, dim3(1), dim3(256), 0, st, scratch, g, inv_denom, loss);
    LAUNCH_CHECK("mse_sum kernels");
    return MMT_OK;
}

// ------------------------------------------------------------------------------------ optimiser
extern "C" int mmt_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                             const size_t* counts, int nchunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                             int step, mmt_stream_t stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !counts) return fail(MMT_EINVAL, "null pointer argument");
    if (nchunks <= 0 || step <= 0) return fail(MMT_EINVAL, "adam: nchunks %d, step %d must be positive", nchunks, step);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // bias corrections in double, as torch.optim.Adam takes them (python floats): step_size = lr / (1 - beta1^t), sqrt(1 - beta2^t)
    const float step_size = (float)((double)lr / (1.0 - pow((double)beta1, (double)step)));
    const float bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    for (int c0 = 0; c0 < nchunks; c0 += MMT_ADAM_MAX_CHUNKS) {
        AdamChunks C; memset(&C, 0, sizeof(C));
        const int nc = std::min(MMT_ADAM_MAX_CHUNKS, nchunks - c0);
        size_t nmax = 0;
        for (int c = 0; c < nc; ++c) {
            C.p[c] = params[c0 + c]; C.g[c] = grads[c0 + c]; C.m[c] = exp_avg[c0 + c]; C.v[c] = exp_avg_sq[c0 + c]; C.n[c] = counts[c0 + c];
            if (!C.p[c] || !C.g[c] || !C.m[c] || !C.v[c]) return fail(MMT_EINVAL, "adam: null pointer in chunk %d", c0 + c);
            nmax = std::max(nmax, counts[c0 + c]);
        }
        const int gx = (int)std::min<size_t>(512, (nmax + 1023) / 1024);
        hipLaunchKernelGGL(adam_step_kernel, dim3(gx > 0 ? gx : 1, nc), dim3(256), 0, st, C, step_size, beta1, beta2, eps, weight_decay, bc2s);
        LAUNCH_CHECK("adam_step_kernel");
    }
    return MMT_OK;
}

// ------------------------------------------------------------------------------------ metric
extern "C" int mmt_ccc_forward(const float* pred, const float* target, const int32_t* lengths, double* ccc, int B, int T, mmt_stream_t stream) {
    if (!pred || !target || !lengths || !ccc) return fail(MMT_EINVAL, "null pointer argument");
    if (B <= 0 || T <= 0) return fail(MMT_EINVAL, "bad shape B=%d T=%d", B, T);
    hipLaunchKernelGGL(ccc_kernel, dim3(B), dim3(256), 0, static_cast<hipStream_t>(stream), pred, target, lengths, ccc, T);
    LAUNCH_CHECK("ccc_kernel");
    return MMT_OK;
}

// ------------------------------------------------------------------------------------ test hook
// keep[i] = 1 if index i of dropout stream `stream` is kept under (p, seed): lets a test rebuild the exact masks the
// kernels used and replay the reference arithmetic with them.
// attn_Tp == 0: flat stream (index i).  attn_Tp > 0: attention-probability stream, i = (bh*Tp + q)*Tp + key, which the
// attention kernels evaluate as a per-(batch,head) stream with the 32-bit index q*Tp + key.
__global__ void dropout_mask_kernel(DropCfg c, uint64_t n, uint8_t* __restrict__ keep) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        keep[i] = drop_keep(c, i) ? 1 : 0;
}
// Test hook: leave a chosen bit pattern in every LDS word (and a spread of VGPRs) of every CU.  A kernel whose result depends on LDS
// it never wrote (what a freshly powered GPU hands it: the first process on a box) then produces a different answer after this call.
__global__ __launch_bounds__(1024) void poison_lds_kernel(uint32_t pattern, int words, uint32_t* sink) {
    extern __shared__ uint32_t poison_smem[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) poison_smem[i] = pattern;
    __syncthreads();
    // keep the stores alive and hold the CU for a moment so that the grid spreads over all CUs
    uint32_t acc = 0;
    for (int i = threadIdx.x; i < words; i += blockDim.x * 7) acc ^= poison_smem[i];
    if (acc == 0x12345u) sink[0] = acc;
}
// ... and in the vector registers: a wave starts with whatever the previous wave on its SIMD slot left there.
__global__ __launch_bounds__(256, 2) void poison_vgpr_kernel(uint32_t pattern, uint32_t* sink) {
    constexpr int NR = 232;
    uint32_t r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { r[i] = pattern; asm volatile("" : "+v"(r[i])); }
    __builtin_amdgcn_s_sleep(64);
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) { asm volatile("" : "+v"(r[i])); acc += r[i] ^ (uint32_t)i; }
    if (acc == 0x12345u) sink[0] = acc;
}
extern "C" int mmt_debug_poison_lds(uint32_t pattern, void* sink4, mmt_stream_t stream) {
    if (!sink4) return fail(MMT_EINVAL, "null pointer argument");
    static bool configured = false;
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    hipLaunchKernelGGL(poison_lds_kernel, dim3(2048), dim3(1024), 160 * 1024, static_cast<hipStream_t>(stream), pattern, 160 * 256, static_cast<uint32_t*>(sink4));
    LAUNCH_CHECK("poison_lds_kernel");
    hipLaunchKernelGGL(poison_vgpr_kernel, dim3(8192), dim3(256), 0, static_cast<hipStream_t>(stream), pattern, static_cast<uint32_t*>(sink4));
    LAUNCH_CHECK("poison_vgpr_kernel");
    return MMT_OK;
}

extern "C" int mmt_debug_dropout_mask(float p, uint64_t seed, uint32_t stream_id, uint64_t n, uint32_t attn_Tp, uint8_t* keep,
                                      float* scale_out, mmt_stream_t stream) {
    if (!keep) return fail(MMT_EINVAL, "null pointer argument");
    const DropCfg c = make_drop(p, seed, stream_id, attn_Tp ? MMT_ATTN_DROP_BITS : 16);
    if (scale_out) *scale_out = c.scale;        // host pointer
    if (attn_Tp) {                              // attention-probability stream: the generator's own block function (attn_mask.h)
        if (attn_Tp % 32 || n % ((uint64_t)attn_Tp * attn_Tp)) return fail(MMT_EINVAL, "attention mask: n must be a multiple of Tp*Tp, Tp of 32");
        const int nt = (int)(attn_Tp / 32), nbh = (int)(n / ((uint64_t)attn_Tp * attn_Tp));
        const size_t blocks = (size_t)nbh * nt * nt;
        hipLaunchKernelGGL(attn_mask_expand_kernel, dim3((unsigned)((blocks + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream), c, nbh, nt, keep);
        LAUNCH_CHECK("attn_mask_expand_kernel");
        return MMT_OK;
    }
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, static_cast<hipStream_t>(stream), c, n, keep);
    LAUNCH_CHECK("dropout_mask_kernel");
    return MMT_OK;
}
