/* mmt_hip.h — C ABI of libmmt_hip.so, the MI355X (gfx950) implementation of the attention hot path
 * of frankaging/Multimodal-Transformer.
 *
 * The reference has no FFI: its boundary for this path is the Python class surface of
 * transformer/{SFT,MFT,B2-Trans}/multiTransformer.py.  Each entry point below replaces
 * the forward (or the autograd backward) of one of those classes and is what a binding for the
 * reference would call (ctypes stub: INTEGRATION.md).  Conventions:
 *   - every pointer is a DEVICE pointer unless named host_*; tensors are contiguous fp32 in the
 *     reference's own layouts: activations (B,T,d) row-major, mask (B,T,1) float {0,1} (a window is
 *     blanked where mask == 0, multiTransformer.py:30), nn.Linear weights (out,in);
 *   - the library allocates nothing: the caller passes a workspace of *_workspace_bytes() bytes that
 *     MUST be zero-filled when first created and may then be reused for calls of the same shape
 *     (pad regions are never written; saved-for-backward tensors live in it between a forward and its
 *     backward);
 *   - all work is enqueued on `stream` (a hipStream_t); nothing synchronises, so calls are capturable
 *     into a hipGraph;
 *   - return 0 on success, non-zero MMT_E* on error with a message in mmt_last_error() (thread local).
 *     Asynchronous HIP faults surface at the caller's next synchronisation, as in PyTorch.
 */
#ifndef MMT_HIP_H
#define MMT_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MMT_OK 0
#define MMT_EINVAL 1      /* bad shape / null pointer */
#define MMT_EUNSUPPORTED 2 /* e.g. d_k > 64, d_model > 512, d % 4 != 0 */
#define MMT_EWORKSPACE 3  /* workspace too small */
#define MMT_EHIP 4        /* a HIP runtime call failed */

typedef void* mmt_stream_t; /* hipStream_t */

int mmt_abi_version(void);
const char* mmt_last_error(void);

/* ---- Per-launch-site timing with HIP events on the caller's stream (eager mode only: do not enable
 * while capturing a hipGraph).  Used by bench.py for the roofline of the dominant kernel. */
int mmt_profile_enable(int on);
int mmt_profile_reset(void);
int mmt_profile_num_sites(void);
const char* mmt_profile_site_name(int site);
/* waits for the recorded events; total_ms[num_sites], launches[num_sites] */
int mmt_profile_collect(float* total_ms, int* launches);

/* ---- Encoder stack: N x (pre-norm self-attention sublayer + pre-norm FFN sublayer) + final LayerNorm.
 * Replaces Encoder.forward(x, mask)                       transformer/MFT/multiTransformer.py:73-76
 *   with EncoderLayer.forward / SublayerConnection.forward                              :103-104,114-116
 *        MultiHeadedAttention.forward + attention()                                      :22-34,47-65
 *        PositionwiseFeedForward.forward                                                 :19-20
 *        LayerNorm.forward (unbiased std, eps added to std)                              :88-91
 * `params`: one flat fp32 buffer, per layer in the reference's registration order
 *   self_attn.linears.{0,1,2,3}.{weight(d,d),bias(d)}, feed_forward.w_1.{weight(f,d),bias(f)},
 *   feed_forward.w_2.{weight(d,f),bias(d)}, sublayer.0.norm.{a_2,b_2}, sublayer.1.norm.{a_2,b_2},
 * followed by the stack's norm.{a_2,b_2}.  mmt_encoder_param_count() gives its length.
 * Eval-mode (dropout = identity) when dropout_p == 0; see mmt_encoder_forward's `dropout_p`, `seed`. */
size_t mmt_encoder_param_count(int d, int f, int n_layers);
size_t mmt_encoder_workspace_bytes(int B, int T, int d, int h, int f, int n_layers);
/* ... for calls with dropout_p == 0 only (eval mode): without the attention-dropout bit masks (2 x n_layers x B*h x Tp^2/4 bytes, the
 * last region of the full workspace: an eval workspace is a prefix of a train one).  A forward and its backward take the same kind. */
size_t mmt_encoder_workspace_bytes_eval(int B, int T, int d, int h, int f, int n_layers);

int mmt_encoder_forward(const float* x, const float* mask, const float* params, float* y,
                        void* workspace, size_t workspace_bytes,
                        int B, int T, int d, int h, int f, int n_layers, float eps,
                        float dropout_p, uint64_t seed, mmt_stream_t stream);

/* Backward of the above (what torch autograd derives for the reference).  Must follow a forward on the
 * same workspace.  dx: (B,T,d); dparams: flat, same layout as params (fully overwritten). */
int mmt_encoder_backward(const float* dy, const float* x, const float* mask, const float* params,
                         float* dx, float* dparams,
                         void* workspace, size_t workspace_bytes,
                         int B, int T, int d, int h, int f, int n_layers, float eps,
                         float dropout_p, uint64_t seed, mmt_stream_t stream);

/* Device-resident dropout seed: the same pair for a step that is captured into a hipGraph and replayed, where a seed passed by value
 * would be frozen at capture (the reference's nn.Dropout draws fresh masks every step: transformer/MFT/multiTransformer.py:17,45,101,
 * torch keeps its generator state on the device under graph capture for the same reason).  `seed_state`: one uint64 in device memory,
 * owned by the caller.  The forward's first kernel copies it into the workspace (the seed of THIS forward and of its backward) and
 * advances it (splitmix64), so every replay draws new masks; the masks of a step are those mmt_debug_dropout_mask gives for the value
 * seed_state held before that step.  The backward takes no seed: it reads the workspace. */
int mmt_encoder_forward_devseed(const float* x, const float* mask, const float* params, float* y,
                                void* workspace, size_t workspace_bytes,
                                int B, int T, int d, int h, int f, int n_layers, float eps,
                                float dropout_p, uint64_t* seed_state, mmt_stream_t stream);
int mmt_encoder_backward_devseed(const float* dy, const float* x, const float* mask, const float* params,
                                 float* dx, float* dparams,
                                 void* workspace, size_t workspace_bytes,
                                 int B, int T, int d, int h, int f, int n_layers, float eps,
                                 float dropout_p, mmt_stream_t stream);

/* ---- LayerNorm alone.  Replaces LayerNorm.forward            transformer/MFT/multiTransformer.py:88-91
 * stats: (M,2) fp32 scratch kept for the backward (mean, 1/(std+eps)).
 * colpart: scratch of mmt_layernorm_scratch_floats(M,d) floats. */
size_t mmt_layernorm_scratch_floats(int M, int d);
int mmt_layernorm_forward(const float* x, const float* a_2, const float* b_2, float* y, float* stats,
                          int M, int d, float eps, mmt_stream_t stream);
int mmt_layernorm_backward(const float* dy, const float* x, const float* a_2, const float* stats,
                           float* dx, float* da_2, float* db_2, float* scratch,
                           int M, int d, float eps, mmt_stream_t stream);

/* ---- Scaled dot-product attention core on already-projected q, k, v.
 * Replaces attention(query, key, value, mask)                 transformer/MFT/multiTransformer.py:22-34
 * q,k,v,ctx: (B,T,d) fp32 with head `i` in columns [i*d/h,(i+1)*d/h) (the layout before the reference's
 * .view(B,-1,h,d_k).transpose(1,2)); mask (B,T,1) blanks QUERY rows; may be NULL.
 * dropout_p, seed: the reference's nn.Dropout on p_attn (:32-33) in train mode (0 = eval / dropout=None); the decisions are
 * drawn by the forward into the workspace and re-read by the backward (same dropout_p and seed must be passed); they are
 * dropout stream 0 of mmt_debug_dropout_mask. */
size_t mmt_sdpa_workspace_bytes(int B, int T, int d, int h);
size_t mmt_sdpa_workspace_bytes_eval(int B, int T, int d, int h);      /* dropout_p == 0 calls: without the dropout bit masks */
int mmt_sdpa_forward(const float* q, const float* k, const float* v, const float* mask, float* ctx,
                     void* workspace, size_t workspace_bytes, int B, int T, int d, int h,
                     float dropout_p, uint64_t seed, mmt_stream_t stream);
int mmt_sdpa_backward(const float* dctx, const float* mask, float* dq, float* dk, float* dv,
                      void* workspace, size_t workspace_bytes, int B, int T, int d, int h,
                      float dropout_p, uint64_t seed, mmt_stream_t stream);

/* ---- Fused affine map  y = act(x W^T + b) [* rowscale] on bf16 MFMA.
 * Replaces nn.Linear (+ F.relu) call sites of the path: PositionwiseFeedForward (:15-20), the four
 * attention projections (:43,55,65), embeds and read-out MLPs (:270,296,340-342,400-402).
 * x (M,K), W (N,K), b (N) or NULL, y (M,N), all fp32; act: 0 none, 1 ReLU, 2 tanh, 3 sigmoid; rowscale (M) or NULL. */
size_t mmt_linear_workspace_bytes(int M, int K, int N);
int mmt_linear_forward(const float* x, const float* W, const float* b, const float* rowscale, float* y,
                       void* workspace, size_t workspace_bytes, int M, int K, int N, int act, mmt_stream_t stream);
/* dx (M,K) or NULL; dW (N,K), db (N) or NULL.  `y` is the forward output (activation-derivative source when act != 0). */
int mmt_linear_backward(const float* dy, const float* x, const float* W, const float* y, const float* rowscale,
                        float* dx, float* dW, float* db,
                        void* workspace, size_t workspace_bytes, int M, int K, int N, int act, mmt_stream_t stream);

/* ... with train-mode dropout fused in: on the INPUT (x -> drop(x) while the A tile is staged: the `Dropout(0.1) -> Linear -> ReLU` embed of
 * NLPTransformer, transformer/SFT/multiTransformer.py:431-433,461) and / or behind the ReLU (act must be 1: MFN's out_dropout on
 * relu(out_fc1), transformer/MFT/multiTransformer.py:244-245).  Streams 2000 (input, index m*KP + k, KP = K rounded up to 64) and 2001
 * (output, index m*NP + n) of mmt_debug_dropout_mask.  seed_state: NULL (seed by value) or a device-resident seed as in
 * mmt_encoder_forward_devseed; the backward then passes device_seeded = 1 (the seed sits in the workspace) and any seed. */
int mmt_linear_dropout_forward(const float* x, const float* W, const float* b, const float* rowscale, float* y,
                               void* workspace, size_t workspace_bytes, int M, int K, int N, int act,
                               float in_dropout_p, float out_dropout_p, uint64_t seed, uint64_t* seed_state, mmt_stream_t stream);
int mmt_linear_dropout_backward(const float* dy, const float* x, const float* W, const float* y, const float* rowscale,
                                float* dx, float* dW, float* db,
                                void* workspace, size_t workspace_bytes, int M, int K, int N, int act,
                                float in_dropout_p, float out_dropout_p, uint64_t seed, int device_seeded, mmt_stream_t stream);

/* ---- Data movement between the kernels of a sequence model, so that its forward and backward run no library kernel: up to any number of
 * strided 2-D fp32 copies per call (24 per launch).  Replaces the torch.cat / stack / permute / slicing / broadcasting glue of
 * MFN.forward (cStar = [c_{t-1}; c_t] :212-217, [h; mem] :241-243), MultiTransformer.forward (permute :300, `* mask` :310) and of the SFT decoder
 * (transformer/SFT/multiTransformer.py:463-483) and their autograd twins.
 * dst[perm(r)][c] (+)= rowscale[.] * (src[r][c] + src2[r][c]) for r < rows, c < cols.  src NULL: zeros; src2 optional; a stride of 0
 * broadcasts one row; perm 1: source rows are batch-major (b*T+t), destination rows time-major (t*B+b); perm 2: the reverse; rowscale is
 * indexed by the batch-major row under a permutation.  Segments of one call must not overlap in their destinations. */
typedef struct {
    const float* src; const float* src2; float* dst; const float* rowscale;
    int rows, cols, src_ld, src2_ld, dst_ld, perm, pB, pT, accumulate;
} mmt_copy_seg;
int mmt_copy2d(const mmt_copy_seg* host_segs, int nsegs, mmt_stream_t stream);

/* attended = softmax(logits, over the N features) * v, att kept for the backward.   Replaces transformer/MFT/multiTransformer.py:218-219
 * backward: dlogits = att * (dout * v - sum_n dout v att), dv = dout * att. */
int mmt_softmax_mul_forward(const float* logits, const float* v, float* att, float* out, int M, int N, mmt_stream_t stream);
int mmt_softmax_mul_backward(const float* dout, const float* att, const float* v, float* dlogits, float* dv, int M, int N, mmt_stream_t stream);
/* out[c] = sum over `rows` rows of x (row stride ld): gradient of a row that the forward broadcast over the batch. */
int mmt_colsum(const float* x, float* out, int rows, int cols, int ld, mmt_stream_t stream);

/* Highway combine of the window encoder with the Dropout(0.3) the front-end applies to it.
 * Replaces `x_gate * x_proj + (1 - x_gate) * x_conv_out` (transformer/SFT/models.py:47-55; MFT / B2-Trans twins) and `self.dropout(...)`
 * (transformer/SFT/models.py:132-134): out = drop(gate * proj + (1 - gate) * x) over n elements; proj / gate are the outputs of the two
 * Linear layers (mmt_linear_forward, act 0 / 3).  Dropout stream 3000 of mmt_debug_dropout_mask, index = element.
 * seed_state: NULL (seed by value) or a device-resident seed as in mmt_encoder_forward_devseed; then `seedblock` (2 uint64 of device
 * memory) receives the seed of this call and its stream keys, and the backward takes the same block instead of a seed.
 * backward: g = drop'(dout); dx = g (1 - gate) [the direct path only: add the two Linear layers' input gradients], dproj = g gate,
 * dgate = g (proj - x) [w.r.t. the gate's OUTPUT: the sigmoid's derivative is mmt_linear_backward's, act 3]. */
int mmt_highway_forward(const float* x, const float* proj, const float* gate, float* out, size_t n,
                        float dropout_p, uint64_t seed, uint64_t* seed_state, uint64_t* seedblock, mmt_stream_t stream);
int mmt_highway_backward(const float* dout, const float* x, const float* proj, const float* gate, float* dx, float* dproj, float* dgate,
                         size_t n, float dropout_p, uint64_t seed, const uint64_t* seedblock, mmt_stream_t stream);

/* accum[0] |= word[0] on the device, in stream order: keeps a kernel's device error word (mmt_lstm_scan_*: first word of the workspace)
 * beyond the life of its workspace, also inside a captured hipGraph.  Both pointers are device memory. */
int mmt_error_accumulate(const uint32_t* word, uint32_t* accum, mmt_stream_t stream);

/* ---- LSTM recurrence over T steps with the input projection already applied.
 * Replaces the per-step nn.LSTMCell loop of MFN.forward       transformer/MFT/multiTransformer.py:200-208
 * and the per-step nn.LSTM call of the SFT decoder            transformer/SFT/multiTransformer.py:471-476.
 * gx (T,B,4H) = x_t W_ih^T + b_ih + b_hh (gate order i,f,g,o); W_rec (4H,H) multiplies h_{t-1};
 * h0,c0 (B,H) or NULL (zeros).  Outputs h_all, c_all (T,B,H) and the gate activations acts (T,B,4H)
 * kept for the backward.  H % 4 == 0, H <= 256.
 * Device error word: the first uint32 of `workspace` is zeroed by every call and is non-zero once the kernels have run
 * if a scan for H > 128 (four workgroups per sequence exchanging h through memory) timed out waiting for a partner
 * workgroup — e.g. another process held the CUs.  The outputs are then INVALID.  Nothing synchronises here: read the word
 * after the stream has drained (the Python wrapper does, functional.check_device_errors()). */
size_t mmt_lstm_scan_workspace_bytes(int H);
int mmt_lstm_scan_forward(const float* gx, const float* W_rec, const float* h0, const float* c0,
                          float* h_all, float* c_all, float* acts, void* workspace, size_t workspace_bytes,
                          int T, int B, int H, mmt_stream_t stream);
/* dh_all, dc_all: gradients on the (T,B,H) outputs (either may be NULL).  dgx (T,B,4H) is the gradient of gx
 * (also the operand of the batched dW_ih / dW_rec products); dh0, dc0 (B,H) or NULL. */
int mmt_lstm_scan_backward(const float* dh_all, const float* dc_all, const float* W_rec, const float* c0,
                           const float* c_all, const float* acts, float* dgx, float* dh0, float* dc0,
                           void* workspace, size_t workspace_bytes, int T, int B, int H, mmt_stream_t stream);

/* ---- MFN delta-memory recurrence.  Replaces the memory update inside MFN.forward's time loop
 *                                                             transformer/MFT/multiTransformer.py:221-224
 * apre (T,B,128): gamma{1,2}_fc1 applied to the `attended` part of `both` (+bias), rows [gamma1(64); gamma2(64)];
 * Wm (128,128): the memory columns of gamma{1,2}_fc1.weight stacked the same way; W2 (2,128,64), b2 (2,128):
 * gamma{1,2}_fc2; chat (T,B,128) = cHat.  Outputs mem_all (T,B,128) and u_all (T,B,128), g_all (T,B,256) kept
 * for the backward.  mem_dim must be 128 and h_gamma 64 (the reference's constants, :133,140-141).
 * dropout_p / seed: gamma{1,2}_dropout on relu(fc1) in train mode (0 = eval). */
size_t mmt_mfn_mem_scan_workspace_bytes(void);
int mmt_mfn_mem_scan_forward(const float* apre, const float* chat, const float* Wm, const float* W2, const float* b2,
                             float* mem_all, float* u_all, float* g_all, void* workspace, size_t workspace_bytes,
                             int T, int B, int mem_dim, int h_gamma, float dropout_p, uint64_t seed, mmt_stream_t stream);
/* the same with a device-resident seed (see mmt_encoder_forward_devseed); the backward needs none (u_all keeps the dropped values) */
int mmt_mfn_mem_scan_forward_devseed(const float* apre, const float* chat, const float* Wm, const float* W2, const float* b2,
                                     float* mem_all, float* u_all, float* g_all, void* workspace, size_t workspace_bytes,
                                     int T, int B, int mem_dim, int h_gamma, float dropout_p, uint64_t* seed_state, mmt_stream_t stream);
/* dmem_all (T,B,128) or NULL -> dchat (T,B,128), dapre (T,B,128), dz_all (T,B,256) (pre-sigmoid gate gradients). */
int mmt_mfn_mem_scan_backward(const float* dmem_all, const float* chat, const float* mem_all, const float* u_all,
                              const float* g_all, const float* Wm, const float* W2,
                              float* dchat, float* dapre, float* dz_all, void* workspace, size_t workspace_bytes,
                              int T, int B, int mem_dim, int h_gamma, float dropout_p, mmt_stream_t stream);

/* ---- Window encoder: Conv1d(D -> F, kernel 2, bias) over the W positions (tokens / frames) of each window followed by
 *      the global max-pool over the W-1 conv positions.  Replaces CNN.forward
 *                                                             transformer/SFT/models.py:57-79 (call site :118-127;
 *                                                             same class in MFT/models.py:57-79, B2-Trans/models.py:57-79)
 * x fp32 (N, W, D) contiguous: N = B*T windows (the reference loops over the batch; windows are independent), D % 4 == 0,
 * W >= 2.  weight fp32 (F, D, 2) in nn.Conv1d's layout, bias (F).  out fp32 (N, F); argmax int32 (N, F): the conv position
 * of the maximum (first one on ties), saved for the backward.  Only kernel size 2 (the reference's constant, :82). */
size_t mmt_convpool_workspace_bytes(int N, int W, int D, int F);
int mmt_convpool_forward(const float* x, const float* weight, const float* bias, float* out, int32_t* argmax,
                         void* workspace, size_t workspace_bytes, int N, int W, int D, int F, mmt_stream_t stream);
/* dout (N, F) -> dweight (F, D, 2), dbias (F).  There is no dx: the windows are input data (the reference never
 * differentiates them either). */
int mmt_convpool_backward(const float* x, const float* dout, const int32_t* argmax, float* dweight, float* dbias,
                          void* workspace, size_t workspace_bytes, int N, int W, int D, int F, mmt_stream_t stream);

/* ---- Training loss and its gradient in one pass.
 * Replaces criterion(output, target) / sum(lengths) and its autograd backward   transformer/SFT/train.py:133-139 (criterion :538)
 * loss[0] = sum((pred - target)^2) * inv_denom (fp64 accumulation, deterministic);  dpred = 2 (pred - target) * inv_denom.
 * pred, target, dpred: n fp32 (16-byte aligned); scratch: mmt_mse_sum_scratch_doubles(n) doubles; all on the device. */
size_t mmt_mse_sum_scratch_doubles(size_t n);
int mmt_mse_sum_forward(const float* pred, const float* target, float inv_denom, float* loss, float* dpred, double* scratch,
                        size_t n, mmt_stream_t stream);

/* ---- Optimiser step.
 * Replaces optimizer.step() of torch.optim.Adam(lr, weight_decay) (L2 form, not AdamW)   transformer/SFT/train.py:621, called at :141
 * on `nchunks` contiguous fp32 ranges (parameter, gradient, first and second moment of chunk c: counts[c] elements each) in one
 * launch per 48 chunks.  `step` = 1, 2, ... (bias corrections 1 - beta^step are taken on the host).  The pointer arrays are HOST arrays
 * of device pointers. */
int mmt_adam_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                  const size_t* counts, int nchunks, float lr, float beta1, float beta2, float eps, float weight_decay,
                  int step, mmt_stream_t stream);

/* ---- Concordance correlation coefficient per sequence, on the device.
 * Replaces eval_ccc + the per-sequence host loop of evaluate()     transformer/SFT/train.py:42-50, :236-238
 * pred, target: (B, T) fp32 row-major (the (B,T,1) valence tensors); lengths: B int32 on the device; sequence b uses its first
 * lengths[b] windows.  ccc: B float64 on the device: 2 cov / (var_t + var_p + (mean_p - mean_t)^2) with population moments
 * (np.var, np.cov(bias=True)), accumulated in fp64; NaN for a sequence with fewer than 2 windows or zero spread in both. */
int mmt_ccc_forward(const float* pred, const float* target, const int32_t* lengths, double* ccc, int B, int T, mmt_stream_t stream);

/* ---- Test hook: the keep-mask (1 = kept) of dropout stream `stream_id` for indices [0,n) under (p, seed), and the
 * scale applied to kept values (host pointer, may be NULL).  Streams used by the encoder stack for layer l:
 * 4l+0 attention probabilities, index ((b*h+head)*Tp + q)*Tp + key (Tp = T rounded up to 32; pass attn_Tp = Tp and
 *      n = B*h*Tp*Tp; 0 for every other stream).  These are stored bit masks drawn once per forward (csrc/attn_mask.h);
 * 4l+1 / 4l+3 sublayer outputs and 4l+2 FFN hidden, index m*NP + n (NP = width rounded up to 64);
 * 1000: MFN gamma hidden, index (t*B+b)*128 + j. */
int mmt_debug_dropout_mask(float p, uint64_t seed, uint32_t stream_id, uint64_t n, uint32_t attn_Tp, uint8_t* keep,
                           float* host_scale_out, mmt_stream_t stream);

/* ---- Test hook: fill every LDS word of every CU, and the vector registers of every SIMD, with `pattern` (e.g. 0x7FC00000, a NaN).
 * Kernels must not depend on LDS or registers they did not write; tests run a workload, poison, run it again and require identical results.  `sink4`: any 4 writable device bytes. */
int mmt_debug_poison_lds(uint32_t pattern, void* sink4, mmt_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MMT_HIP_H */
