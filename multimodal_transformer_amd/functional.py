"""torch.autograd bindings of the HIP entry points (include/mmt_hip.h).

Each Function is the forward/backward pair of one reference class:
  encoder_stack  <- Encoder.forward                  transformer/MFT/multiTransformer.py:73-76
  layer_norm     <- LayerNorm.forward                :88-91
  sdpa           <- attention()                      :22-34
  linear         <- nn.Linear (+ReLU) call sites     :15-20,43,55,65
All tensors must live on a HIP device; there is no CPU path.
"""
import torch

from . import _lib


def _f32c(t):
    return t.detach().contiguous().float() if t is not None else None


def _f32c16(t):
    """contiguous fp32 AND 16-byte aligned (a contiguous view at an odd storage offset is not): for kernels that move four floats per lane"""
    t = _f32c(t)
    return t.clone() if t is not None and t.data_ptr() % 16 else t


# Sub-batch streams.  Sequences are independent, so one stack call may run as k sub-batches on k HIP streams (forked and joined by event,
# inside one hipGraph too).  Every kernel of a stack launches all its workgroups in ONE round at the BASELINE sizes, i.e. a step is a
# chain of ~34 kernel latencies; with two half-batches in flight the attention kernels (bound by vector issue) of one run beside the row
# chains (bound by latency) of the other, and one half's tail overlaps the other's head.  Measured at configs[3]: 32 sequences as 2 x 16:
# -3.5 % step time, 64 as 2 x 32: -11 %; 4 streams were no better than 2.  The pieces use seeds of their own (different dropout masks).
_SPLIT_STREAMS = _lib.StreamFork()


def _row_chunks(B, k):
    k = max(1, min(int(k), B))
    base, rem = divmod(B, k)
    out, lo = [], 0
    for i in range(k):
        hi = lo + base + (1 if i < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def _chunk_seed(seed, i, nsplit=1):
    if isinstance(seed, (list, tuple)):
        return seed[i]
    if isinstance(seed, _lib.DeviceSeed):
        if nsplit > 1:          # seed_advance_kernel reads and writes the state word: sub-batches on concurrent streams need one each
            raise ValueError("encoder_stack: %d sub-batch streams need a list of %d DeviceSeeds, got one" % (nsplit, nsplit))
        return seed
    return seed if i == 0 else _lib.mix64(seed, i)


class _EncoderStackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, flat_params, h, d_ff, n_layers, eps, dropout_p, seed, _needs=False, nsplit=1):
        lib = _lib.load()
        _lib.require_hip(x, mask, flat_params)
        x_, m_, p_ = _f32c(x), _f32c(mask), _f32c(flat_params)
        B, T, d = x_.shape
        if m_.numel() != B * T:
            raise ValueError("mask must have B*T = %d elements (shape (B,T,1)), got %s" % (B * T, tuple(mask.shape)))
        need = lib.mmt_encoder_param_count(d, d_ff, n_layers)
        if p_.numel() != need:
            raise ValueError("flat parameter buffer has %d elements, expected %d" % (p_.numel(), need))
        train = dropout_p > 0.0                    # eval-mode workspaces carry no dropout bit masks
        if isinstance(seed, (list, tuple)) and len(seed) < nsplit:
            raise ValueError("encoder_stack: one seed per sub-batch stream")
        y = torch.empty_like(x_)
        chunks = _row_chunks(B, nsplit)
        main, streams = _SPLIT_STREAMS.begin(x_.device, len(chunks))
        parts = []
        for i, ((b0, b1), st) in enumerate(zip(chunks, streams)):
            with torch.cuda.stream(st):
                Bi = b1 - b0
                nbytes = (lib.mmt_encoder_workspace_bytes if train else lib.mmt_encoder_workspace_bytes_eval)(Bi, T, d, h, d_ff, n_layers)
                if nbytes == 0:
                    _lib.check(lib.mmt_encoder_forward(None, None, None, None, None, 0, Bi, T, d, h, d_ff, n_layers, eps, 0.0, 0, None))
                ws = _lib.POOL.get(nbytes, x_.device, tag=("encoder", Bi, T, d, h, d_ff, n_layers, train))
                sd = _chunk_seed(seed, i, len(chunks))
                xp, mp, yp = _lib.ptr(x_) + 4 * b0 * T * d, _lib.ptr(m_) + 4 * b0 * T, _lib.ptr(y) + 4 * b0 * T * d
                if isinstance(sd, _lib.DeviceSeed):  # device-resident seed: read and advanced by the launch itself (hipGraph replays)
                    _lib.check(lib.mmt_encoder_forward_devseed(xp, mp, _lib.ptr(p_), yp, _lib.ptr(ws), nbytes,
                                                               Bi, T, d, h, d_ff, n_layers, eps, dropout_p, sd.ptr(), _lib.stream_ptr()))
                else:
                    _lib.check(lib.mmt_encoder_forward(xp, mp, _lib.ptr(p_), yp, _lib.ptr(ws), nbytes,
                                                       Bi, T, d, h, d_ff, n_layers, eps, dropout_p, sd, _lib.stream_ptr()))
                parts.append((b0, b1, ws, nbytes, sd))
        _SPLIT_STREAMS.end(main, streams)
        needs_bwd = _needs or any(ctx.needs_input_grad)
        if needs_bwd:
            ctx.save_for_backward(x_, m_, p_)
            ctx.parts = parts
            ctx.cfg = (B, T, d, h, d_ff, n_layers, eps, dropout_p)
        else:
            for part in parts:
                _lib.POOL.put(part[2])
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x_, m_, p_ = ctx.saved_tensors
        B, T, d, h, d_ff, n_layers, eps, dropout_p = ctx.cfg
        if ctx.parts is None:
            raise RuntimeError("encoder_stack: backward called twice on the same forward (workspace already released)")
        parts, ctx.parts = ctx.parts, None
        dy_ = _f32c(dy)
        dx = torch.empty_like(x_)
        dps = [torch.empty_like(p_) for _ in parts]     # fresh buffers per call: returned gradient views never alias later calls
        main, streams = _SPLIT_STREAMS.begin(x_.device, len(parts))
        for (b0, b1, ws, nbytes, sd), dp, st in zip(parts, dps, streams):
            with torch.cuda.stream(st):
                Bi = b1 - b0
                gp, xp, mp, dxp = (_lib.ptr(dy_) + 4 * b0 * T * d, _lib.ptr(x_) + 4 * b0 * T * d, _lib.ptr(m_) + 4 * b0 * T,
                                   _lib.ptr(dx) + 4 * b0 * T * d)
                if isinstance(sd, _lib.DeviceSeed):  # the forward left its seed in the workspace
                    _lib.check(lib.mmt_encoder_backward_devseed(gp, xp, mp, _lib.ptr(p_), dxp, _lib.ptr(dp), _lib.ptr(ws), nbytes,
                                                                Bi, T, d, h, d_ff, n_layers, eps, dropout_p, _lib.stream_ptr()))
                else:
                    _lib.check(lib.mmt_encoder_backward(gp, xp, mp, _lib.ptr(p_), dxp, _lib.ptr(dp), _lib.ptr(ws), nbytes,
                                                        Bi, T, d, h, d_ff, n_layers, eps, dropout_p, sd, _lib.stream_ptr()))
                _lib.POOL.put(ws)
        _SPLIT_STREAMS.end(main, streams)
        if len(dps) > 1:                                # the sub-batches' parameter gradients, summed into the first buffer
            n = p_.numel()
            for other in dps[1:]:
                copy2d([_seg(dps[0], n, 1, n, src=other, src_ld=n, acc=True)])
        return dx, None, dps[0], None, None, None, None, None, None, None, None


def _seed_arg(seed):
    if isinstance(seed, (list, tuple)):
        return [_seed_arg(v) for v in seed]
    return seed if isinstance(seed, _lib.DeviceSeed) else int(seed)


def encoder_stack(x, mask, flat_params, h, d_ff, n_layers, eps=1e-6, dropout_p=0.0, seed=0, nsplit=1):
    """``seed``: a python int (by value) or a ``_lib.DeviceSeed`` (device-resident: fresh masks at every hipGraph replay), or one
    of either per sub-batch stream; ``nsplit``: run the batch as that many sub-batches on HIP streams of their own (see _SPLIT_STREAMS)."""
    return _EncoderStackFn.apply(x, mask, flat_params, int(h), int(d_ff), int(n_layers), float(eps), float(dropout_p), _seed_arg(seed),
                                 False, int(nsplit))


class _EncoderStackParamsFn(torch.autograd.Function):
    """Same stack, taking the individual parameter tensors (in the reference's registration order).  The flat buffer is
    assembled inside ``forward`` and — the point of this variant — ``backward`` hands every parameter a VIEW of ONE flat
    gradient buffer, so ``p.grad`` of all parameters alias a single allocation: data-parallel training all-reduces that
    buffer in place with one collective and no staging copies (``parallel.allreduce_gradients``)."""

    @staticmethod
    def forward(ctx, x, mask, h, d_ff, n_layers, eps, dropout_p, seed, flat, nsplit, *params):
        # `flat`: the parameters' own storage when they are views of one buffer (multiTransformer.Encoder keeps them that way), else
        # None and the buffer is assembled here (one concatenation kernel per step)
        if flat is None:
            flat = torch.cat([q.detach().reshape(-1) for q in params]).float()
        ctx.shapes = [tuple(q.shape) for q in params]
        ctx.inner = _Ctx()
        y = _EncoderStackFn.forward(ctx.inner, x, mask, flat, h, d_ff, n_layers, eps, dropout_p, seed, _needs=any(ctx.needs_input_grad),
                                    nsplit=nsplit)
        return y

    @staticmethod
    def backward(ctx, dy):
        dx, _, dflat = _EncoderStackFn.backward(ctx.inner, dy)[:3]
        grads, off = [], 0
        for shp in ctx.shapes:
            n = 1
            for v in shp:
                n *= v
            grads.append(dflat[off:off + n].view(shp))
            off += n
        return (dx, None, None, None, None, None, None, None, None, None) + tuple(grads)


class _Ctx:
    """Minimal stand-in for the autograd context when one Function drives another's static methods."""
    needs_input_grad = (False, False, False)

    def save_for_backward(self, *t):
        self.saved_tensors = t


def encoder_stack_params(x, mask, params, h, d_ff, n_layers, eps=1e-6, dropout_p=0.0, seed=0, flat=None, nsplit=1):
    return _EncoderStackParamsFn.apply(x, mask, int(h), int(d_ff), int(n_layers), float(eps), float(dropout_p), _seed_arg(seed), flat,
                                       int(nsplit), *params)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, a_2, b_2, eps):
        lib = _lib.load()
        _lib.require_hip(x, a_2, b_2)
        x_, a_, b_ = _f32c(x), _f32c(a_2), _f32c(b_2)
        d = x_.shape[-1]
        M = x_.numel() // d
        y = torch.empty_like(x_)
        stats = torch.empty(M, 2, dtype=torch.float32, device=x_.device)
        _lib.check(lib.mmt_layernorm_forward(_lib.ptr(x_), _lib.ptr(a_), _lib.ptr(b_), _lib.ptr(y), _lib.ptr(stats), M, d, eps,
                                             _lib.stream_ptr()))
        ctx.save_for_backward(x_, a_, stats)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x_, a_, stats = ctx.saved_tensors
        d = x_.shape[-1]
        M = x_.numel() // d
        dy_ = _f32c(dy)
        dx = torch.empty_like(x_)
        da = torch.empty_like(a_)
        db = torch.empty_like(a_)
        scratch = torch.empty(lib.mmt_layernorm_scratch_floats(M, d), dtype=torch.float32, device=x_.device)
        _lib.check(lib.mmt_layernorm_backward(_lib.ptr(dy_), _lib.ptr(x_), _lib.ptr(a_), _lib.ptr(stats), _lib.ptr(dx),
                                              _lib.ptr(da), _lib.ptr(db), _lib.ptr(scratch), M, d, ctx.eps, _lib.stream_ptr()))
        return dx, da, db, None


def layer_norm(x, a_2, b_2, eps=1e-6):
    return _LayerNormFn.apply(x, a_2, b_2, float(eps))


class _SdpaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, mask, h, dropout_p, seed):
        lib = _lib.load()
        _lib.require_hip(q, k, v, mask)
        q_, k_, v_, m_ = _f32c(q), _f32c(k), _f32c(v), _f32c(mask)
        B, T, d = q_.shape
        if k_.shape != q_.shape or v_.shape != q_.shape:
            raise NotImplementedError("sdpa: query, key and value must share the shape (B,T,d)")
        if m_ is not None and m_.numel() != B * T:
            raise NotImplementedError("sdpa: only the reference's query-row mask of shape (B,T,1) is supported")
        train = dropout_p > 0.0
        nbytes = (lib.mmt_sdpa_workspace_bytes if train else lib.mmt_sdpa_workspace_bytes_eval)(B, T, d, h)
        if nbytes == 0:
            _lib.check(lib.mmt_sdpa_forward(None, None, None, None, None, None, 0, B, T, d, h, 0.0, 0, None))
        ws = _lib.POOL.get(nbytes, q_.device, tag=("sdpa", B, T, d, h, train))
        out = torch.empty_like(q_)
        _lib.check(lib.mmt_sdpa_forward(_lib.ptr(q_), _lib.ptr(k_), _lib.ptr(v_), _lib.ptr(m_), _lib.ptr(out), _lib.ptr(ws), nbytes,
                                        B, T, d, h, dropout_p, seed, _lib.stream_ptr()))
        if any(ctx.needs_input_grad):
            ctx.ws, ctx.cfg, ctx.mask = ws, (B, T, d, h, nbytes, dropout_p, seed), m_
        else:
            _lib.POOL.put(ws)
        return out

    @staticmethod
    def backward(ctx, dctx):
        lib = _lib.load()
        B, T, d, h, nbytes, dropout_p, seed = ctx.cfg
        g = _f32c(dctx)
        dq, dk, dv = (torch.empty_like(g) for _ in range(3))
        _lib.check(lib.mmt_sdpa_backward(_lib.ptr(g), _lib.ptr(ctx.mask), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(ctx.ws),
                                         nbytes, B, T, d, h, dropout_p, seed, _lib.stream_ptr()))
        _lib.POOL.put(ctx.ws)
        ctx.ws = None
        return dq, dk, dv, None, None, None, None


def sdpa(q, k, v, mask, h, dropout_p=0.0, seed=0):
    """q,k,v: (B,T,d) with head i in columns [i*d/h,(i+1)*d/h); mask (B,T,1) blanks query rows; -> (B,T,d).
    dropout_p / seed: train-mode dropout on the probabilities (transformer/MFT/multiTransformer.py:32-33)."""
    return _SdpaFn.apply(q, k, v, mask, int(h), float(dropout_p), int(seed))


def _raw_linear_fwd(x_, W_, b_, act=0, r_=None, in_p=0.0, out_p=0.0, seed=0):
    """y = rowscale * drop_out(act(drop_in(x) W^T + b)) on contiguous fp32 device tensors, outside autograd -> (y, workspace, bytes)."""
    lib = _lib.load()
    K, N = x_.shape[-1], W_.shape[0]
    M = x_.numel() // K
    if W_.shape[1] != K:
        raise ValueError("linear: weight %s does not match input features %d" % (tuple(W_.shape), K))
    nbytes = lib.mmt_linear_workspace_bytes(M, K, N)
    ws = _lib.POOL.get(nbytes, x_.device, tag=("linear", M, K, N))     # zero pads survive reuse: the kernels never write them
    y = torch.empty(x_.shape[:-1] + (N,), dtype=torch.float32, device=x_.device)
    if in_p > 0.0 or out_p > 0.0:
        dev_seed = isinstance(seed, _lib.DeviceSeed)
        _lib.check(lib.mmt_linear_dropout_forward(_lib.ptr(x_), _lib.ptr(W_), _lib.ptr(b_), _lib.ptr(r_), _lib.ptr(y), _lib.ptr(ws), nbytes,
                                                  M, K, N, act, in_p, out_p, 0 if dev_seed else int(seed),
                                                  seed.ptr() if dev_seed else None, _lib.stream_ptr()))
    else:
        _lib.check(lib.mmt_linear_forward(_lib.ptr(x_), _lib.ptr(W_), _lib.ptr(b_), _lib.ptr(r_), _lib.ptr(y), _lib.ptr(ws), nbytes,
                                          M, K, N, act, _lib.stream_ptr()))
    return y, ws, nbytes


def _raw_linear_bwd(g, x_, W_, y_, r_, ws, nbytes, need_x, need_w, need_b, act=0, in_p=0.0, out_p=0.0, seed=0):
    """-> (dx, dW, db) of the affine map above (each None unless asked for); hands the workspace back to the pool."""
    lib = _lib.load()
    K, N = x_.shape[-1], W_.shape[0]
    M = x_.numel() // K
    dx = torch.empty_like(x_) if need_x else None
    dW = torch.empty_like(W_) if need_w else None
    db = torch.empty(N, dtype=torch.float32, device=x_.device) if need_b else None
    if in_p > 0.0 or out_p > 0.0:
        dev_seed = isinstance(seed, _lib.DeviceSeed)
        _lib.check(lib.mmt_linear_dropout_backward(_lib.ptr(g), _lib.ptr(x_), _lib.ptr(W_), _lib.ptr(y_), _lib.ptr(r_), _lib.ptr(dx),
                                                   _lib.ptr(dW), _lib.ptr(db), _lib.ptr(ws), nbytes, M, K, N, act, in_p, out_p,
                                                   0 if dev_seed else int(seed), 1 if dev_seed else 0, _lib.stream_ptr()))
    else:
        _lib.check(lib.mmt_linear_backward(_lib.ptr(g), _lib.ptr(x_), _lib.ptr(W_), _lib.ptr(y_), _lib.ptr(r_), _lib.ptr(dx),
                                           _lib.ptr(dW), _lib.ptr(db), _lib.ptr(ws), nbytes, M, K, N, act, _lib.stream_ptr()))
    _lib.POOL.put(ws)
    return dx, dW, db


class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, rowscale, act, in_p, out_p, seed):
        _lib.require_hip(x, W, b, rowscale)
        x_, W_, b_, r_ = _f32c(x), _f32c(W), _f32c(b), _f32c(rowscale)
        if r_ is not None and r_.numel() != x_.numel() // x_.shape[-1]:
            raise ValueError("linear: rowscale must have one entry per row")
        y, ws, nbytes = _raw_linear_fwd(x_, W_, b_, act, r_, in_p, out_p, seed)
        ctx.cfg = (act, nbytes, b is not None, in_p, out_p, seed)
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x_, W_, y if act != 0 else None, r_)
            ctx.ws = ws
        else:
            _lib.POOL.put(ws)
        return y

    @staticmethod
    def backward(ctx, dy):
        x_, W_, y_, r_ = ctx.saved_tensors
        act, nbytes, has_b, in_p, out_p, seed = ctx.cfg
        dx, dW, db = _raw_linear_bwd(_f32c(dy), x_, W_, y_, r_, ctx.ws, nbytes, ctx.needs_input_grad[0], ctx.needs_input_grad[1],
                                     has_b and ctx.needs_input_grad[2], act, in_p, out_p, seed)
        ctx.ws = None
        return dx, dW, db, None, None, None, None, None


def linear(x, weight, bias=None, act=0, rowscale=None, in_dropout=0.0, out_dropout=0.0, seed=0):
    """y = rowscale * drop_out(act(drop_in(x) W^T + b)); act: 0 none, 1 ReLU, 2 tanh, 3 sigmoid.  in_dropout / out_dropout: train-mode
    dropout probabilities fused into the kernel (out_dropout behind ReLU only); seed: python int or ``_lib.DeviceSeed``."""
    K = x.shape[-1]
    if K % 4:
        # the row kernels stage their A tile in 16-byte pieces of fp32: an input width that is no multiple of 4 (none occurs in the reference's
        # models) is brought to the next one with zero columns on x and on W — the product is unchanged, the gradients of the pads are dropped
        # by cat_cols' backward (an input-dropout mask then indexes the padded width, like every mask indexes padded leading dimensions)
        pad = 4 - K % 4
        x = cat_cols([x, torch.zeros(*x.shape[:-1], pad, dtype=x.dtype, device=x.device)])
        weight = cat_cols([weight, torch.zeros(weight.shape[0], pad, dtype=weight.dtype, device=weight.device)])
    return _LinearFn.apply(x, weight, bias, rowscale, int(act), float(in_dropout), float(out_dropout),
                           seed if isinstance(seed, _lib.DeviceSeed) else int(seed))


class _HighwayFn(torch.autograd.Function):
    """The Highway layer of the window encoder with the front-end's Dropout(0.3) behind it, as ONE autograd node:
        out = drop(gate * proj + (1 - gate) * x),  proj = x Wp^T + bp,  gate = sigmoid(x Wg^T + bg)
    (transformer/SFT/models.py:27-55,132-134).  One node, so that the three gradient paths into x (direct, through proj, through gate) are
    summed by a copy2d launch instead of by autograd's library adds."""

    @staticmethod
    def forward(ctx, x, Wp, bp, Wg, bg, p, seed):
        lib = _lib.load()
        _lib.require_hip(x, Wp, bp, Wg, bg)
        x_, Wp_, bp_, Wg_, bg_ = _f32c(x), _f32c(Wp), _f32c(bp), _f32c(Wg), _f32c(bg)
        proj, wsp, nbp = _raw_linear_fwd(x_, Wp_, bp_)
        gate, wsg, nbg = _raw_linear_fwd(x_, Wg_, bg_, act=3)
        out = torch.empty_like(x_)
        dev_seed = isinstance(seed, _lib.DeviceSeed) and p > 0.0
        block = torch.empty(2, dtype=torch.int64, device=x_.device) if dev_seed else None       # seed + stream keys of this call, for the backward
        _lib.check(lib.mmt_highway_forward(_lib.ptr(x_), _lib.ptr(proj), _lib.ptr(gate), _lib.ptr(out), x_.numel(), p,
                                           0 if dev_seed else int(seed), seed.ptr() if dev_seed else None, _lib.ptr(block), _lib.stream_ptr()))
        if any(ctx.needs_input_grad):
            ctx.save_for_backward(x_, Wp_, Wg_, proj, gate, block)
            ctx.ws = (wsp, nbp, wsg, nbg)
        else:
            _lib.POOL.put(wsp); _lib.POOL.put(wsg)
        ctx.cfg = (p, 0 if dev_seed else int(seed))
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x_, Wp_, Wg_, proj, gate, block = ctx.saved_tensors
        p, seed = ctx.cfg
        wsp, nbp, wsg, nbg = ctx.ws
        ctx.ws = None
        d_ = _f32c(dout)
        dx, dproj, dgate = torch.empty_like(x_), torch.empty_like(x_), torch.empty_like(x_)
        _lib.check(lib.mmt_highway_backward(_lib.ptr(d_), _lib.ptr(x_), _lib.ptr(proj), _lib.ptr(gate), _lib.ptr(dx), _lib.ptr(dproj), _lib.ptr(dgate),
                                            x_.numel(), p, seed, _lib.ptr(block), _lib.stream_ptr()))
        need_x = ctx.needs_input_grad[0]
        dx1, dWp, dbp = _raw_linear_bwd(dproj, x_, Wp_, None, None, wsp, nbp, need_x, ctx.needs_input_grad[1], ctx.needs_input_grad[2])
        dx2, dWg, dbg = _raw_linear_bwd(dgate, x_, Wg_, gate, None, wsg, nbg, need_x, ctx.needs_input_grad[3], ctx.needs_input_grad[4], act=3)
        if need_x:
            K = x_.shape[-1]
            copy2d([_seg(dx, K, x_.numel() // K, K, src=dx1, src_ld=K, src2=dx2, src2_ld=K, acc=True)])        # dx += dx1 + dx2
        return (dx if need_x else None), dWp, dbp, dWg, dbg, None, None


def highway(x, Wp, bp, Wg, bg, dropout_p=0.0, seed=0):
    """drop(gate * proj + (1 - gate) * x) with proj = x Wp^T + bp, gate = sigmoid(x Wg^T + bg); seed: python int or ``_lib.DeviceSeed``
    (train-mode dropout: stream 3000 of dropout_mask, index = element)."""
    return _HighwayFn.apply(x, Wp, bp, Wg, bg, float(dropout_p), seed if isinstance(seed, _lib.DeviceSeed) else int(seed))


class _LstmScanFn(torch.autograd.Function):
    """h_all, c_all = scan(gx, W_rec, h0, c0): the recurrent half of an LSTM layer.

    gx (T,B,4H) already holds x_t W_ih^T + b_ih + b_hh.  Replaces the nn.LSTMCell loop of MFN
    (transformer/MFT/multiTransformer.py:200-208) and the nn.LSTM step loop of the SFT decoder
    (transformer/SFT/multiTransformer.py:471-476)."""

    @staticmethod
    def forward(ctx, gx, W_rec, h0, c0):
        lib = _lib.load()
        ctx.set_materialize_grads(False)            # an unused output (the decoder never reads c_all) arrives as None, not as a zero fill
        _lib.require_hip(gx, W_rec, h0, c0)
        gx_, W_, h0_, c0_ = _f32c(gx), _f32c(W_rec), _f32c(h0), _f32c(c0)
        T, B, H4 = gx_.shape
        H = H4 // 4
        if W_.shape != (4 * H, H):
            raise ValueError("lstm_scan: W_rec must be (4H,H) = (%d,%d), got %s" % (4 * H, H, tuple(W_.shape)))
        nbytes = lib.mmt_lstm_scan_workspace_bytes(H)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=gx_.device)
        h_all = torch.empty(T, B, H, dtype=torch.float32, device=gx_.device)
        c_all = torch.empty_like(h_all)
        acts = torch.empty(T, B, 4 * H, dtype=torch.float32, device=gx_.device)
        _lib.check(lib.mmt_lstm_scan_forward(_lib.ptr(gx_), _lib.ptr(W_), _lib.ptr(h0_), _lib.ptr(c0_), _lib.ptr(h_all), _lib.ptr(c_all),
                                             _lib.ptr(acts), _lib.ptr(ws), nbytes, T, B, H, _lib.stream_ptr()))
        if H > 128:                                 # four-CU scan: its exchange waits are bounded and report through this word
            _lib.ERRORS.watch(ws[:4].view(torch.int32), "lstm_scan forward (T=%d, B=%d, H=%d)" % (T, B, H))
        ctx.save_for_backward(W_, h0_, c0_, h_all, c_all, acts)
        ctx.dims = (T, B, H, nbytes)
        return h_all, c_all

    @staticmethod
    def backward(ctx, dh_all, dc_all):
        lib = _lib.load()
        W_, h0_, c0_, h_all, c_all, acts = ctx.saved_tensors
        T, B, H, nbytes = ctx.dims
        dev = h_all.device
        dh_, dc_ = _f32c(dh_all), _f32c(dc_all)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        dgx = torch.empty(T, B, 4 * H, dtype=torch.float32, device=dev)
        dh0 = torch.empty(B, H, dtype=torch.float32, device=dev)
        dc0 = torch.empty(B, H, dtype=torch.float32, device=dev)
        _lib.check(lib.mmt_lstm_scan_backward(_lib.ptr(dh_), _lib.ptr(dc_), _lib.ptr(W_), _lib.ptr(c0_), _lib.ptr(c_all), _lib.ptr(acts),
                                              _lib.ptr(dgx), _lib.ptr(dh0), _lib.ptr(dc0), _lib.ptr(ws), nbytes, T, B, H, _lib.stream_ptr()))
        if H > 128:
            _lib.ERRORS.watch(ws[:4].view(torch.int32), "lstm_scan backward (T=%d, B=%d, H=%d)" % (T, B, H))
        dW = None
        if ctx.needs_input_grad[1]:
            # dW_rec = sum_{t,b} dG[t,b,:]^T h_{t-1}[b,:]  — a window-contraction: the weight-gradient GEMM
            hprev = torch.empty_like(h_all)          # h_{t-1}: h0 (or zeros), then h_all shifted by one step
            copy2d([_seg(hprev, H, B, H, src=h0_, src_ld=H), _seg(hprev, H, (T - 1) * B, H, src=h_all, src_ld=H, dst_off=B * H)])
            M = T * B
            lb = lib.mmt_linear_workspace_bytes(M, H, 4 * H)
            lws = _lib.POOL.get(lb, dev, tag=("linear", M, H, 4 * H))
            dW = torch.empty_like(W_)
            _lib.check(lib.mmt_linear_backward(_lib.ptr(dgx), _lib.ptr(hprev), _lib.ptr(W_), None, None, None, _lib.ptr(dW), None,
                                               _lib.ptr(lws), lb, M, H, 4 * H, 0, _lib.stream_ptr()))
            _lib.POOL.put(lws)
        return (dgx, dW, dh0 if (h0_ is not None and ctx.needs_input_grad[2]) else None,
                dc0 if (c0_ is not None and ctx.needs_input_grad[3]) else None)


def lstm_scan(gx, W_rec, h0=None, c0=None):
    return _LstmScanFn.apply(gx, W_rec, h0, c0)


class _ConvPoolFn(torch.autograd.Function):
    """out = max over positions of Conv1d(D -> F, kernel 2)(window) + bias — the reference's CNN.forward
    (transformer/SFT/models.py:57-79).  x (N,W,D) is input data: no gradient flows to it."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        lib = _lib.load()
        _lib.require_hip(x, weight, bias)
        if x.requires_grad:
            raise NotImplementedError("conv_maxpool: the windows are input data; a gradient w.r.t. them is not implemented "
                                      "(the reference never asks for one)")
        x_, w_, b_ = _f32c(x), _f32c(weight), _f32c(bias)
        N, W, D = x_.shape
        F_ = w_.shape[0]
        if w_.shape != (F_, D, 2):
            raise NotImplementedError("conv_maxpool: weight must be (F, D, 2) = nn.Conv1d(D, F, kernel_size=2).weight, got %s"
                                      % (tuple(w_.shape),))
        nbytes = lib.mmt_convpool_workspace_bytes(N, W, D, F_)
        if nbytes == 0:
            _lib.check(lib.mmt_convpool_forward(None, None, None, None, None, None, 0, N, W, D, F_, None))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x_.device)
        out = torch.empty(N, F_, dtype=torch.float32, device=x_.device)
        arg = torch.empty(N, F_, dtype=torch.int32, device=x_.device)
        _lib.check(lib.mmt_convpool_forward(_lib.ptr(x_), _lib.ptr(w_), _lib.ptr(b_), _lib.ptr(out), _lib.ptr(arg), _lib.ptr(ws), nbytes,
                                            N, W, D, F_, _lib.stream_ptr()))
        ctx.save_for_backward(x_, arg)
        ctx.dims = (N, W, D, F_, nbytes)
        ctx.mark_non_differentiable(arg)
        ctx.set_materialize_grads(False)        # no zeros tensor for the index output's "gradient" (that would be a library fill kernel)
        return out, arg

    @staticmethod
    def backward(ctx, dout, _darg):
        lib = _lib.load()
        x_, arg = ctx.saved_tensors
        N, W, D, F_, nbytes = ctx.dims
        if dout is None:
            return None, None, None
        g = _f32c(dout)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=g.device)
        dw = torch.empty(F_, D, 2, dtype=torch.float32, device=g.device)
        db = torch.empty(F_, dtype=torch.float32, device=g.device)
        _lib.check(lib.mmt_convpool_backward(_lib.ptr(x_), _lib.ptr(g), _lib.ptr(arg), _lib.ptr(dw), _lib.ptr(db), _lib.ptr(ws), nbytes,
                                             N, W, D, F_, _lib.stream_ptr()))
        return None, dw, db


def conv_maxpool(x, weight, bias):
    """x (N,W,D) windows, weight (F,D,2), bias (F) -> (out (N,F), argmax positions (N,F) int32)."""
    return _ConvPoolFn.apply(x, weight, bias)


class _MfnMemScanFn(torch.autograd.Function):
    """mem_all = scan(apre, chat, Wm, W2, b2): the MFN memory recurrence
    (transformer/MFT/multiTransformer.py:221-224) with everything that does not depend on mem batched before."""

    @staticmethod
    def forward(ctx, apre, chat, Wm, W2, b2, dropout_p, seed):
        lib = _lib.load()
        _lib.require_hip(apre, chat, Wm, W2, b2)
        a_, c_, Wm_, W2_, b2_ = _f32c(apre), _f32c(chat), _f32c(Wm), _f32c(W2), _f32c(b2)
        T, B, U = a_.shape
        MD, HG = c_.shape[-1], W2_.shape[-1]
        if U != 2 * HG or Wm_.shape != (U, MD) or W2_.shape != (2, MD, HG) or b2_.shape != (2, MD):
            raise ValueError("mfn_mem_scan: inconsistent shapes")
        dev = a_.device
        nbytes = lib.mmt_mfn_mem_scan_workspace_bytes()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        mem_all = torch.empty(T, B, MD, dtype=torch.float32, device=dev)
        u_all = torch.empty(T, B, U, dtype=torch.float32, device=dev)
        g_all = torch.empty(T, B, 2 * MD, dtype=torch.float32, device=dev)
        if isinstance(seed, _lib.DeviceSeed):
            _lib.check(lib.mmt_mfn_mem_scan_forward_devseed(_lib.ptr(a_), _lib.ptr(c_), _lib.ptr(Wm_), _lib.ptr(W2_), _lib.ptr(b2_), _lib.ptr(mem_all),
                                                            _lib.ptr(u_all), _lib.ptr(g_all), _lib.ptr(ws), nbytes, T, B, MD, HG,
                                                            dropout_p, seed.ptr(), _lib.stream_ptr()))
        else:
            _lib.check(lib.mmt_mfn_mem_scan_forward(_lib.ptr(a_), _lib.ptr(c_), _lib.ptr(Wm_), _lib.ptr(W2_), _lib.ptr(b2_), _lib.ptr(mem_all),
                                                    _lib.ptr(u_all), _lib.ptr(g_all), _lib.ptr(ws), nbytes, T, B, MD, HG,
                                                    dropout_p, seed, _lib.stream_ptr()))
        ctx.save_for_backward(c_, Wm_, W2_, mem_all, u_all, g_all)
        ctx.dims = (T, B, U, MD, HG, nbytes)
        ctx.dropout_p = dropout_p
        return mem_all

    @staticmethod
    def backward(ctx, dmem):
        lib = _lib.load()
        c_, Wm_, W2_, mem_all, u_all, g_all = ctx.saved_tensors
        T, B, U, MD, HG, nbytes = ctx.dims
        dev = c_.device
        dm = _f32c(dmem)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dchat = torch.empty_like(c_)
        dapre = torch.empty_like(u_all)
        dz = torch.empty_like(g_all)
        _lib.check(lib.mmt_mfn_mem_scan_backward(_lib.ptr(dm), _lib.ptr(c_), _lib.ptr(mem_all), _lib.ptr(u_all), _lib.ptr(g_all), _lib.ptr(Wm_),
                                                 _lib.ptr(W2_), _lib.ptr(dchat), _lib.ptr(dapre), _lib.ptr(dz), _lib.ptr(ws), nbytes,
                                                 T, B, MD, HG, ctx.dropout_p, _lib.stream_ptr()))
        M = T * B
        st = _lib.stream_ptr()
        # batched weight gradients (window contractions): dWm = dapre^T mem_prev ; dW2_g = dz_g^T u_g ; db2 = sum dz
        mem_prev = torch.empty_like(mem_all)
        dzs = [torch.empty(T, B, MD, dtype=torch.float32, device=dev) for _ in range(2)]
        us = [torch.empty(T, B, HG, dtype=torch.float32, device=dev) for _ in range(2)]
        segs = [_seg(mem_prev, MD, B, MD), _seg(mem_prev, MD, (T - 1) * B, MD, src=mem_all, src_ld=MD, dst_off=B * MD)]
        for g in range(2):
            segs += [_seg(dzs[g], MD, M, MD, src=dz, src_ld=2 * MD, src_off=g * MD), _seg(us[g], HG, M, HG, src=u_all, src_ld=2 * HG, src_off=g * HG)]
        copy2d(segs)
        dWm = torch.empty_like(Wm_)
        lb = lib.mmt_linear_workspace_bytes(M, MD, U)
        lws = _lib.POOL.get(lb, dev, tag=("linear", M, MD, U))
        _lib.check(lib.mmt_linear_backward(_lib.ptr(dapre), _lib.ptr(mem_prev), _lib.ptr(Wm_), None, None, None, _lib.ptr(dWm), None,
                                           _lib.ptr(lws), lb, M, MD, U, 0, st))
        _lib.POOL.put(lws)
        dW2 = torch.empty_like(W2_)
        db2 = torch.empty(2, MD, dtype=torch.float32, device=dev)
        lb2 = lib.mmt_linear_workspace_bytes(M, HG, MD)
        for g in range(2):
            dzg, ug = dzs[g], us[g]
            lws2 = _lib.POOL.get(lb2, dev, tag=("linear", M, HG, MD))
            _lib.check(lib.mmt_linear_backward(_lib.ptr(dzg), _lib.ptr(ug), _lib.ptr(W2_[g]), None, None, None, _lib.ptr(dW2[g]),
                                               _lib.ptr(db2[g]), _lib.ptr(lws2), lb2, M, HG, MD, 0, st))
            _lib.POOL.put(lws2)
        return dapre, dchat, dWm, dW2, db2, None, None


def mfn_mem_scan(apre, chat, Wm, W2, b2, dropout_p=0.0, seed=0):
    return _MfnMemScanFn.apply(apre, chat, Wm, W2, b2, float(dropout_p), _seed_arg(seed))


class _MseSumLossFn(torch.autograd.Function):
    """loss = sum((pred - target)^2) / denom and its gradient in one pass — the reference's per-batch loss
    (transformer/SFT/train.py:133-137: MSELoss(reduction='sum') divided by sum(lengths))."""

    @staticmethod
    def forward(ctx, pred, target, denom):
        lib = _lib.load()
        _lib.require_hip(pred, target)
        p_, t_ = _f32c16(pred), _f32c16(target)
        if p_.shape != t_.shape:
            raise ValueError("mse_sum_loss: pred %s and target %s differ in shape" % (tuple(pred.shape), tuple(target.shape)))
        n = p_.numel()
        loss = torch.empty((), dtype=torch.float32, device=p_.device)
        dpred = torch.empty_like(p_)
        scratch = torch.empty(lib.mmt_mse_sum_scratch_doubles(n), dtype=torch.float64, device=p_.device)
        _lib.check(lib.mmt_mse_sum_forward(_lib.ptr(p_), _lib.ptr(t_), 1.0 / float(denom), _lib.ptr(loss), _lib.ptr(dpred), _lib.ptr(scratch),
                                           n, _lib.stream_ptr()))
        ctx.save_for_backward(dpred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g, None, None


def mse_sum_loss(pred, target, denom):
    """sum((pred - target)^2) / denom with the gradient produced in the same pass (train-step semantics of the reference)."""
    return _MseSumLossFn.apply(pred, target, float(denom))


def mse_sum_loss_backward(pred, target, denom):
    """``loss = mse_sum_loss(pred, target, denom); loss.backward()`` (transformer/SFT/train.py:133-139) without the two kernels autograd
    spends on the seed gradient (a fill with 1.0 and a multiplication by it): the loss kernel's gradient 2 (pred - target) / denom is
    handed straight to ``pred.backward``.  Returns the detached loss."""
    lib = _lib.load()
    _lib.require_hip(pred, target)
    p_, t_ = _f32c16(pred), _f32c16(target)
    if p_.shape != t_.shape:
        raise ValueError("mse_sum_loss: pred %s and target %s differ in shape" % (tuple(pred.shape), tuple(target.shape)))
    n = p_.numel()
    loss = torch.empty((), dtype=torch.float32, device=p_.device)
    dpred = torch.empty_like(p_)
    scratch = torch.empty(lib.mmt_mse_sum_scratch_doubles(n), dtype=torch.float64, device=p_.device)
    _lib.check(lib.mmt_mse_sum_forward(_lib.ptr(p_), _lib.ptr(t_), 1.0 / float(denom), _lib.ptr(loss), _lib.ptr(dpred), _lib.ptr(scratch),
                                       n, _lib.stream_ptr()))
    pred.backward(dpred.view(pred.shape))
    return loss


# ---------------------------------------------------------------------------------------------------------------------------
# Data movement between the kernels (csrc/glue.h, mmt_copy2d): everything the reference does with torch.cat / stack / permute /
# slicing / broadcasting around the MFN gate and the SFT decoder, and the autograd twins of those ops, as strided 2-D copies in
# hand-written kernels — several per launch.
def _seg(dst, dst_ld, rows, cols, src=None, src_ld=0, src2=None, src2_ld=0, rowscale=None, perm=0, B=0, T=0, acc=False,
         dst_off=0, src_off=0, src2_off=0):
    """One copy segment on fp32 device tensors; *_off are ELEMENT offsets into the tensors' storage views."""
    def at(t, off):
        return None if t is None else t.data_ptr() + 4 * int(off)
    return _lib.CopySeg(at(src, src_off), at(src2, src2_off), at(dst, dst_off), at(rowscale, 0), int(rows), int(cols), int(src_ld),
                        int(src2_ld), int(dst_ld), int(perm), int(B), int(T), 1 if acc else 0)


def copy2d(segs):
    """Run a list of ``_seg`` copies (destinations must not overlap) — 24 per kernel launch."""
    import ctypes
    segs = [g for g in segs if g.rows > 0 and g.cols > 0]
    if not segs:
        return
    arr = (_lib.CopySeg * len(segs))(*segs)
    _lib.check(_lib.load().mmt_copy2d(ctypes.cast(arr, ctypes.c_void_p), len(segs), _lib.stream_ptr()))


def _new(*shape, like):
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


class _TimeMajorFn(torch.autograd.Function):
    """(B,T,d) -> (T,B,d) contiguous (the reference's ``permute(1,0,2)``, transformer/MFT/multiTransformer.py:300) as ONE copy."""

    @staticmethod
    def forward(ctx, x):
        _lib.require_hip(x)
        x_ = _f32c(x)
        B, T, d = x_.shape
        y = _new(T, B, d, like=x_)
        copy2d([_seg(y, d, B * T, d, src=x_, src_ld=d, perm=1, B=B, T=T)])
        return y

    @staticmethod
    def backward(ctx, dy):
        return batch_major(dy)


class _BatchMajorFn(torch.autograd.Function):
    """(T,B,d) -> (B,T,d) contiguous, optionally times a per-window scale (the reference's ``* mask.float()``, :310)."""

    @staticmethod
    def forward(ctx, x, rowscale):
        _lib.require_hip(x, rowscale)
        x_, r_ = _f32c(x), _f32c(rowscale)
        T, B, d = x_.shape
        y = _new(B, T, d, like=x_)
        copy2d([_seg(y, d, B * T, d, src=x_, src_ld=d, perm=2, B=B, T=T, rowscale=r_)])
        ctx.r = r_
        return y

    @staticmethod
    def backward(ctx, dy):
        g = _f32c(dy)
        B, T, d = g.shape
        dx = _new(T, B, d, like=g)
        copy2d([_seg(dx, d, B * T, d, src=g, src_ld=d, perm=1, B=B, T=T, rowscale=ctx.r)])
        return dx, None


def time_major(x):
    return _TimeMajorFn.apply(x)


def batch_major(x, rowscale=None):
    """(T,B,d) -> (B,T,d); rowscale: (B,T[,1]) factors per window of the OUTPUT (e.g. the mask)."""
    return _BatchMajorFn.apply(x, rowscale)


class _DecoderPackFn(torch.autograd.Function):
    """Operands of the SFT decoder's recurrence from nn.LSTM's parameters (transformer/SFT/multiTransformer.py:463-476; step t feeds
    [o_{t-1}; enc_t] with o = h):  Wx = W_ih[:, d:] (multiplies enc_t, batched over T),  W_rec = W_ih[:, :d] + W_hh (multiplies h_{t-1}),
    a copy of W_hh (multiplies h0 in step 0, where o_{-1} = 0) and bias = b_ih + b_hh.  ONE node, so that the gradients of the several
    uses of W_ih and W_hh are combined here by the copy kernel and not by autograd's accumulation."""

    @staticmethod
    def forward(ctx, W_ih, W_hh, b_ih, b_hh):
        _lib.require_hip(W_ih, W_hh, b_ih, b_hh)
        Wi, Wh, bi, bh = _f32c(W_ih), _f32c(W_hh), _f32c(b_ih), _f32c(b_hh)
        G, d = Wh.shape
        if Wi.shape != (G, 2 * d):
            raise ValueError("decoder_pack: weight_ih must be (4d, 2d)")
        Wx, Wrec, Whh, bias = _new(G, d, like=Wi), _new(G, d, like=Wi), _new(G, d, like=Wi), _new(G, like=Wi)
        copy2d([_seg(Wx, d, G, d, src=Wi, src_ld=2 * d, src_off=d),
                _seg(Wrec, d, G, d, src=Wi, src_ld=2 * d, src2=Wh, src2_ld=d),
                _seg(Whh, d, G, d, src=Wh, src_ld=d),
                _seg(bias, G, 1, G, src=bi, src_ld=G, src2=bh, src2_ld=G)])
        ctx.dims = (G, d)
        return Wx, Wrec, Whh, bias

    @staticmethod
    def backward(ctx, dWx, dWrec, dWhh, dbias):
        G, d = ctx.dims
        like = next(t for t in (dWx, dWrec, dWhh, dbias) if t is not None)

        def z(t, *shape):
            return _f32c(t) if t is not None else torch.zeros(*shape, dtype=torch.float32, device=like.device)
        dWx, dWrec, dWhh, dbias = z(dWx, G, d), z(dWrec, G, d), z(dWhh, G, d), z(dbias, G)
        dWi, dWh = _new(G, 2 * d, like=like), _new(G, d, like=like)
        copy2d([_seg(dWi, 2 * d, G, d, src=dWrec, src_ld=d), _seg(dWi, 2 * d, G, d, src=dWx, src_ld=d, dst_off=d),
                _seg(dWh, d, G, d, src=dWrec, src_ld=d, src2=dWhh, src2_ld=d)])
        return dWi, dWh, dbias, dbias


def decoder_pack(W_ih, W_hh, b_ih, b_hh):
    return _DecoderPackFn.apply(W_ih, W_hh, b_ih, b_hh)


class _Add2Fn(torch.autograd.Function):
    """a + b for two tensors of one shape (bias_ih + bias_hh, W_ih[:, :d] + W_hh); the gradient passes to both unchanged."""

    @staticmethod
    def forward(ctx, a, b):
        _lib.require_hip(a, b)
        a_, b_ = _f32c(a), _f32c(b)
        if a_.shape != b_.shape:
            raise ValueError("add2: shapes differ")
        y = torch.empty_like(a_)
        n = a_.shape[-1] if a_.dim() > 1 else a_.numel()
        copy2d([_seg(y, n, a_.numel() // n, n, src=a_, src_ld=n, src2=b_, src2_ld=n)])
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add2(a, b):
    return _Add2Fn.apply(a, b)


class _AddRow0Fn(torch.autograd.Function):
    """gx[0] += row for a (T,B,n) tensor and a (1,n) row broadcast over the batch, IN PLACE (the reference's first decoder step sees
    h0 W_hh^T, transformer/SFT/multiTransformer.py:466-471); backward: the row's gradient is the column sum of dgx[0]."""

    @staticmethod
    def forward(ctx, gx, row):
        _lib.require_hip(gx, row)
        T, B, n = gx.shape
        if not gx.is_contiguous() or gx.dtype != torch.float32:
            raise ValueError("add_row0: a contiguous fp32 (T,B,n) tensor is modified in place")
        r_ = _f32c(row).reshape(-1)
        copy2d([_seg(gx, n, B, n, src=r_, src_ld=0, acc=True)])
        ctx.mark_dirty(gx)
        ctx.dims = (B, n, tuple(row.shape))
        return gx

    @staticmethod
    def backward(ctx, dy):
        B, n, rshape = ctx.dims
        g = _f32c(dy)
        drow = _new(n, like=g)
        _lib.check(_lib.load().mmt_colsum(_lib.ptr(g), _lib.ptr(drow), B, n, n, _lib.stream_ptr()))
        return dy, drow.view(rshape)


def add_row0(gx, row):
    return _AddRow0Fn.apply(gx, row)


class _BroadcastRowsFn(torch.autograd.Function):
    """(1,n) parameter row -> (B,n) contiguous (``dec_c0[0].expand(B, d)``); backward: column sum."""

    @staticmethod
    def forward(ctx, row, B):
        _lib.require_hip(row)
        r_ = _f32c(row).reshape(-1)
        n = r_.numel()
        y = _new(B, n, like=r_)
        copy2d([_seg(y, n, B, n, src=r_, src_ld=0)])
        ctx.dims = (B, n, tuple(row.shape))
        return y

    @staticmethod
    def backward(ctx, dy):
        B, n, rshape = ctx.dims
        g = _f32c(dy)
        drow = _new(n, like=g)
        _lib.check(_lib.load().mmt_colsum(_lib.ptr(g), _lib.ptr(drow), B, n, n, _lib.stream_ptr()))
        return drow.view(rshape), None


def broadcast_rows(row, B):
    return _BroadcastRowsFn.apply(row, int(B))


class _CatColsFn(torch.autograd.Function):
    """torch.cat(tensors, dim=-1) of tensors that share their leading dimensions; backward: the column split."""

    @staticmethod
    def forward(ctx, *ts):
        _lib.require_hip(*ts)
        ts_ = [_f32c(t) for t in ts]
        widths = [t.shape[-1] for t in ts_]
        rows = ts_[0].numel() // widths[0]
        W = sum(widths)
        y = _new(*ts_[0].shape[:-1], W, like=ts_[0])
        segs, col = [], 0
        for t, w in zip(ts_, widths):
            if t.numel() // w != rows:
                raise ValueError("cat_cols: leading dimensions differ")
            segs.append(_seg(y, W, rows, w, src=t, src_ld=w, dst_off=col))
            col += w
        copy2d(segs)
        ctx.widths, ctx.shapes = widths, [tuple(t.shape) for t in ts_]
        return y

    @staticmethod
    def backward(ctx, dy):
        g = _f32c(dy)
        W = sum(ctx.widths)
        rows = g.numel() // W
        outs, segs, col = [], [], 0
        for w, shp in zip(ctx.widths, ctx.shapes):
            o = _new(*shp, like=g)
            segs.append(_seg(o, w, rows, w, src=g, src_ld=W, src_off=col))
            outs.append(o)
            col += w
        copy2d(segs)
        return tuple(outs)


def cat_cols(tensors):
    return _CatColsFn.apply(*tensors)


class _MfnGateFn(torch.autograd.Function):
    """Everything of MFN.forward behind the per-modality LSTM scans (transformer/MFT/multiTransformer.py:212-247) as ONE autograd node:
    cStar assembly (one-step shift of c + concatenation), att1 MLP, softmax * cStar, att2 MLP, the `attended` part of both gamma fc1
    layers, the memory scan, [h ; mem] and the read-out MLP with its dropout.  Forward and backward call the C entry points in
    sequence: no library kernel runs, nothing is concatenated or sliced by torch, and gradients that meet (cStar and `attended` feed two
    consumers each) are added by the copy kernel instead of by autograd."""
    NP = 20     # parameter tensors, in this order: att1_fc1.{w,b} att1_fc2 att2_fc1 att2_fc2 gamma1_fc1 gamma1_fc2 gamma2_fc1 gamma2_fc2 out_fc1 out_fc2

    @staticmethod
    def forward(ctx, nm, pg, seed_g, p_out, seed_out, *ts):
        lib = _lib.load()
        _lib.require_hip(*ts)
        hs = [_f32c(t) for t in ts[:nm]]
        cs = [_f32c(t) for t in ts[nm:2 * nm]]
        P = [_f32c(t) for t in ts[2 * nm:]]
        (a11w, a11b, a12w, a12b, a21w, a21b, a22w, a22b, g11w, g11b, g12w, g12b, g21w, g21b, g22w, g22b, o1w, o1b, o2w, o2b) = P
        T, B = hs[0].shape[0], hs[0].shape[1]
        M = T * B
        Hs = [h.shape[2] for h in hs]
        SH, MD, HG = sum(Hs), g12w.shape[0], g12w.shape[1]
        A = 2 * SH
        like = hs[0]
        # cStar = [c_{t-1} of every modality ; c_t of every modality]   (:212-217)
        c_star = _new(M, A, like=like)
        segs, col = [], 0
        for c, H in zip(cs, Hs):
            segs += [_seg(c_star, A, B, H, dst_off=col), _seg(c_star, A, M - B, H, src=c, src_ld=H, dst_off=B * A + col)]
            col += H
        for c, H in zip(cs, Hs):
            segs.append(_seg(c_star, A, M, H, src=c, src_ld=H, dst_off=col))
            col += H
        copy2d(segs)
        a1, ws_a11, nb_a11 = _raw_linear_fwd(c_star, a11w, a11b, 1)
        logits, ws_a12, nb_a12 = _raw_linear_fwd(a1, a12w, a12b, 0)
        att, attended = torch.empty_like(logits), torch.empty_like(logits)
        _lib.check(lib.mmt_softmax_mul_forward(_lib.ptr(logits), _lib.ptr(c_star), _lib.ptr(att), _lib.ptr(attended), M, A, _lib.stream_ptr()))
        a2, ws_a21, nb_a21 = _raw_linear_fwd(attended, a21w, a21b, 1)
        c_hat, ws_a22, nb_a22 = _raw_linear_fwd(a2, a22w, a22b, 2)
        # gamma fc1 = [attended part | memory part] of `both` (:221-223), gamma1 rows above gamma2 rows
        Wa, Wm, b1 = _new(2 * HG, A, like=like), _new(2 * HG, MD, like=like), _new(2 * HG, like=like)
        W2, b2 = _new(2, MD, HG, like=like), _new(2, MD, like=like)
        segs = []
        for i, (w, b, w2, bb2) in enumerate(((g11w, g11b, g12w, g12b), (g21w, g21b, g22w, g22b))):
            segs += [_seg(Wa, A, HG, A, src=w, src_ld=A + MD, dst_off=i * HG * A),
                     _seg(Wm, MD, HG, MD, src=w, src_ld=A + MD, src_off=A, dst_off=i * HG * MD),
                     _seg(b1, HG, 1, HG, src=b, src_ld=HG, dst_off=i * HG),
                     _seg(W2, HG, MD, HG, src=w2, src_ld=HG, dst_off=i * MD * HG),
                     _seg(b2, MD, 1, MD, src=bb2, src_ld=MD, dst_off=i * MD)]
        copy2d(segs)
        apre, ws_ap, nb_ap = _raw_linear_fwd(attended, Wa, b1, 0)
        nbm = lib.mmt_mfn_mem_scan_workspace_bytes()
        wsm = torch.empty(nbm, dtype=torch.uint8, device=like.device)
        mem_all, u_all, g_all = _new(M, MD, like=like), _new(M, 2 * HG, like=like), _new(M, 2 * MD, like=like)
        args = (_lib.ptr(apre), _lib.ptr(c_hat), _lib.ptr(Wm), _lib.ptr(W2), _lib.ptr(b2), _lib.ptr(mem_all), _lib.ptr(u_all), _lib.ptr(g_all),
                _lib.ptr(wsm), nbm, T, B, MD, HG, pg)
        if isinstance(seed_g, _lib.DeviceSeed):
            _lib.check(lib.mmt_mfn_mem_scan_forward_devseed(*args, seed_g.ptr(), _lib.stream_ptr()))
        else:
            _lib.check(lib.mmt_mfn_mem_scan_forward(*args, int(seed_g), _lib.stream_ptr()))
        # [h of every modality ; mem]  (:241-243) and the read-out MLP (:244-246)
        last = _new(M, SH + MD, like=like)
        segs, col = [], 0
        for h, H in zip(hs, Hs):
            segs.append(_seg(last, SH + MD, M, H, src=h, src_ld=H, dst_off=col))
            col += H
        segs.append(_seg(last, SH + MD, M, MD, src=mem_all, src_ld=MD, dst_off=col))
        copy2d(segs)
        hid, ws_o1, nb_o1 = _raw_linear_fwd(last, o1w, o1b, 1, None, 0.0, p_out, seed_out)
        out, ws_o2, nb_o2 = _raw_linear_fwd(hid, o2w, o2b, 0)
        if not any(ctx.needs_input_grad):               # inference: nothing is kept, the workspaces go straight back to the pool
            for w in (ws_a11, ws_a12, ws_a21, ws_a22, ws_ap, ws_o1, ws_o2):
                _lib.POOL.put(w)
            ctx.k = None
            return out.view(T, B, o2w.shape[0])
        ctx.k = dict(nm=nm, T=T, B=B, Hs=Hs, SH=SH, MD=MD, HG=HG, A=A, pg=pg, p_out=p_out, seed_out=seed_out,
                     P=P, c_star=c_star, a1=a1, att=att, attended=attended, a2=a2, c_hat=c_hat, Wa=Wa, Wm=Wm, W2=W2,
                     mem_all=mem_all, u_all=u_all, g_all=g_all, last=last, hid=hid,
                     ws=dict(a11=(ws_a11, nb_a11), a12=(ws_a12, nb_a12), a21=(ws_a21, nb_a21), a22=(ws_a22, nb_a22), ap=(ws_ap, nb_ap),
                             o1=(ws_o1, nb_o1), o2=(ws_o2, nb_o2)))
        return out.view(T, B, o2w.shape[0])

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        k = ctx.k
        if k is None:
            raise RuntimeError("mfn_gate: backward called twice on the same forward (workspaces already released)")
        ctx.k = None
        nm, T, B, Hs, SH, MD, HG, A = k["nm"], k["T"], k["B"], k["Hs"], k["SH"], k["MD"], k["HG"], k["A"]
        (a11w, a11b, a12w, a12b, a21w, a21b, a22w, a22b, g11w, g11b, g12w, g12b, g21w, g21b, g22w, g22b, o1w, o1b, o2w, o2b) = k["P"]
        M = T * B
        st = _lib.stream_ptr()
        g = _f32c(dout).reshape(M, -1)
        like = g
        d_hid, dWo2, dbo2 = _raw_linear_bwd(g, k["hid"], o2w, None, None, *k["ws"]["o2"], True, True, True)
        d_last, dWo1, dbo1 = _raw_linear_bwd(d_hid, k["last"], o1w, k["hid"], None, *k["ws"]["o1"], True, True, True, 1, 0.0, k["p_out"], k["seed_out"])
        d_hs = [_new(T, B, H, like=like) for H in Hs]
        d_mem = _new(M, MD, like=like)
        segs, col = [], 0
        for o, H in zip(d_hs, Hs):
            segs.append(_seg(o, H, M, H, src=d_last, src_ld=SH + MD, src_off=col))
            col += H
        segs.append(_seg(d_mem, MD, M, MD, src=d_last, src_ld=SH + MD, src_off=col))
        copy2d(segs)
        # memory scan backward, then its batched weight gradients (window contractions)
        nbm = lib.mmt_mfn_mem_scan_workspace_bytes()
        wsm = torch.empty(nbm, dtype=torch.uint8, device=like.device)
        d_chat, d_apre, dz = _new(M, MD, like=like), _new(M, 2 * HG, like=like), _new(M, 2 * MD, like=like)
        _lib.check(lib.mmt_mfn_mem_scan_backward(_lib.ptr(d_mem), _lib.ptr(k["c_hat"]), _lib.ptr(k["mem_all"]), _lib.ptr(k["u_all"]), _lib.ptr(k["g_all"]),
                                                 _lib.ptr(k["Wm"]), _lib.ptr(k["W2"]), _lib.ptr(d_chat), _lib.ptr(d_apre), _lib.ptr(dz), _lib.ptr(wsm), nbm,
                                                 T, B, MD, HG, k["pg"], st))
        mem_prev = _new(M, MD, like=like)
        dzs, us = [_new(M, MD, like=like) for _ in range(2)], [_new(M, HG, like=like) for _ in range(2)]
        segs = [_seg(mem_prev, MD, B, MD), _seg(mem_prev, MD, M - B, MD, src=k["mem_all"], src_ld=MD, dst_off=B * MD)]
        for i in range(2):
            segs += [_seg(dzs[i], MD, M, MD, src=dz, src_ld=2 * MD, src_off=i * MD), _seg(us[i], HG, M, HG, src=k["u_all"], src_ld=2 * HG, src_off=i * HG)]
        copy2d(segs)

        def wgrad(gg, xx, Wt, want_b):
            """dW = gg^T xx (and db = column sums of gg) through the weight-gradient path of the affine map"""
            Mm, Kk, Nn = xx.shape[0], xx.shape[1], gg.shape[1]
            lb = lib.mmt_linear_workspace_bytes(Mm, Kk, Nn)
            lws = _lib.POOL.get(lb, like.device, tag=("linear", Mm, Kk, Nn))
            dW = _new(Nn, Kk, like=like)
            db = _new(Nn, like=like) if want_b else None
            _lib.check(lib.mmt_linear_backward(_lib.ptr(gg), _lib.ptr(xx), _lib.ptr(Wt), None, None, None, _lib.ptr(dW), _lib.ptr(db),
                                               _lib.ptr(lws), lb, Mm, Kk, Nn, 0, st))
            _lib.POOL.put(lws)
            return dW, db
        dWm, _ = wgrad(d_apre, mem_prev, k["Wm"], False)
        dW2b2 = [wgrad(dzs[i], us[i], k["W2"][i], True) for i in range(2)]
        d_att, dWa, db1 = _raw_linear_bwd(d_apre, k["attended"], k["Wa"], None, None, *k["ws"]["ap"], True, True, True)
        d_a2, dWa22, dba22 = _raw_linear_bwd(d_chat, k["a2"], a22w, k["c_hat"], None, *k["ws"]["a22"], True, True, True, 2)
        d_att2, dWa21, dba21 = _raw_linear_bwd(d_a2, k["attended"], a21w, k["a2"], None, *k["ws"]["a21"], True, True, True, 1)
        copy2d([_seg(d_att, A, M, A, src=d_att2, src_ld=A, acc=True)])                 # `attended` feeds att2_fc1 and both gamma fc1
        d_logits, d_cstar = torch.empty_like(d_att), torch.empty_like(d_att)
        _lib.check(lib.mmt_softmax_mul_backward(_lib.ptr(d_att), _lib.ptr(k["att"]), _lib.ptr(k["c_star"]), _lib.ptr(d_logits), _lib.ptr(d_cstar), M, A, st))
        d_a1, dWa12, dba12 = _raw_linear_bwd(d_logits, k["a1"], a12w, None, None, *k["ws"]["a12"], True, True, True)
        d_cs2, dWa11, dba11 = _raw_linear_bwd(d_a1, k["c_star"], a11w, k["a1"], None, *k["ws"]["a11"], True, True, True, 1)
        # dc_t = d cStar[new part]_t + d cStar[prev part]_{t+1}; both cStar paths (att1 MLP input, the product) summed on the way
        d_cs = [_new(T, B, H, like=like) for H in Hs]
        segs, col = [], 0
        for o, H in zip(d_cs, Hs):
            pc, nc = col, SH + col
            segs += [_seg(o, H, M, H, src=d_cstar, src_ld=A, src_off=nc, src2=d_cs2, src2_ld=A, src2_off=nc)]
            col += H
        copy2d(segs)
        segs, col = [], 0
        for o, H in zip(d_cs, Hs):
            segs += [_seg(o, H, M - B, H, src=d_cstar, src_ld=A, src_off=B * A + col, src2=d_cs2, src2_ld=A, src2_off=B * A + col, acc=True)]
            col += H
        copy2d(segs)
        # gradients of the gamma parameters back into the reference's tensors
        dg = [(_new(HG, A + MD, like=like), _new(HG, like=like), _new(MD, HG, like=like), _new(MD, like=like)) for _ in range(2)]
        segs = []
        for i, (dw, db, dw2, db2) in enumerate(dg):
            segs += [_seg(dw, A + MD, HG, A, src=dWa, src_ld=A, src_off=i * HG * A),
                     _seg(dw, A + MD, HG, MD, src=dWm, src_ld=MD, src_off=i * HG * MD, dst_off=A),
                     _seg(db, HG, 1, HG, src=db1, src_ld=HG, src_off=i * HG),
                     _seg(dw2, HG, MD, HG, src=dW2b2[i][0], src_ld=HG),
                     _seg(db2, MD, 1, MD, src=dW2b2[i][1], src_ld=MD)]
        copy2d(segs)
        gp = (dWa11, dba11, dWa12, dba12, dWa21, dba21, dWa22, dba22, dg[0][0], dg[0][1], dg[0][2], dg[0][3],
              dg[1][0], dg[1][1], dg[1][2], dg[1][3], dWo1, dbo1, dWo2, dbo2)
        return (None, None, None, None, None) + tuple(d_hs) + tuple(d_cs) + gp


def mfn_gate(hs, cs, params, gamma_dropout=0.0, gamma_seed=0, out_dropout=0.0, out_seed=0):
    """hs, cs: per-modality (T,B,H_m) LSTM states; params: the 20 gate tensors (see _MfnGateFn.NP) -> (T,B,output_dim)."""
    def sd(v):
        return v if isinstance(v, _lib.DeviceSeed) else int(v)
    return _MfnGateFn.apply(len(hs), float(gamma_dropout), sd(gamma_seed), float(out_dropout), sd(out_seed), *hs, *cs, *params)


def check_device_errors():
    """Synchronise and raise if an asynchronous kernel reported an error through its device error word (mmt_hip.h)."""
    _lib.ERRORS.check()


def poison_lds(device, pattern=0x7FC00000):
    """Test hook: leave `pattern` in every LDS word of every CU (see include/mmt_hip.h)."""
    lib = _lib.load()
    sink = torch.zeros(1, dtype=torch.int32, device=device)
    _lib.check(lib.mmt_debug_poison_lds(int(pattern), _lib.ptr(sink), _lib.stream_ptr()))


def dropout_mask(p, seed, stream_id, n, device, attn_Tp=0):
    """Test hook: (keep mask as a bool tensor of n entries, scale of kept values) of one dropout stream."""
    import ctypes
    lib = _lib.load()
    keep = torch.empty(n, dtype=torch.uint8, device=device)
    sc = ctypes.c_float(0.0)
    _lib.check(lib.mmt_debug_dropout_mask(float(p), int(seed), int(stream_id), int(n), int(attn_Tp), _lib.ptr(keep),
                                          ctypes.cast(ctypes.pointer(sc), ctypes.c_void_p), _lib.stream_ptr()))
    return keep.bool(), float(sc.value)
