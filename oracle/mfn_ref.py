"""Oracle: MFN delta-memory attention gate (TEST INFRASTRUCTURE — see oracle/__init__.py).

Restates transformer/MFT/multiTransformer.py:118-248.  Dropout (gamma{1,2}_dropout :222-223, out_dropout :245) is taken as explicit
multipliers (keep / (1 - p), or None = identity = eval mode), so a train-mode pass of the HIP path can be replayed with the masks its
kernels drew.  The statement here is
*phase-ordered* (all LSTM steps first, then everything that depends only on the cell
states, then the memory recurrence, then the per-step read-out) because that is the
decomposition the HIP path uses; the arithmetic per element is the reference's.
"""
import torch

HIDDEN = {"linguistic": 88, "emotient": 16, "acoustic": 48, "image": 88}  # :128
MEM_DIM = 128                                                                  # :133


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTMCell semantics (gate order i, f, g, o), as called at :208."""
    gates = x @ w_ih.transpose(0, 1) + b_ih + h @ w_hh.transpose(0, 1) + b_hh
    H = h.shape[-1]
    i = torch.sigmoid(gates[..., 0 * H:1 * H])
    f = torch.sigmoid(gates[..., 1 * H:2 * H])
    g = torch.tanh(gates[..., 2 * H:3 * H])
    o = torch.sigmoid(gates[..., 3 * H:4 * H])
    c_new = f * c + i * g
    return o * torch.tanh(c_new), c_new


def _fc(p, name, x):
    return x @ p[name + ".weight"].transpose(0, 1) + p[name + ".bias"]


def mfn_gate(p, prefix, inputs, mods, gamma_drop=None, out_drop=None):
    """inputs: {mod: (T, B, d_mod)} -> (B, T, 1).  Lines :181-248.

    ``mods`` order fixes the concatenation order of the per-modality states (:212-217, :241-243).
    ``gamma_drop``: (m1, m2), each (T, B, 64): multipliers on relu(gamma1_fc1(both)) and relu(gamma2_fc1(both)) (:222-223).
    ``out_drop``: (T, B, 64): multiplier on relu(out_fc1(last)) (:245).
    """
    first = inputs[mods[0]]
    T, B = first.shape[0], first.shape[1]
    kw = dict(dtype=first.dtype, device=first.device)

    # phase A — per-modality LSTMCell recurrences (:207-208), zero initial state (:196-197)
    hs, cs_prev, cs_new = {}, {}, {}
    for mod in mods:
        H = HIDDEN[mod]
        h = torch.zeros(B, H, **kw)
        c = torch.zeros(B, H, **kw)
        hs[mod], cs_prev[mod], cs_new[mod] = [], [], []
        w = [p["%slstm_%s.%s" % (prefix, mod, n)] for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        for t in range(T):
            cs_prev[mod].append(c)
            h, c = lstm_cell(inputs[mod][t], h, c, *w)
            hs[mod].append(h)
            cs_new[mod].append(c)
    h_all = torch.cat([torch.stack(hs[m]) for m in mods], dim=-1)            # (T,B,sumH)
    c_star = torch.cat([torch.stack(cs_prev[m]) for m in mods] +
                       [torch.stack(cs_new[m]) for m in mods], dim=-1)       # (T,B,2 sumH) :215-217

    # phase B — terms that depend only on the cell states (:218-220)
    att = torch.softmax(_fc(p, prefix + "att1_fc2", torch.relu(_fc(p, prefix + "att1_fc1", c_star))), dim=-1)
    attended = att * c_star
    c_hat = torch.tanh(_fc(p, prefix + "att2_fc2", torch.relu(_fc(p, prefix + "att2_fc1", attended))))

    # phase C — memory recurrence (:221-224), zero initial memory (:198)
    mem = torch.zeros(B, MEM_DIM, **kw)
    mems = []
    for t in range(T):
        both = torch.cat([attended[t], mem], dim=-1)
        u1 = torch.relu(_fc(p, prefix + "gamma1_fc1", both))
        u2 = torch.relu(_fc(p, prefix + "gamma2_fc1", both))
        if gamma_drop is not None:
            u1, u2 = u1 * gamma_drop[0][t], u2 * gamma_drop[1][t]
        g1 = torch.sigmoid(_fc(p, prefix + "gamma1_fc2", u1))
        g2 = torch.sigmoid(_fc(p, prefix + "gamma2_fc2", u2))
        mem = g1 * mem + g2 * c_hat[t]
        mems.append(mem)
    mem_all = torch.stack(mems)                                              # (T,B,128)

    # phase D — per-step read-out (:238-247)
    last = torch.cat([h_all, mem_all], dim=-1)
    hid = torch.relu(_fc(p, prefix + "out_fc1", last))
    if out_drop is not None:
        hid = hid * out_drop
    out = _fc(p, prefix + "out_fc2", hid)                                    # (T,B,1)
    return out.permute(1, 0, 2)
