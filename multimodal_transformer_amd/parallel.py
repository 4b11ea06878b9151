"""Data parallelism over narrative sequences: one process per GPU, gradients summed with RCCL.

The reference is single-process; sequences never interact (attention is within a
sequence, the loss is a sum over sequences), so the batch dimension shards with ONE exchange step per
training step: a SUM all-reduce of the gradients.  Equivalence with the single-GPU reference step needs
the loss scaled by the GLOBAL number of valid windows (transformer/SFT/train.py:137 divides the summed
loss by sum(lengths) of the whole batch) — see ``global_window_count``.

xGMI is point-to-point and the gradient payload is small (MBs), so the exchange is latency-bound: grads
are reduced in as few, as large buffers as possible.  The fused encoder returns all of its parameter
gradients as views of one flat buffer, which is all-reduced in place with no staging copy.
"""
import torch
import torch.distributed as dist


def global_window_count(lengths, group=None):
    """sum(lengths) over every rank's shard (python int)."""
    n = torch.tensor([float(sum(lengths))], dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        n = n.to(dev)
        dist.all_reduce(n, op=dist.ReduceOp.SUM, group=group)
    return int(round(n.item()))


def gradient_buckets(params):
    """Group existing gradients by the allocation they live in.

    Returns (flat_bases, loose): ``flat_bases`` are 1-D tensors spanning a whole allocation that several gradients are
    views of (the fused encoder hands all its parameters views of one flat buffer): reduce them in place;
    ``loose`` are gradients that own their storage (to be coalesced)."""
    groups, order = {}, []
    for p in params:
        g = p.grad
        if g is None:               # e.g. the reference's dead attn{mod}/ff{mod} parameters
            continue
        key = g.untyped_storage().data_ptr()
        if key not in groups:
            groups[key] = []
            order.append(key)
        groups[key].append(g)
    bases, loose = [], []
    for key in order:
        gs = groups[key]
        st = gs[0].untyped_storage()
        covered = sum(g.numel() * g.element_size() for g in gs)
        if len(gs) > 1 and all(g.is_contiguous() for g in gs) and covered == st.nbytes():
            bases.append(torch.empty(0, dtype=gs[0].dtype, device=gs[0].device).set_(st, 0, (st.nbytes() // gs[0].element_size(),)))
        else:
            loose.extend(gs)
    return bases, loose


def allreduce_gradients(params, group=None, force=False):
    """SUM all-reduce of every gradient, in place.  Returns the number of collectives issued.
    A world of one rank needs no exchange and issues none, unless ``force`` (used to exercise the collective path itself)."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    if dist.get_world_size(group) == 1 and not force:
        return 0
    bases, loose = gradient_buckets(list(params))
    n = 0
    for b in bases:
        dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
        n += 1
    if loose:
        flat = torch._utils._flatten_dense_tensors(loose)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        torch._foreach_copy_(loose, list(torch._utils._unflatten_dense_tensors(flat, loose)))    # one multi-tensor kernel
        n += 1
    return n


def shard_batch(n_sequences, rank, world_size):
    """Contiguous shard [lo, hi) of the sequence axis for this rank (sizes differ by at most one)."""
    base, rem = divmod(n_sequences, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
