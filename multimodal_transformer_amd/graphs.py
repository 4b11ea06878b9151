"""hipGraph capture of a training step, with the one failure that kills the process turned into an exception.

What goes wrong without it (ROCm 7.2, torch 2.10; repro: tools/repro_capture_stale_autograd.py).  An eager warm-up step creates the
AccumulateGrad nodes of the parameters on the stream it ran on — normally the default stream.  If anything still references that
step's autograd graph (its loss, its output) when the step is captured on another stream, autograd re-uses those nodes: the captured
backward hands each gradient from the capture stream to the node's OWN stream, i.e. it records an event on the capture stream and makes
the default stream wait for it.  That wait pulls the default stream INTO the capture as a forked branch which nothing ever joins, and
`hipStreamEndCapture` on a capture with an unjoined branch does not return hipErrorStreamCaptureUnjoined on this ROCm, it segfaults
(gpurun_out/dbg4, dbg5 of round 3: `Fatal Python error: Segmentation fault` in `capture_end`).  torch warns about exactly this
("The AccumulateGrad node's stream does not match the stream of the node that produced the incoming gradient ... break CUDA graph
capture"), once per process and only as a warning.

`capture_step` therefore runs the side-stream warm-up that a capture needs anyway with that warning promoted to an error — the stale
graph is found BEFORE `capture_begin`, where raising is harmless — and refuses to capture.  The two related observations of round 3
(a captured hipMemsetAsync and a captured device-to-host copy misbehaving under replay) were the same kind of event: work that a capture
placed on a stream nobody joined; the library launches kernels only (`zero_fill_kernel`, `error_accumulate_kernel`), so neither can
occur in a captured step any more.

torch emits that warning through TORCH_WARN_ONCE: once it has fired anywhere in the process — e.g. from an unguarded raw capture of an
earlier test — a later warm-up would see nothing.  The warm-up therefore runs under `torch.set_warn_always(True)` (restored afterwards),
which makes every occurrence visible, so the guard does not depend on being the first to see the condition
(`test_capture_guard_works_repeatedly_in_one_process`).

A stale graph whose AccumulateGrad nodes sit on a SIDE stream (an eager warm-up that itself ran under `torch.cuda.stream(s)`) was
observed NOT to crash a raw capture (round 4's `test_device_resident_seed_fresh_masks_per_graph_replay` captured that way and emitted
the warning).  The likely reason: `torch.cuda.graph` captures on a side stream of its own and the engine, at the end of the backward,
makes the stream that called `backward()` wait for every stream its nodes ran on, so `s` is joined back into the capture before
`capture_end`; the DEFAULT stream is the one the capture machinery never waits for.  The guard refuses both forms — the surviving one
still runs gradient accumulation on a foreign stream inside the graph — and no test captures that way any more.
"""
import warnings

import torch

_STALE = ("a stale autograd graph of an earlier eager step is still alive (its output or loss is still referenced): its AccumulateGrad "
          "nodes are bound to the stream that step ran on, and capturing a step that re-uses them forks that stream into the capture, "
          "which ends in a segmentation fault inside hipStreamEndCapture on ROCm 7.2.  Drop every reference to earlier outputs / "
          "losses (or call .detach() on what you keep) before capturing.")


class StaleAutogradGraphError(RuntimeError):
    pass


def capture_step(step, warmup=2, graph=None, **graph_kwargs):
    """Warm `step()` up on a side stream, then capture one call of it -> (graph, value returned by the captured call).

    Raises StaleAutogradGraphError (before anything is captured) when the warm-up shows that autograd would hand gradients to
    AccumulateGrad nodes of another stream."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(True)         # torch's default; a caller may have silenced it
    warn_always = torch.is_warn_always_enabled()
    torch.set_warn_always(True)                         # TORCH_WARN_ONCE would hide the condition after its first occurrence in the process
    with warnings.catch_warnings():
        warnings.filterwarnings("error", message=r".*AccumulateGrad node's stream does not match.*")
        try:
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    step()
        except UserWarning as w:
            torch.cuda.synchronize()
            raise StaleAutogradGraphError(_STALE) from w
        except RuntimeError as e:                       # a warning raised inside the autograd engine's thread arrives re-wrapped
            if "AccumulateGrad node's stream does not match" in str(e):
                torch.cuda.synchronize()
                raise StaleAutogradGraphError(_STALE) from e
            raise
        finally:
            torch.set_warn_always(warn_always)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = graph if graph is not None else torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, **graph_kwargs):
        out = step()
    return g, out
