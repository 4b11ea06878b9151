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
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = graph if graph is not None else torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, **graph_kwargs):
        out = step()
    return g, out
