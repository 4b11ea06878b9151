#!/usr/bin/env python3
"""Throughput of the attention hot path on MI355X: windows/sec, forward+backward.

    python bench.py --gpus N --steps K --warmup W          (N>1 without a launcher: starts its own N ranks through torch.distributed.run)

One step = one TRAIN-MODE (dropout 0.1) forward + MSE loss + backward pass (all input, weight and
LayerNorm gradients) over one synthetic batch already resident in HBM; with N>1 each rank holds its own
B sequences (weak scaling) and the step ends with the RCCL SUM all-reduce of the gradients.
A window is one time-step of one sequence: windows/step = N*B*T.

Primary workload (`value`): the encoder-stack hot path of the SFT configuration the metric is quoted on
(C4: T=500, d_model=128, heads=8, N=6, d_ff=128, 32 sequences per GPU).  Rank 0 prints ONE JSON line with
  roofline     — the dominant kernel of the step (by HIP-event time measured here, on the stream the
                 kernels run on): algorithmic FLOPs per launch / average duration vs the dense bf16 MFMA peak;
  cpu_baseline — the CPU oracle (a port of the reference path, fp32 torch CPU) timed on this box's host
                 cores on a bounded sample of the same workload;
  with_adam    — the same step followed by the reference's optimiser (Adam; optim.FlatAdam, one launch) [+ the all-reduce];
  full_model   — (N=1) the whole SFT sequence model NLPTransformer(512 -> d): embed + encoder + LSTM
                 decoder + MLP + mask, same batch shape, for the end-to-end picture;
  raw_pipeline — (N=1) raw windows -> valence: the 3-modality MultiCNNTransformer (CNN k=2 + max-pool + Highway
                 front-end, fusion, NLPTransformer) with the raw fp32 windows resident in HBM; conv_fwd = the
                 MFMA fraction of the window-encoder kernel;
  mft_model    — (N=1) the whole MFT model of configs[2] (3 modality encoders + MFN gate, T=300).
  config_full_batch — (N=1) configs[3] at its full batch of 256 sequences on ONE GPU (the 32-sequence slice above is the
                 weak-scaling anchor; the batch also fits one GPU);
  mft_configs4 — (N=1) the whole MFT model at the configs[4] per-GPU slice (3 modalities, T=1000, 64 sequences, d=256).
`roofline.traffic` comes from profiles/r*_pmc_per_kernel.json (separate rocprofv3 --pmc passes of this workload) and only if
that summary was collected on the kernel sources now running (sha of csrc/); otherwise it is null with the reason.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)

# per-GPU slices of the BASELINE.json configs (N=6, d_ff=128: constructor defaults of the reference)
WORKLOADS = {
    "C4": dict(desc="SFT encoder stack T=500 d_model=128 heads=8 N=6 d_ff=128, 32 sequences/GPU", B=32, T=500, d=128, h=8, N=6, f=128),
    "C2": dict(desc="SFT encoder stack T=300 d_model=40 heads=4 N=6 d_ff=128, 32 sequences/GPU", B=32, T=300, d=40, h=4, N=6, f=128),
    "C3e": dict(desc="MFT per-modality encoder stack T=300 d_model=256 heads=8 N=6 d_ff=128, 32 sequences/GPU", B=32, T=300, d=256, h=8, N=6, f=128),
    "C5e": dict(desc="MFT per-modality encoder stack T=1000 d_model=256 heads=8 N=6 d_ff=128, 64 sequences/GPU", B=64, T=1000, d=256, h=8, N=6, f=128),
}
DROPOUT = 0.1


def flops_per_window_layer_fwd(d, T, f):
    """Algorithmic matmul FLOPs per window per layer, forward: 8d^2 + 4Td + 4df."""
    return 8 * d * d + 4 * T * d + 4 * d * f


def site_flops_per_launch(site, M, T, d, f, n_layers=1):
    """Algorithmic FLOPs of one launch of a kernel site (no credit for recomputation)."""
    table = {
        "rowgemm<FRAG,LN>:ln1+qkv": 6 * d * d,
        "attn_fwd_kernel": 4 * T * d,                                   # QK^T and PV
        "chain:outproj+res>ln2+ffn1>ffn2+res": 2 * d * d + 4 * d * f,
        "chain:outproj+res>ln2+ffn1>ffn2+res>ln1+qkv(next)": 2 * d * d + 4 * d * f + 6 * d * d,
        "chain:bwd_ffn2>bwd_ffn1+ln2>bwd_outproj->dO": 4 * d * f + 2 * d * d,
        "chain:bwd_qkv+ln1>bwd_ffn2(below)>bwd_ffn1+ln2>bwd_outproj->dO": 6 * d * d + 4 * d * f + 2 * d * d,
        "attn_bwd_dkv_kernel": 6 * T * d,                               # dV, dP, dK  (S recomputed: no credit)
        "attn_bwd_dq_kernel": 2 * T * d,                                # dQ          (S, dP recomputed: no credit)
        "attn_bwd_pair16_kernel": 8 * T * d,                            # dV, dP, dK, dQ in one launch (S recomputed: no credit)
        "rowgemm<LNBWD>:bwd_qkv+ln1": 6 * d * d,
        "wgrad_kernel": (8 * d * d + 4 * d * f) * n_layers,             # one launch covers every layer
        "rowgemm<PLAIN>:outproj+res": 2 * d * d,
        "rowgemm<PLAIN,LN>:ln2+ffn1+relu": 2 * d * f,
        "rowgemm<PLAIN>:ffn2+res": 2 * d * f,
        "rowgemm<PLAIN>:bwd_ffn2": 2 * d * f,
        "rowgemm<LNBWD>:bwd_ffn1+ln2": 2 * d * f,
        "rowgemm<FRAG>:bwd_outproj->dO": 2 * d * d,
    }
    return table.get(site, 0) * M


SITE_KERNELS = {          # launch site -> kernel symbol prefix in the rocprofv3 tables
    "attn_fwd_kernel": "attn_fwd_kernel", "attn_bwd_dkv_kernel": "attn_bwd_dkv_kernel", "attn_bwd_dq_kernel": "attn_bwd_dq_kernel",
    "attn_bwd_pair16_kernel": "attn_bwd_pair16_kernel",
    "chain:outproj+res>ln2+ffn1>ffn2+res": "encoder_post_attn_fwd_kernel",
    "chain:outproj+res>ln2+ffn1>ffn2+res>ln1+qkv(next)": "encoder_post_attn_fwd4_kernel",
    "chain:bwd_ffn2>bwd_ffn1+ln2>bwd_outproj->dO": "encoder_pre_attn_bwd_kernel",
    "chain:bwd_qkv+ln1>bwd_ffn2(below)>bwd_ffn1+ln2>bwd_outproj->dO": "encoder_bwd_boundary_kernel",
    "rowgemm<FRAG,LN>:ln1+qkv": "rowgemm_kernel<1, true>", "rowgemm<LNBWD>:bwd_qkv+ln1": "rowgemm_kernel<2, false>",
    "wgrad_kernel": "wgrad_kernel",
}


def kernel_source_sha():
    """sha256 over the kernel sources (csrc/*.h, api.hip): identifies the code a counter summary was collected on.
    (The GPU box has no .git, so the git head cannot serve; tools/summarize_profiles.py records the same hash.)"""
    import hashlib
    csrc = os.path.join(ROOT, "multimodal_transformer_amd", "csrc")
    hsh = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".h", ".hip")):
            with open(os.path.join(csrc, name), "rb") as fh:
                hsh.update(name.encode() + b"\0" + fh.read())
    return hsh.hexdigest()[:16]


HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s

# Vector-issue ceilings of the attention kernels at d_k = 16: cycles of SIMD issue per 32x32 score tile from the kernel's
# per-score instruction list and the per-instruction costs measured by tools/valu_micro.hip (profiles/r05_valu_micro.txt: v_exp_f32 8.2,
# v_bfe_i32 / v_cvt_pk_bf16_f32 / v_max3_f32 4.2, v_and / v_mul / v_add 2.3, v_fma 2.4 per wave64 instruction and SIMD with >= 2 waves
# resident; an MFMA holds the SIMD's issue for 8), against the matrix pipe's 1017 FLOP per cycle and SIMD (2.5 PFLOP/s / 1024 SIMDs / 2.4 GHz).
# value = (credited FLOP per tile / issue cycles per tile) / 1017 = the fraction of the bf16 MFMA peak at which vector issue saturates.
VALU_CEILING = {
    # per register (16 per tile): exp, bfe, and, mul, fma, cvt = 8.2 + 4.2 + 2.3 + 2.3 + 2.4 + 4.2; 8 MFMAs; 8 * 32 * 32 * 16 FLOP credited
    "attn_bwd_pair16_kernel": (8 * 32 * 32 * 16) / (16 * (8.2 + 4.2 + 2.3 + 2.3 + 2.4 + 4.2) + 8 * 8) / 1017.0,
    # per register: exp, bfe, and, add (row sum), half a max3, half a cvt; 3 MFMAs; 4 * 32 * 32 * 16 FLOP credited
    "attn_fwd_kernel": (4 * 32 * 32 * 16) / (16 * (8.2 + 4.2 + 2.3 + 2.3 + 2.1 + 2.1) + 3 * 8) / 1017.0,
}


def pmc_table(train, cfg):
    """the newest counter table under profiles/ collected for this workload and mode on the running kernel sources -> (table, path) or (None, why)"""
    import glob
    sha, why = kernel_source_sha(), "no counter table of this workload and mode under profiles/"
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_per_kernel*.json")), reverse=True):
        try:
            with open(path) as fh:
                tab = json.load(fh)
        except (OSError, ValueError):
            continue
        meta, rel = tab.get("_meta", {}), os.path.relpath(path, ROOT)
        if meta.get("workload") != cfg["desc"] or bool(meta.get("train")) != bool(train):
            continue
        if meta.get("csrc_sha") != sha:
            if not why.startswith("stale"):
                why = "stale: %s was collected on kernel sources %s, running %s" % (rel, meta.get("csrc_sha"), sha)
            continue
        return tab, rel
    return None, why


def step_hbm(train, cfg, ms_per_step):
    """HBM-side bytes of one whole step (every kernel of this library in the counter table x its launches per profiled step) and the
    average rate they amount to over the measured step time -> dict, or None with the reason"""
    tab, src = pmc_table(train, cfg)
    if tab is None:
        return {"hbm_bytes_per_step": None, "source": src}
    if "steps_profiled" not in tab["_meta"]:
        return {"hbm_bytes_per_step": None, "source": src + ": collected without --headline-only (its dispatch counts mix other shapes)"}
    steps = float(tab["_meta"]["steps_profiled"])
    tot = 0.0
    for k, v in tab.items():
        if k == "_meta" or k.startswith("at::") or k.startswith("__amd") or "hbm_bytes_per_launch" not in v:
            continue
        tot += v["hbm_bytes_per_launch"] * v.get("dispatches", 0) / steps
    gbps = tot / (ms_per_step * 1e-3) / 1e9
    return {"hbm_bytes_per_step": int(tot), "hbm_gbps": round(gbps, 1), "hbm_frac": round(gbps / HBM_PEAK_GBPS, 4), "source": src}


def pmc_traffic(site, train, cfg):
    """HBM-side bytes per launch of the site's kernel from a committed rocprofv3 counter summary (separate --pmc passes of
    this same workload; FETCH_SIZE x2 + WRITE_SIZE, KiB units — MI355X_MICROARCH.md 'HBM').  bench.py cannot run the
    profiler around itself, so the figure is read from profiles/ — and ONLY from a summary collected on exactly the kernel
    sources that are running now (`_meta.csrc_sha`), for this workload and mode; otherwise traffic is null and the reason is
    given instead of a stale number."""
    if site not in SITE_KERNELS:
        return None, "no counter mapping for this launch site"
    tab, src = pmc_table(train, cfg)
    if tab is None:
        return None, src
    pref = SITE_KERNELS[site]
    for k, v in tab.items():
        if k.startswith(pref) and "hbm_bytes_per_launch" in v:
            return int(v["hbm_bytes_per_launch"]), "%s (%s)" % (src, k)
    return None, "%s has no entry for %s" % (src, pref)


def roofline_of(prof, nsteps, M, T, d, f, N, train, cfg, top=12):
    """{site: (total_ms, launches)} of `nsteps` profiled steps -> (kernel_ms_per_step, roofline block of the dominant kernel).
    `cfg`: the WORKLOADS entry whose counter table (profiles/) holds this kernel at this shape."""
    kernel_ms = {k: round(v[0] / nsteps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])[:top]}
    name, (tot_ms, cnt) = max(prof.items(), key=lambda kv: kv[1][0])
    avg_s = tot_ms / cnt * 1e-3
    fl = site_flops_per_launch(name, M, T, d, f, N)
    ach = fl / avg_s / 1e12 if avg_s > 0 else 0.0
    traffic, traffic_src = pmc_traffic(name, train, cfg)
    roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 5), "traffic": traffic, "traffic_source": traffic_src,
            "avg_launch_us": round(avg_s * 1e6, 2), "launches_per_step": cnt // nsteps, "flops_per_launch": fl,
            "share_of_kernel_time": round(tot_ms / sum(v[0] for v in prof.values()), 3)}
    # the same kernel against the chip's HBM peak (north_star: "rocprof HBM GB/s ... reported against chip peak")
    roof["hbm_gbps"] = round(traffic / avg_s / 1e9, 1) if traffic and avg_s > 0 else None
    roof["hbm_frac"] = round(roof["hbm_gbps"] / HBM_PEAK_GBPS, 4) if roof["hbm_gbps"] is not None else None
    # ... and against the ceiling its own vector instructions set (the softmax arithmetic issues on the SIMD the MFMAs issue on)
    if name in VALU_CEILING:
        roof["valu_ceiling_frac"] = round(VALU_CEILING[name], 4)
        roof["frac_of_valu_ceiling"] = round(roof["frac"] / VALU_CEILING[name], 3)
    return kernel_ms, roof


def profiled(step, nsteps):
    """per-kernel HIP-event times of `nsteps` eager calls of `step`, modality streams serialised (an event pair only times its own stream)"""
    from multimodal_transformer_amd import _lib
    prev = os.environ.get("MMT_MODALITY_STREAMS")
    os.environ["MMT_MODALITY_STREAMS"] = "0"
    try:
        step()
        torch.cuda.synchronize()
        _lib.profile(True)
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile(False)
    finally:
        if prev is None:
            os.environ.pop("MMT_MODALITY_STREAMS", None)
        else:
            os.environ["MMT_MODALITY_STREAMS"] = prev
    return prof


def make_encoder(cfg):
    from multimodal_transformer_amd import multiTransformer as MT
    torch.manual_seed(1)                                    # transformer/SFT/train.py:522
    return MT.Encoder(MT.EncoderLayer(cfg["d"], MT.MultiHeadedAttention(cfg["h"], cfg["d"]),
                                      MT.PositionwiseFeedForward(cfg["d"], cfg["f"], DROPOUT), DROPOUT), cfg["N"])


def cpu_baseline(cfg, seconds_budget=15.0):
    """Oracle (port of the reference path) fwd+bwd on host cores, bounded sample of the same workload."""
    import oracle
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)          # the GPU box gives one GPU a 16-CPU share; more threads only oversubscribe it
    torch.set_num_threads(cores)
    enc = make_encoder(cfg)
    p = {k: v.detach().clone().requires_grad_() for k, v in enc.state_dict().items()}
    Bs = max(1, min(cfg["B"], 4))
    T, d = cfg["T"], cfg["d"]
    x = torch.randn(Bs, T, d, requires_grad=True)
    mask = torch.ones(Bs, T, 1)
    tgt = torch.rand(Bs, T, d)

    def step():
        for v in p.values():
            v.grad = None
        x.grad = None
        y = oracle.encoder_stack(p, "", x, mask, cfg["h"])
        (((y - tgt) ** 2).sum() / (Bs * T)).backward()

    step()
    t0 = time.perf_counter()
    n = 0
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 > seconds_budget or n >= 10:
            break
    dt = (time.perf_counter() - t0) / n
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(Bs * T / dt, 1), "unit": "windows/s", "cores": cores, "cpu": cpu_model, "kind": "port",
            "sample": "%d steps of the CPU oracle (fp32 torch-CPU port of the reference path, dropout as identity) on %d of the %d "
                      "sequences, same T/d/h/N; %.3f s/step" % (n, Bs, cfg["B"], dt)}


class Runner:
    """warm-up, optional hipGraph capture, timed loop of `fwd_bwd` (+ gradient all-reduce when world > 1)."""

    def __init__(self, fwd_bwd, params, world, use_graph, warmup):
        from multimodal_transformer_amd import parallel
        self.fwd_bwd, self.params, self.world, self.parallel = fwd_bwd, params, world, parallel
        self.graph, self.launch = None, "eager"
        self.exchange = "none" if world == 1 else "eager"       # where the gradient all-reduce runs: behind the graph, or as a node of it
        for _ in range(max(1, warmup)):
            self.eager()
        torch.cuda.synchronize()
        if use_graph:
            import torch.distributed as dist
            # With RCCL the collective is captured as a node of the step's graph (it starts the moment the finalize kernel ends and costs no
            # launch of its own); the gloo rehearsal and any failure to capture it keep it behind the replay.
            # Opt-in (MMT_BENCH_CAPTURE_ALLREDUCE=1): capturing it was verified with a one-rank RCCL group on the GPU box
            # (tools/probe_graph_allreduce.py); no multi-GPU node was available to verify it at N > 1.
            with_exchange = (world > 1 and dist.is_initialized() and dist.get_backend() == "nccl"
                             and os.environ.get("MMT_BENCH_CAPTURE_ALLREDUCE") == "1")
            for attempt in ((True, False) if with_exchange else (False,)):
                try:
                    def captured_step(with_ar=attempt):
                        fwd_bwd()
                        if with_ar:
                            self.parallel.allreduce_gradients(self.params)
                    # side-stream warm-up + capture; refuses (before capture_begin) a stale autograd graph of an eager step, which would
                    # otherwise end the process inside hipStreamEndCapture (multimodal_transformer_amd/graphs.py)
                    from multimodal_transformer_amd import graphs
                    g, _ = graphs.capture_step(captured_step, warmup=2, capture_error_mode="thread_local" if attempt else "global")
                    self.graph, self.launch = g, "hipgraph"
                    if attempt:
                        self.exchange = "captured"
                    break
                except Exception as e:  # noqa: BLE001
                    print("graph capture%s failed (%s)" % (" with the all-reduce" if attempt else "", str(e).splitlines()[0]), file=sys.stderr)
                    self.graph = None
                    torch.cuda.synchronize()
        for _ in range(max(1, warmup)):
            self.step()
        torch.cuda.synchronize()

    def eager(self):
        self.fwd_bwd()
        self.parallel.allreduce_gradients(self.params)

    def step(self):
        if self.graph is not None:
            self.graph.replay()
            if self.exchange != "captured":
                self.parallel.allreduce_gradients(self.params)
        else:
            self.eager()

    def allreduce_ms(self, steps):
        """mean device time of the gradient exchange alone (HIP events on the launch stream), outside the timed region"""
        if self.world == 1:
            return None
        tot = 0.0
        for _ in range(steps):
            if self.graph is not None:
                self.graph.replay()
            else:
                self.fwd_bwd()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self.parallel.allreduce_gradients(self.params)
            b.record()
            b.synchronize()
            tot += a.elapsed_time(b)
        return tot / steps

    def timed(self, steps, dist=None, dev=None):
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if self.world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same args>`
    as a CHILD process (never exec: this process stays a plain relay and makes no GPU call), pass the ranks' stderr through, print the
    one JSON line rank 0 wrote and return the launcher's exit code."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):     # a scheduler's presets must not leak into the ranks
        env.pop(k, None)
    # --standalone: the launcher picks a free rendezvous port itself (a port found by bind-and-close here could be taken before it is used)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           "--nproc-per-node", str(n), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:
        s_ = ln.strip()
        if s_.startswith("{") and '"metric"' in s_:
            line = s_
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        print("bench.py: the %d ranks ended without a result line" % n, file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks (one per GPU); default: the launcher's WORLD_SIZE, else 1")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-model", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="only the headline step (profiling runs: every dispatch of a kernel is then one of "
                    "the headline shape, so per-dispatch counter averages and dispatch counts belong to it alone)")
    ap.add_argument("--eval-mode", action="store_true", help="dropout as identity (parity-test arithmetic) instead of train mode")
    ap.add_argument("--batch", type=int, default=0, help="override the per-GPU number of sequences (experiments only)")
    ap.add_argument("--profile-steps", type=int, default=5, help="extra eager steps with per-kernel HIP-event timing")
    args = ap.parse_args()
    if args.headline_only:
        args.no_full_model = True

    # Under a launcher (torchrun sets RANK and LOCAL_RANK for every rank) the world size is the launcher's; a WORLD_SIZE that some
    # scheduler preset for a process that is nobody's rank means nothing.
    launched = "RANK" in os.environ and "LOCAL_RANK" in os.environ
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    rank = int(os.environ.get("RANK", "0")) if launched else 0
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
    rehearsal = os.environ.get("MMT_BENCH_REHEARSAL") == "1"      # developer switch: N ranks on ONE GPU over gloo (code-path check only)
    if rehearsal:
        local_rank = 0
    if args.gpus is None:
        args.gpus = world                                   # `torchrun --nproc-per-node N bench.py` without --gpus
    if not launched and args.gpus > 1:
        # Called plainly (`python bench.py --gpus N`): start the N ranks ourselves, as fresh children of a process that has
        # not touched the GPU yet (no HIP call above this line), relay rank 0's JSON line and leave with the launcher's code.
        sys.exit(spawn_ranks(args.gpus))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d was started by a launcher with WORLD_SIZE=%d" % (args.gpus, world))
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from multimodal_transformer_amd import _lib
    from multimodal_transformer_amd.functional import mse_sum_loss, mse_sum_loss_backward
    cfg = dict(WORKLOADS[args.workload])
    if args.batch > 0:
        cfg["B"] = args.batch
        cfg["desc"] += " [batch overridden to %d]" % args.batch
    B, T, d, h, N, f = (cfg[k] for k in ("B", "T", "d", "h", "N", "f"))
    M = B * T
    train = not args.eval_mode
    enc = make_encoder(cfg).to(dev)
    enc.train(train)
    params = list(enc.parameters())
    g = torch.Generator(device="cpu").manual_seed(1 + rank)
    x = torch.randn(B, T, d, generator=g).to(dev).requires_grad_()
    tgt = torch.rand(B, T, d, generator=g).to(dev)
    mask = torch.ones(B, T, 1, device=dev)                   # throughput runs use full-length sequences (§8d)
    nvalid = float(world * B * T)

    calls = [0]                                             # every execution of the headline step in this process (eager or while capturing)

    def fwd_bwd():
        calls[0] += 1
        for p in params:
            p.grad = None
        x.grad = None
        mse_sum_loss_backward(enc(x, mask), tgt, nvalid)

    run = Runner(fwd_bwd, params, world, not args.no_graph, args.warmup)
    elapsed = run.timed(args.steps, dist, dev)
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * M * args.steps / elapsed
    ar_ms = run.allreduce_ms(max(5, args.steps // 2))

    # ---- the same step followed by the reference's optimiser (Adam lr 1e-4, weight decay 1e-4: transformer/SFT/train.py:621);
    #      reported beside the headline, never as it ("with and without Adam + all-reduce")
    adam = None
    if (rank == 0 or world > 1) and not args.headline_only:
        from multimodal_transformer_amd.optim import FlatAdam
        opt = FlatAdam(params, lr=1e-4, weight_decay=1e-4)       # torch.optim.Adam's update as one launch over the flat parameter buffer
        def step_adam():
            run.step()
            opt.step()
        for _ in range(2):
            step_adam()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step_adam()
        torch.cuda.synchronize()
        adam = {"ms_per_step": round(1e3 * (time.perf_counter() - t0) / args.steps, 4),
                "what": "step + Adam (optim.FlatAdam: torch.optim.Adam's update, one launch over the flat buffer of the %d parameter tensors)%s"
                        % (len(params), " + gradient all-reduce" if world > 1 else "")}

    # ---- the same step as two half-batches on two HIP streams inside one graph (Encoder.sub_batch_streams = 2): reported beside the
    #      headline, never as it — the headline's kernels have the GPU to themselves, which is what the roofline line prices
    two_streams = None
    if rank == 0 and world == 1 and B >= 2 and not args.headline_only:
        enc.sub_batch_streams = 2
        try:
            trun = Runner(fwd_bwd, params, 1, not args.no_graph, 3)
            nt_ = max(5, args.steps // 2)
            elt = trun.timed(nt_)
            two_streams = {"what": "the %d sequences as two halves on two HIP streams in one graph (Encoder.sub_batch_streams = 2; the halves "
                                   "draw dropout masks of their own)" % B,
                           "value": round(M * nt_ / elt, 1), "unit": "windows/s", "ms_per_step": round(1e3 * elt / nt_, 4), "launch": trun.launch}
            del trun
        finally:
            enc.sub_batch_streams = 1

    # ---- per-kernel timing (HIP events on the launch stream), eager, outside the timed region
    roofline, kernel_ms = None, {}
    if rank == 0 and args.profile_steps > 0:
        _lib.profile(True)
        for _ in range(args.profile_steps):
            fwd_bwd()
        torch.cuda.synchronize()
        prof = _lib.profile_collect()
        _lib.profile(False)
        kernel_ms, roofline = roofline_of(prof, args.profile_steps, M, T, d, f, N, train, cfg, top=99)

    # ---- whole SFT sequence model, same batch shape (N=1 only)
    full = None
    if rank == 0 and world == 1 and not args.no_full_model and args.workload in ("C4", "C2"):
        from multimodal_transformer_amd import multiTransformer as MT
        torch.manual_seed(1)
        model = MT.NLPTransformer(512, embed_dim=d, h=h, N=N, d_ff=f, dropout=DROPOUT, device=dev)
        model.train(train)
        mparams = list(model.parameters())
        xin = torch.tanh(torch.randn(B, T, 512, generator=g)).to(dev)      # post-fusion tanh features (SFT/models.py:138)
        tgt1 = torch.rand(B, T, 1, generator=g).to(dev)
        lengths = [T] * B

        def model_step():
            for p in mparams:
                p.grad = None
            mse_sum_loss_backward(model(xin, mask, lengths), tgt1, B * T)

        mrun = Runner(model_step, mparams, 1, not args.no_graph, 3)
        el = mrun.timed(max(5, args.steps // 2))
        nst = max(5, args.steps // 2)
        _lib.profile(True)
        for _ in range(3):
            model_step()
        torch.cuda.synchronize()
        mp_ = _lib.profile_collect()
        _lib.profile(False)
        full = {"model": "NLPTransformer(512, embed_dim=%d, h=%d): Dropout+Linear+ReLU embed, encoder, LSTM decoder, MLP, mask" % (d, h),
                "value": round(M * nst / el, 1), "unit": "windows/s", "ms_per_step": round(1e3 * el / nst, 4), "launch": mrun.launch,
                "kernel_ms_per_step": {k: round(v[0] / 3, 4) for k, v in sorted(mp_.items(), key=lambda kv: -kv[1][0])[:8]}}

    # ---- raw windows -> valence: the 3-modality SFT pipeline of transformer/SFT/models.py (CNN k=2 + max-pool + Highway per
    #      modality, fusion, NLPTransformer) at the configs[3] per-GPU slice, N=1 only.  d_model = 128 as in configs[3].
    pipeline = None
    if rank == 0 and world == 1 and not args.no_full_model and args.workload == "C4":
        from multimodal_transformer_amd import models as MM, multiTransformer as MT
        torch.manual_seed(1)
        pm = ["acoustic", "image", "linguistic"]
        pd = {"acoustic": 88, "image": 1000, "linguistic": 300}          # transformer/SFT/train.py:534
        pw = {"acoustic": 10, "image": 30, "linguistic": 33}             # frames / tokens per window
        pmodel = MM.MultiCNNTransformer(pm, pd, device=dev)
        pmodel.Transformer = MT.NLPTransformer(512, embed_dim=d, h=h, N=N, d_ff=f, dropout=DROPOUT, device=dev)
        pmodel.train(train)
        pparams = list(pmodel.parameters())
        praw = {m: torch.randn(B, T, pw[m], pd[m], generator=g).to(dev) for m in pm}
        ptgt = torch.rand(B, T, 1, generator=g).to(dev)

        def pipe_step():
            for p in pparams:
                p.grad = None
            mse_sum_loss_backward(pmodel(praw, [T] * B, mask), ptgt, B * T)

        prun = Runner(pipe_step, pparams, 1, not args.no_graph, 3)
        pn = max(5, args.steps // 3)
        pel = prun.timed(pn)
        os.environ["MMT_MODALITY_STREAMS"] = "0"           # per-kernel event times are only meaningful without concurrent streams
        _lib.profile(True)
        for _ in range(3):
            pipe_step()
        torch.cuda.synchronize()
        pp = _lib.profile_collect()
        _lib.profile(False)
        os.environ.pop("MMT_MODALITY_STREAMS", None)
        conv_flop = sum(2.0 * M * (pw[m] - 1) * 2 * pd[m] * pmodel.window_embed_size[m] for m in pm)
        cf = pp.get("convpool_fwd_kernel", (0.0, 1))
        pipeline = {"model": "MultiCNNTransformer(acoustic 10x88, image 30x1000, linguistic 33x300 raw windows -> CNN+max-pool+Highway -> "
                             "fusion 812->512 -> NLPTransformer(512, embed_dim=%d)); raw inputs fp32, %.2f GB resident in HBM" %
                             (d, sum(v.numel() for v in praw.values()) * 4 / 1e9),
                    "value": round(M * pn / pel, 1), "unit": "windows/s", "ms_per_step": round(1e3 * pel / pn, 4), "launch": prun.launch,
                    "conv_fwd": {"ms_per_step": round(cf[0] / 3, 4), "algorithmic_tflops": round(conv_flop / (cf[0] / 3 * 1e-3) / 1e12, 1),
                                 "mfma_frac": round(conv_flop / (cf[0] / 3 * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
                                 "what": "3 launches/step (one per modality), timed with the modality streams serialised: 2*(W-1)*2D*F FLOP per window"},
                    "kernel_ms_per_step": {k: round(v[0] / 3, 4) for k, v in sorted(pp.items(), key=lambda kv: -kv[1][0])[:6]}}
        del praw, pmodel, prun
        torch.cuda.empty_cache()

    # ---- whole MFT model at configs[2] (3 modalities, T=300, 32 sequences), N=1 only
    mft = None
    if rank == 0 and world == 1 and not args.no_full_model and args.workload == "C4":
        from multimodal_transformer_amd import multiTransformer as MT
        torch.manual_seed(1)
        mods = ["acoustic", "image", "linguistic"]                       # transformer/MFT/train.py default modality set order
        dims = {"acoustic": 88, "image": 256, "linguistic": 300}
        Bm, Tm = 32, 300
        mmodel = MT.MultiTransformer(mods, dims, device=dev)
        mmodel.train(train)
        mparams2 = list(mmodel.parameters())
        xin2 = {m: torch.randn(Bm, Tm, dims[m], generator=g).to(dev) for m in mods}
        mask2 = torch.ones(Bm, Tm, 1, device=dev)
        tgt2 = torch.rand(Bm, Tm, 1, generator=g).to(dev)

        def mft_step():
            for p in mparams2:
                p.grad = None
            mse_sum_loss_backward(mmodel(xin2, mask2, [Tm] * Bm), tgt2, Bm * Tm)

        nst2 = max(5, args.steps // 2)
        mrun2 = Runner(mft_step, mparams2, 1, not args.no_graph, 3)
        el2 = mrun2.timed(nst2)
        if mrun2.launch == "hipgraph":
            # the three modality encoders run on concurrent streams; hipGraph replay serialises part of that concurrency on
            # ROCm 7.2, so the eager launch can be the faster one: time both, report the better with its label
            mrun2e = Runner(mft_step, mparams2, 1, False, 2)
            el2e = mrun2e.timed(nst2)
            if el2e < el2:
                el2, mrun2 = el2e, mrun2e
        mft = {"model": "MultiTransformer(acoustic 88, image 256, linguistic 300 -> 256): 3 embeds + 3 encoder stacks (d=256, h=8, N=6) "
                        "on concurrent streams + MFN gate; T=300, 32 sequences (configs[2])",
               "value": round(Bm * Tm * nst2 / el2, 1), "unit": "windows/s", "ms_per_step": round(1e3 * el2 / nst2, 4), "launch": mrun2.launch}
        fpw3 = 3 * 3 * 6 * flops_per_window_layer_fwd(256, Tm, 128) + 3 * 1.35e6                  # three stacks + MFN gate
        mft["algorithmic_mflop_per_window"] = round(fpw3 / 1e6, 2)
        mft["step_mfma_frac"] = round(mft["value"] * fpw3 / (MFMA_BF16_PEAK_TFLOPS * 1e12), 5)
        if args.profile_steps > 0:          # per-kernel times (modality streams serialised); a launch covers ONE modality's stack
            mft["kernel_ms_per_step"], mft["roofline"] = roofline_of(profiled(mft_step, 2), 2, Bm * Tm, Tm, 256, 128, 6, train, WORKLOADS["C3e"])

    # ---- configs[3] at its full batch (256 sequences) on one GPU
    full_batch = None
    if rank == 0 and world == 1 and not args.no_full_model and args.workload == "C4" and args.batch == 0:
        Bf = 256
        xf = torch.randn(Bf, T, d, generator=g).to(dev).requires_grad_()
        tgtf = torch.rand(Bf, T, d, generator=g).to(dev)
        maskf = torch.ones(Bf, T, 1, device=dev)

        def full_step():
            for p in params:
                p.grad = None
            xf.grad = None
            mse_sum_loss_backward(enc(xf, maskf), tgtf, Bf * T)

        frun = Runner(full_step, params, 1, not args.no_graph, 2)
        nf = max(5, args.steps // 2)
        elf = frun.timed(nf)
        vf = Bf * T * nf / elf
        full_batch = {"workload": "SFT encoder stack T=500 d_model=128 heads=8 N=6 d_ff=128, 256 sequences on ONE GPU (configs[3] whole batch)",
                      "value": round(vf, 1), "unit": "windows/s", "ms_per_step": round(1e3 * elf / nf, 4), "launch": frun.launch,
                      "step_mfma_frac": round(vf * 3 * N * flops_per_window_layer_fwd(d, T, f) / (MFMA_BF16_PEAK_TFLOPS * 1e12), 5)}
        enc.sub_batch_streams = 2
        try:
            frun2 = Runner(full_step, params, 1, not args.no_graph, 2)
            elf2 = frun2.timed(nf)
            full_batch["two_streams"] = {"value": round(Bf * T * nf / elf2, 1), "ms_per_step": round(1e3 * elf2 / nf, 4), "launch": frun2.launch}
            del frun2
        finally:
            enc.sub_batch_streams = 1
        del xf, tgtf, frun
        torch.cuda.empty_cache()

    # ---- whole MFT model at the configs[4] per-GPU slice (3 modalities, T=1000, 64 sequences, d=256), N=1 only
    mft4 = None
    if rank == 0 and world == 1 and not args.no_full_model and args.workload == "C4":
        from multimodal_transformer_amd import multiTransformer as MT
        torch.manual_seed(1)
        mods = ["acoustic", "image", "linguistic"]
        dims = {"acoustic": 88, "image": 256, "linguistic": 300}
        B4, T4 = 64, 1000
        m4 = MT.MultiTransformer(mods, dims, device=dev)
        m4.train(train)
        p4 = list(m4.parameters())
        x4 = {m: torch.randn(B4, T4, dims[m], generator=g).to(dev) for m in mods}
        mask4 = torch.ones(B4, T4, 1, device=dev)
        tgt4 = torch.rand(B4, T4, 1, generator=g).to(dev)

        def mft4_step():
            for p in p4:
                p.grad = None
            mse_sum_loss_backward(m4(x4, mask4, [T4] * B4), tgt4, B4 * T4)

        n4 = 5
        r4 = Runner(mft4_step, p4, 1, not args.no_graph, 2)
        el4 = r4.timed(n4)
        if r4.launch == "hipgraph":
            r4e = Runner(mft4_step, p4, 1, False, 1)
            el4e = r4e.timed(n4)
            if el4e < el4:
                el4, r4 = el4e, r4e
        v4 = B4 * T4 * n4 / el4
        fpw4 = 3 * 3 * 6 * flops_per_window_layer_fwd(256, T4, 128) + 3 * 1.35e6          # three stacks + MFN gate
        mft4 = {"model": "MultiTransformer(acoustic 88, image 256, linguistic 300 -> 256): 3 embeds + 3 encoder stacks (d=256, h=8, N=6) + "
                         "MFN gate; T=1000, 64 sequences (configs[4] per-GPU slice)",
                "value": round(v4, 1), "unit": "windows/s", "ms_per_step": round(1e3 * el4 / n4, 4), "launch": r4.launch,
                "algorithmic_mflop_per_window": round(fpw4 / 1e6, 2),
                "step_mfma_frac": round(v4 * fpw4 / (MFMA_BF16_PEAK_TFLOPS * 1e12), 5)}
        if args.profile_steps > 0:
            mft4["kernel_ms_per_step"], mft4["roofline"] = roofline_of(profiled(mft4_step, 2), 2, B4 * T4, T4, 256, 128, 6, train, WORKLOADS["C5e"])
        del x4, m4, r4
        torch.cuda.empty_cache()

    if rank == 0:
        fpw = 3 * N * flops_per_window_layer_fwd(d, T, f)
        out = {
            "metric": "windows/sec fwd+bwd", "value": round(value, 1), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": cfg["desc"] + "; fwd + MSE loss + bwd (input, weight, LayerNorm grads)"
                       + ("; + RCCL SUM all-reduce of gradients" if world > 1 else ""),
                       "name": args.workload, "global_batch": world * B, "seq_len": T, "d_model": d, "heads": h, "layers": N, "d_ff": f,
                       "parallelism": "dp%d" % world,
                       "dropout": ("train mode, p=%.2f at the reference's four encoder sites, in-kernel generator" % DROPOUT) if train
                       else "eval mode (identity)",
                       "lengths": "full"},
            "launch": run.launch,
            "algorithmic_mflop_per_window": round(fpw / 1e6, 3),
            "step_mfma_frac": round(value * fpw / (world * MFMA_BF16_PEAK_TFLOPS * 1e12), 5),
            "roofline": roofline,
            "step_hbm": step_hbm(train, cfg, ms_per_step),
            "fwd_bwd_calls": calls[0],
            "kernel_ms_per_step": kernel_ms,
        }
        if ar_ms is not None:
            out["allreduce_ms"] = round(ar_ms, 4)
            out["allreduce"] = run.exchange
            out["allreduce_what"] = "mean device time of the gradient SUM all-reduce alone (HIP events around it, after the graph replay)"
        if adam is not None:
            out["with_adam"] = adam
        if two_streams is not None:
            out["two_streams"] = two_streams
        if full_batch is not None:
            out["config_full_batch"] = full_batch
        if mft4 is not None:
            out["mft_configs4"] = mft4
        if full is not None:
            out["full_model"] = full
        if pipeline is not None:
            out["raw_pipeline"] = pipeline
        if mft is not None:
            out["mft_model"] = mft
        if not args.no_cpu_baseline and world == 1:        # N = 1 only: the other ranks must not wait on 15 s of CPU work
            out["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
