"""ctypes binding of libmmt_hip.so (include/mmt_hip.h).

There is no CPU fallback: if the library cannot be loaded, or a tensor is not on a HIP
device, the callers raise.  The library is built in-tree by ``build.py`` (hipcc, gfx950).
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MMT_LIB_PATH") or os.path.join(_HERE, "libmmt_hip.so")     # MMT_LIB_PATH: developer builds (tools/)

_c = ctypes
_P, _F, _I, _SZ, _U64 = _c.c_void_p, _c.c_float, _c.c_int, _c.c_size_t, _c.c_uint64

# name -> (restype, argtypes); mirrors include/mmt_hip.h one to one
SIGNATURES = {
    "mmt_abi_version": (_I, []),
    "mmt_last_error": (_c.c_char_p, []),
    "mmt_profile_enable": (_I, [_I]),
    "mmt_profile_reset": (_I, []),
    "mmt_profile_num_sites": (_I, []),
    "mmt_profile_site_name": (_c.c_char_p, [_I]),
    "mmt_profile_collect": (_I, [_P, _P]),
    "mmt_encoder_param_count": (_SZ, [_I, _I, _I]),
    "mmt_encoder_workspace_bytes": (_SZ, [_I] * 6),
    "mmt_encoder_workspace_bytes_eval": (_SZ, [_I] * 6),
    "mmt_encoder_forward": (_I, [_P, _P, _P, _P, _P, _SZ] + [_I] * 6 + [_F, _F, _U64, _P]),
    "mmt_encoder_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _SZ] + [_I] * 6 + [_F, _F, _U64, _P]),
    "mmt_encoder_forward_devseed": (_I, [_P, _P, _P, _P, _P, _SZ] + [_I] * 6 + [_F, _F, _P, _P]),
    "mmt_encoder_backward_devseed": (_I, [_P, _P, _P, _P, _P, _P, _P, _SZ] + [_I] * 6 + [_F, _F, _P]),
    "mmt_layernorm_scratch_floats": (_SZ, [_I, _I]),
    "mmt_layernorm_forward": (_I, [_P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "mmt_layernorm_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P]),
    "mmt_sdpa_workspace_bytes": (_SZ, [_I] * 4),
    "mmt_sdpa_workspace_bytes_eval": (_SZ, [_I] * 4),
    "mmt_sdpa_forward": (_I, [_P, _P, _P, _P, _P, _P, _SZ] + [_I] * 4 + [_F, _U64, _P]),
    "mmt_sdpa_backward": (_I, [_P, _P, _P, _P, _P, _P, _SZ] + [_I] * 4 + [_F, _U64, _P]),
    "mmt_linear_workspace_bytes": (_SZ, [_I] * 3),
    "mmt_linear_forward": (_I, [_P, _P, _P, _P, _P, _P, _SZ] + [_I] * 4 + [_P]),
    "mmt_linear_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ] + [_I] * 4 + [_P]),
    "mmt_linear_dropout_forward": (_I, [_P, _P, _P, _P, _P, _P, _SZ] + [_I] * 4 + [_F, _F, _U64, _P, _P]),
    "mmt_linear_dropout_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _SZ] + [_I] * 4 + [_F, _F, _U64, _I, _P]),
    "mmt_copy2d": (_I, [_P, _I, _P]),
    "mmt_softmax_mul_forward": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "mmt_softmax_mul_backward": (_I, [_P, _P, _P, _P, _P, _I, _I, _P]),
    "mmt_colsum": (_I, [_P, _P, _I, _I, _I, _P]),
    "mmt_highway_forward": (_I, [_P, _P, _P, _P, _SZ, _F, _U64, _P, _P, _P]),
    "mmt_highway_backward": (_I, [_P, _P, _P, _P, _P, _P, _P, _SZ, _F, _U64, _P, _P]),
    "mmt_error_accumulate": (_I, [_P, _P, _P]),
    "mmt_lstm_scan_workspace_bytes": (_SZ, [_I]),
    "mmt_lstm_scan_forward": (_I, [_P] * 8 + [_SZ] + [_I] * 3 + [_P]),
    "mmt_lstm_scan_backward": (_I, [_P] * 10 + [_SZ] + [_I] * 3 + [_P]),
    "mmt_mfn_mem_scan_workspace_bytes": (_SZ, []),
    "mmt_mfn_mem_scan_forward": (_I, [_P] * 9 + [_SZ] + [_I] * 4 + [_F, _U64, _P]),
    "mmt_mfn_mem_scan_forward_devseed": (_I, [_P] * 9 + [_SZ] + [_I] * 4 + [_F, _P, _P]),
    "mmt_mfn_mem_scan_backward": (_I, [_P] * 11 + [_SZ] + [_I] * 4 + [_F, _P]),
    "mmt_convpool_workspace_bytes": (_SZ, [_I] * 4),
    "mmt_convpool_forward": (_I, [_P] * 6 + [_SZ] + [_I] * 4 + [_P]),
    "mmt_convpool_backward": (_I, [_P] * 6 + [_SZ] + [_I] * 4 + [_P]),
    "mmt_mse_sum_scratch_doubles": (_SZ, [_SZ]),
    "mmt_mse_sum_forward": (_I, [_P, _P, _F, _P, _P, _P, _SZ, _P]),
    "mmt_adam_step": (_I, [_P, _P, _P, _P, _P, _I, _F, _F, _F, _F, _F, _I, _P]),
    "mmt_ccc_forward": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "mmt_debug_dropout_mask": (_I, [_F, _U64, _c.c_uint32, _U64, _c.c_uint32, _P, _P, _P]),
    "mmt_debug_poison_lds": (_I, [_c.c_uint32, _P, _P]),
}



class CopySeg(ctypes.Structure):
    """mmt_copy_seg of include/mmt_hip.h"""
    _fields_ = [("src", _P), ("src2", _P), ("dst", _P), ("rowscale", _P), ("rows", _I), ("cols", _I), ("src_ld", _I), ("src2_ld", _I),
                ("dst_ld", _I), ("perm", _I), ("pB", _I), ("pT", _I), ("accumulate", _I)]


_lib = None
_lock = threading.Lock()


def load():
    """Load (building first if the .so is absent and hipcc exists).  Raises RuntimeError otherwise."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            from . import build as _build
            try:
                _build.build(verbose=False)
            except Exception as e:  # noqa: BLE001
                raise RuntimeError("libmmt_hip.so is missing and could not be built with hipcc: %s" % e)
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise RuntimeError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return _lib


def check(rc):
    if rc != 0:
        msg = load().mmt_last_error()
        raise RuntimeError("libmmt_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def require_hip(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("multimodal_transformer_amd runs on MI355X only: got a %s tensor; there is no CPU path "
                               "(move the module and its inputs to a HIP device)" % t.device)


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream


SIDE_LANES = {}       # raw stream handle -> lane id (>= 1); filled by StreamFork


class StreamFork:
    """n independent pieces of work on n HIP streams (the first on the caller's stream), forked and joined by event
    (``wait_stream``), which is also legal inside hipGraph capture; autograd replays each backward node on the stream of its
    forward.  Every side stream owns a workspace lane (see WorkspacePool).  Users: the modalities of the MFT / of the window
    encoder (independent until the MFN gate / the fusion layer) and the sub-batches of one encoder stack (functional.encoder_stack)."""

    def __init__(self, env_switch=None):
        self._side = {}
        self._env = env_switch

    def begin(self, device, n):
        main = torch.cuda.current_stream(device)
        if n < 2 or (self._env and os.environ.get(self._env, "1") == "0"):
            return main, [main] * n
        side = self._side.setdefault(str(device), [])
        while len(side) < n - 1:
            side.append(torch.cuda.Stream(device=device))
            SIDE_LANES[int(side[-1].cuda_stream)] = len(SIDE_LANES) + 1       # own workspace lane (see WorkspacePool)
        streams = [main] + side[:n - 1]
        for s in streams[1:]:
            s.wait_stream(main)
        return main, streams

    @staticmethod
    def end(main, streams, tensors=()):
        for s in set(streams):
            if s is not main:
                main.wait_stream(s)
        for t in tensors:
            t.record_stream(main)


def on_side_lane(device):
    """True while the current stream is one of StreamFork's side streams (work that is already one of several concurrent pieces)."""
    return int(torch.cuda.current_stream(device).cuda_stream) in SIDE_LANES


class WorkspacePool:
    """Zero-initialised device workspaces, reused per (device, stream lane, size).

    The kernels rely on pad regions of a workspace staying zero; they never write them, so a buffer
    can be reused for the same shape without clearing.  A buffer is held by the autograd node between
    forward and backward and handed back afterwards.  Lanes: the side streams on which the per-modality
    encoders of the MFT run concurrently are registered in SIDE_LANES and each owns its buffers; every
    other stream (the caller's, a hipGraph capture stream) is lane 0 with the usual in-order semantics.
    So concurrent streams are never handed a workspace another stream's kernels may still be using, and a
    graph capture after warm-up finds its buffers in the pool instead of allocating (and re-zeroing) them.
    """

    def __init__(self):
        self._free = {}
        self._home = {}
        self._mutex = threading.Lock()

    def get(self, nbytes, device, tag=None):
        """``tag`` names the call site AND its shape: two shapes can need the same number of bytes with a different internal
        layout (measured: (B=4,T=50) and (B=2,T=100) encoder workspaces are both 6,158,080 bytes), and the zero pads of one
        would then be live data of the other.  Every caller passes one."""
        key = (str(device), SIDE_LANES.get(int(torch.cuda.current_stream(device).cuda_stream), 0), int(nbytes), tag)
        with self._mutex:
            lst = self._free.get(key)
            buf = lst.pop() if lst else None
        if buf is None:
            buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
        with self._mutex:
            self._home[buf.data_ptr()] = key
        return buf

    def put(self, buf):
        with self._mutex:
            key = self._home.get(buf.data_ptr())
            if key is not None:
                self._free.setdefault(key, []).append(buf)

    def clear(self):
        with self._mutex:
            self._free.clear()
            self._home.clear()


POOL = WorkspacePool()


class DeviceErrorWatch:
    """Device error words of asynchronous kernels (today: the exchange time-out of the four-CU LSTM scans,
    include/mmt_hip.h mmt_lstm_scan_*).  Eager launches: ``watch`` copies the word to a pinned host slot behind the launch; ``poll``
    (called at every later eager launch, never blocking) and ``check`` (blocking) raise if a drained launch left a non-zero word.
    Inside a hipGraph capture the word is OR-ed into a persistent device word by a kernel of the captured sequence
    (``mmt_error_accumulate``; a device-to-host copy node was seen to run out of order in the replayed graph), and ``check`` reads
    that word after synchronising.  The persistent word must exist before a capture starts: ``prepare`` makes it, and so does any eager
    ``watch`` — a warm-up step."""

    SLOTS = 2048                    # ring of pinned host words for eager launches

    def __init__(self):
        self._pending = []          # (slot, event, description)
        self._mutex = threading.Lock()
        self._host = None
        self._accum = {}            # device -> persistent int32 word for captured launches
        self._captured = []         # descriptions of the captured launches
        self._next = 0

    def prepare(self, device=None):
        with self._mutex:
            if self._host is None:
                self._host = torch.zeros(self.SLOTS, dtype=torch.int32).pin_memory()
            if device is not None and str(device) not in self._accum:
                self._accum[str(device)] = torch.zeros(1, dtype=torch.int32, device=device)

    def watch(self, dev_word, what):
        if torch.cuda.is_current_stream_capturing():
            acc = self._accum.get(str(dev_word.device))
            if acc is None:
                return              # nothing ran eagerly before the capture: this launch stays unwatched
            check(load().mmt_error_accumulate(ptr(dev_word), ptr(acc), stream_ptr()))
            with self._mutex:
                if what not in self._captured:
                    self._captured.append(what)
            return
        self.prepare(dev_word.device)
        with self._mutex:
            slot, self._next = self._next, (self._next + 1) % self.SLOTS
            lapped = any(e[0] == slot for e in self._pending)                    # the slot's previous launch has not been looked at yet
        if lapped:
            self.check()
        self._host[slot:slot + 1].copy_(dev_word, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        with self._mutex:
            self._pending.append((slot, ev, what))
        self.poll()

    @staticmethod
    def _raise(what):
        raise RuntimeError("libmmt_hip device error: %s timed out waiting for a partner workgroup (CUs held by another stream "
                           "or process?); its outputs are invalid.  MMT_NO_CLUSTER_SCAN=1 selects the one-CU scan." % what)

    def poll(self):
        with self._mutex:
            done, rest = [], []
            for e in self._pending:
                (done if e[1].query() else rest).append(e)
            self._pending = rest
        for slot, _, what in done:
            if int(self._host[slot]) != 0:
                self._raise(what)

    def check(self):
        """Synchronise the device and raise if any watched launch — eager, or captured and replayed from a hipGraph — reported an error."""
        torch.cuda.synchronize()
        with self._mutex:
            entries, self._pending = self._pending, []
            accs = list(self._accum.values())
        for slot, _, what in entries:
            if int(self._host[slot]) != 0:
                self._raise(what)
        for acc in accs:
            if int(acc.item()) != 0:
                acc.zero_()
                self._raise("a launch replayed from a hipGraph (%s)" % "; ".join(self._captured[-4:]))


ERRORS = DeviceErrorWatch()

_seed_fallback_counter = [0]


def mix64(*words):
    """splitmix64 chain over integer words -> 63-bit seed (host side; the kernels mix it again per stream)."""
    z = 0x243F6A8885A308D3
    for w in words:
        z = (z + (int(w) & 0xFFFFFFFFFFFFFFFF) + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
    return z & 0x7FFFFFFFFFFFFFFF


class DeviceSeed:
    """A 64-bit dropout seed that lives in device memory (include/mmt_hip.h mmt_encoder_forward_devseed): the forward's first kernel
    reads it, leaves it in the workspace for its backward and advances it, so a training step replayed from a hipGraph draws fresh
    masks at every replay — a seed passed by value would be frozen at capture.  ``peek()`` = the seed the NEXT forward will use
    (``functional.dropout_mask(p, seed, stream, ...)`` replays that step's masks)."""

    def __init__(self, device, value):
        self.state = torch.tensor([int(value) & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=device)

    def peek(self):
        return int(self.state.item()) & 0xFFFFFFFFFFFFFFFF

    def ptr(self):
        return self.state.data_ptr()


def canonical_device(device):
    """torch.device with its index filled in: `torch.device("cuda")`, "cuda" and `cuda:0` name the same device, but compare unequal"""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def _host_seed(device, site):
    rank = 0
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            rank = dist.get_rank()
    except Exception:  # noqa: BLE001
        rank = 0
    base, off = torch.initial_seed(), None
    try:
        if device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
            base, off = gen.initial_seed(), gen.get_offset()
            gen.set_offset(off + 4)                 # philox offsets advance in multiples of 4
    except Exception:  # noqa: BLE001
        off = None
    if off is None:
        _seed_fallback_counter[0] += 1
        off = (1 << 40) + _seed_fallback_counter[0]
    return mix64(base, off, rank, site)


def next_dropout_seed(device, site, holder=None, index=0):
    """Seed of one train-mode forward of one module.

    Eager launches: a python int drawn, like torch's own dropout kernels, from the DEVICE generator (seed, offset), advancing the offset:
    every call of every module instance gets fresh masks, a run restored with its RNG state continues its mask sequence, and
    torch.manual_seed() controls it.  The data-parallel rank is mixed in: ranks seeded alike still drop different units.

    During hipGraph capture (or with MMT_DEVICE_SEED=1): the module's ``DeviceSeed`` — a seed in device memory that the captured
    kernels read and advance, so every REPLAY draws new masks.  It is created at the module's first eager train-mode call (seeded
    from the generator as above); capturing a module that never ran eagerly raises (a warm-up step always precedes a capture)."""
    device = canonical_device(device)
    capturing = device.type == "cuda" and torch.cuda.is_current_stream_capturing()
    if os.environ.get("MMT_DEVICE_SEED") == "0":         # developer switch: by-value seeds even under capture (frozen at capture, as in round 2)
        holder = None
    if holder is not None:
        # one state per dropout site of the module AND per concurrent piece of it (`index`: sub-batch stream): two sites that shared a
        # state word would be correct only while their launches happen to sit on one stream
        seeds = holder.__dict__.setdefault("_dev_seeds", {})
        ds = seeds.get((site, index))
        if ds is None or canonical_device(ds.state.device) != device:
            if capturing:
                raise RuntimeError("hipGraph capture of a train-mode %s before its first eager call: run one warm-up step so that its "
                                   "device-resident dropout seed exists" % type(holder).__name__)
            ds = DeviceSeed(device, _host_seed(device, mix64(site, index)))
            seeds[(site, index)] = ds
        if capturing or os.environ.get("MMT_DEVICE_SEED") == "1":
            return ds
    elif capturing:
        _seed_fallback_counter[0] += 1
        return mix64(torch.initial_seed(), (1 << 40) + _seed_fallback_counter[0], site)     # frozen under replay: stand-alone attention() only
    return _host_seed(device, mix64(site, index) if index else site)


def device_seed(holder, site=None, index=0):
    """The ``DeviceSeed`` of one dropout site of a module (sites: 1 encoder stack, 2 / 4 the MFN's gamma and output dropouts, 5 the SFT
    embedding's input dropout, 6 the window encoder's Dropout(0.3)).  ``site=None`` is accepted only where the module has ONE site."""
    seeds = holder.__dict__.get("_dev_seeds", {})
    if site is None:
        sites = sorted({s for (s, _) in seeds})
        if len(sites) != 1:
            raise KeyError("%s has dropout sites %s: name one" % (type(holder).__name__, sites))
        site = sites[0]
    return seeds[(site, index)]


def profile(on):
    lib = load()
    lib.mmt_profile_reset()
    lib.mmt_profile_enable(1 if on else 0)


def profile_collect():
    """-> {site name: (total_ms, launches)} for every site that ran since the last reset."""
    lib = load()
    n = lib.mmt_profile_num_sites()
    ms = (ctypes.c_float * n)()
    cnt = (ctypes.c_int * n)()
    check(lib.mmt_profile_collect(ctypes.cast(ms, ctypes.c_void_p), ctypes.cast(cnt, ctypes.c_void_p)))
    return {lib.mmt_profile_site_name(i).decode(): (float(ms[i]), int(cnt[i])) for i in range(n) if cnt[i]}
