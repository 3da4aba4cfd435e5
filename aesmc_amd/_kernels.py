"""Host-side launcher for the HIP kernels: validates operands, allocates outputs, enqueues on the
caller's current HIP stream through the C ABI (include/aesmc_hip.h).

Every operand is checked here against what the kernel and its grid assume (device, dtype, shape,
density, alignment) before anything is launched.  Kernels report data-dependent conditions (NaN
log-weights, degenerate rows, out-of-range indices) by OR-ing bits into a per-device int32 word
that the host reads once per ELBO evaluation (`read_flags`) instead of synchronising per timestep.
"""
import contextlib
import ctypes
import threading

import torch

from . import _lib
from . import settings

_DTYPE_TAG = {torch.float32: _lib.F32, torch.float64: _lib.F64}


def _require_hip(t, what):
    if not isinstance(t, torch.Tensor):
        raise AttributeError("{} must be a torch.Tensor. Got: {}".format(what, type(t)))
    if not t.is_cuda:
        raise RuntimeError(
            "aesmc_amd: {} lives on '{}'; this package computes only on a HIP device (MI355X) "
            "and has no CPU fallback.".format(what, t.device))


def _tag(t, what):
    try:
        return _DTYPE_TAG[t.dtype]
    except KeyError:
        raise TypeError("aesmc_amd: {} must be float32 or float64, got {}".format(what, t.dtype))


def _inner_dense(t):
    """True when dims 2.. of `t` are laid out densely (row payload is one contiguous run)."""
    expect = 1
    for size, stride in zip(reversed(t.shape[2:]), reversed(t.stride()[2:])):
        if size != 1 and stride != expect:
            return False
        expect *= size
    return True


def _ptr(t):
    return 0 if t is None else t.data_ptr()


_NO_SWITCH = contextlib.nullcontext()


def _on_device(device):
    """Context that makes `device` current for the launch; free when it already is (the usual
    case: entering torch.cuda.device costs ~4 us of host time per kernel in the eager loop)."""
    if torch.cuda.current_device() == device.index:
        return _NO_SWITCH
    return torch.cuda.device(device)


class KernelTimer:
    """Optional per-kernel timing for bench.py's `roofline` leg.  While installed on the provider
    it keeps a bounded random sample of the launches made (the C-ABI entry point with its operands);
    `summary()` then captures each kernel's sample, launched back to back, into one hipGraph and
    times replays of it between two HIP events on the replaying stream: kernel time on the real
    operands without host gaps (a 9 us kernel cannot be timed through ~10 us Python launches).  The
    sample cycles through distinct operands, so caches are as cold as in the workload.  `bytes` is
    the ALGORITHMIC traffic of a launch: what the operation must move, not
    what the hardware happened to move."""

    def __init__(self, keep=24, seed=0):
        import random
        self.keep = keep
        self.random = random.Random(seed)
        self.launches = {}   # name -> [count, [(fn, nbytes, keepalive), ...]]

    def note(self, name, fn, nbytes, keepalive):
        """`fn` = (entry point, argument tuple whose LAST element is the stream)."""
        entry = self.launches.setdefault(name, [0, []])
        entry[0] += 1
        if len(entry[1]) < self.keep:
            entry[1].append((fn, nbytes, keepalive))
        else:  # reservoir sampling keeps a uniform sample of all launches seen
            slot = self.random.randrange(entry[0])
            if slot < self.keep:
                entry[1][slot] = (fn, nbytes, keepalive)

    def summary(self, repeats=3):
        provider = get()
        provider.timer = None       # the replays (and the K7 call below) must not be noted again
        out = {}
        for name, (count, sample) in list(self.launches.items()):
            def launch_all():
                stream = torch.cuda.current_stream().cuda_stream
                for (entry, args), _, _ in sample:
                    entry(*(args[:-1] + (stream,)))
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):      # warm-up off the default stream, as capture requires
                launch_all()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread_local: with a process group alive RCCL's watchdog thread may call the HIP runtime
            # at any time; that must not invalidate this capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                launch_all()
            graph.replay()
            torch.cuda.synchronize()
            begin = torch.cuda.Event(enable_timing=True)
            end = torch.cuda.Event(enable_timing=True)
            begin.record()
            for _ in range(repeats):
                graph.replay()
            end.record()
            torch.cuda.synchronize()
            del graph
            seconds = begin.elapsed_time(end) * 1e-3 / (repeats * len(sample))
            nbytes = sum(n for _, n, _ in sample) / len(sample)
            out[name] = {"launches": count, "sampled": len(sample), "avg_us": 1e6 * seconds,
                         "bytes_per_launch": nbytes, "GBps": nbytes / seconds / 1e9}
            if name in ("resample_gather", "resample_step", "affine_normal_propagate_resampled",
                        "affine_normal_propagate_drawn"):
                # The algorithmic figure counts a full read of the source (and, for the fused step,
                # K3's re-read of the indices, which it skips); only rows that still have offspring
                # are actually fetched.  Report how many that was on these operands and the bytes
                # that had to move.
                fractions, moved, ess = [], [], []
                for _, _, keep in sample:
                    if name in ("affine_normal_propagate_resampled", "affine_normal_propagate_drawn"):
                        # (x_src, ancestors, eps, y, lw, x_t, ...): the gather's read side is the surviving rows
                        idx, dst = keep[1], keep[5]
                        if idx is None:
                            continue
                        rows = idx.numel()
                        unique = int((idx[:, 1:] != idx[:, :-1]).sum().item()) + idx.size(0)
                        payload = dst.numel() * dst.element_size() / rows
                        fractions.append(unique / rows)
                        noise_rows = 2 if name == "affine_normal_propagate_resampled" else 1     # eps read + x_t written
                        moved.append(rows * (8 + dst.element_size() + noise_rows * payload) + unique * payload)
                        continue
                    idx, dst = (keep[1], keep[2]) if name == "resample_gather" else (keep[2], keep[5])
                    if dst is None:
                        continue
                    if name == "resample_step":   # effective sample size of the weights resampled from (K7)
                        log_ess = get().particle_summary(keep[0], want_log_ess=True)[0]
                        ess.append(float(torch.exp(log_ess.double()).mean().item()) / idx.size(1))
                    rows = idx.numel()
                    unique = int((idx[:, 1:] != idx[:, :-1]).sum().item()) + idx.size(0)
                    payload = dst.numel() * dst.element_size() / rows
                    fractions.append(unique / rows)
                    fixed = rows * 8 if name == "resample_gather" else rows * (keep[0].element_size() + 8)
                    moved.append(fixed + (rows + unique) * payload)
                if fractions:
                    out[name]["unique_ancestor_fraction"] = sum(fractions) / len(fractions)
                    out[name]["moved_bytes_per_launch"] = sum(moved) / len(moved)
                    out[name]["moved_GBps"] = out[name]["moved_bytes_per_launch"] / seconds / 1e9
                if ess:
                    out[name]["ess_over_k"] = sum(ess) / len(ess)
        return out


class HipKernels:
    """The product backend.  One instance per process; per-device state is created lazily."""

    name = "hip"

    def __init__(self):
        self._lib = _lib.load()
        self._flags = {}
        self._lock = threading.Lock()
        self.timer = None  # set to a KernelTimer to time every launch (bench only)
        self._flag_constants = {}
        self.lds_max_particles = int(self._lib.aesmc_ancestor_index_lds_max_particles())
        self._affine_max_dim = None
        self._map_cache = {}        # id(weight) -> (weight, its aesmc_affine_map, (shape, strides))
        self._covers_last = None    # the operands of the last step `affine_logweight_covers` accepted
        self._wide_dim = None       # (aesmc_affine_wide_dim(), _min_dim(), _max_dim()): see `_wide_limits`
        self._pairs = None          # (key, tensor): the interleaved weight pairs of the maps the fused launch met last
        self.evaluation = 0         # bumped by `begin_evaluation`: what a cached weight-pair block belongs to
        self.WEIGHT_PAIRS = settings.knob("AESMC_K16_PAIRS", "1") != "0"      # measurement knob
        self.SCALED_PAIRS = settings.knob("AESMC_K16_SCALED", "1") != "0"     # measurement knob: the constants behind the pairs

    # ---- deferred status word ---------------------------------------------------------------
    def flags(self, device):
        key = torch.device(device).index
        if key is None:
            key = torch.cuda.current_device()
        word = self._flags.get(key)
        if word is None:
            with self._lock:
                word = self._flags.get(key)
                if word is None:
                    word = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", key))
                    self._flags[key] = word
        return word

    def read_flags(self, device):
        """Synchronising read-and-clear of the device status word."""
        word = self.flags(device)
        value = int(word.item())
        if value:
            word.zero_()
        return value

    def discard_flags(self):
        """Clears every device's status word without synchronising (an `infer` was abandoned half way:
        what its kernels flagged must not surface in the next call).  Skipped during a capture."""
        for index, word in list(self._flags.items()):
            with torch.cuda.device(index):
                if not torch.cuda.is_current_stream_capturing():
                    word.zero_()

    def _flag_unless_all(self, valid, bit):
        """ORs `bit` into the status word unless every element of the boolean tensor `valid` is True — device-side, no
        synchronisation, three small launches (all, select, or)."""
        device = valid.device
        constants = self._flag_constants.get((device, bit))
        if constants is None:
            constants = self._flag_constants[(device, bit)] = (
                torch.zeros((), dtype=torch.int32, device=device), torch.full((), bit, dtype=torch.int32, device=device))
        self.flags(device).bitwise_or_(torch.where(valid.all(), constants[0], constants[1]))

    def defer_support_check(self, valid):
        """FLAG_VALUE_OUTSIDE_SUPPORT unless all of `valid` (a value against a distribution's support) holds."""
        self._flag_unless_all(valid, _lib.FLAG_VALUE_OUTSIDE_SUPPORT)

    def defer_parameter_check(self, valid):
        """FLAG_INVALID_PARAMETER unless all of `valid` (a distribution parameter against its constraint) holds."""
        self._flag_unless_all(valid, _lib.FLAG_INVALID_PARAMETER)

    @staticmethod
    def _stream(t):
        # the raw handle of torch's current stream on t's device (torch.cuda.current_stream builds a Stream
        # object around the same call: 3 us of host time per launch)
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        if raw is not None:
            return raw(t.device.index if t.device.index is not None else torch.cuda.current_device())
        return torch.cuda.current_stream(t.device).cuda_stream

    # ---- K1 ------------------------------------------------------------------------------------
    def logweight_lse(self, a, b=None, c=None, want_lw=True, want_lse=True):
        """lw = a + b - c over [B,K]; lse[b] = logsumexp_k lw.  Returns (lw or None, lse or None)."""
        _require_hip(a, "log-prob term")
        tag = _tag(a, "log-prob term")
        if a.dim() != 2:
            raise ValueError("aesmc_amd: log-prob terms must be [batch_size, num_particles], got {}"
                             .format(tuple(a.shape)))
        terms = [a]
        for t in (b, c):
            if t is not None:
                _require_hip(t, "log-prob term")
                if t.shape != a.shape or t.dtype != a.dtype or t.device != a.device:
                    raise ValueError("aesmc_amd: log-prob terms disagree: {} {} {} vs {} {} {}".format(
                        tuple(t.shape), t.dtype, t.device, tuple(a.shape), a.dtype, a.device))
            terms.append(t)
        a, b, c = [None if t is None else t.contiguous() for t in terms]
        B, K = a.shape
        if b is None and c is None and not want_lse:
            return (a if want_lw else None), None
        need_lw = want_lw and not (b is None and c is None)
        lw = torch.empty_like(a) if need_lw else None
        lse = torch.empty(B, dtype=a.dtype, device=a.device) if want_lse else None
        with _on_device(a.device):
            args = (tag, _ptr(a), _ptr(b), _ptr(c), _ptr(lw), _ptr(lse), B, K, self._stream(a))
            _lib.check(self._lib.aesmc_logweight_lse(*args), "aesmc_logweight_lse")
            if self.timer is not None:
                esz = a.element_size()
                terms = 1 + (b is not None) + (c is not None) + (lw is not None)
                self.timer.note("logweight_lse", (self._lib.aesmc_logweight_lse, args),
                                B * K * esz * terms + B * esz, (a, b, c, lw, lse))
        if want_lw and not need_lw:
            lw = a
        return lw, lse

    def logweight_accumulate(self, a, b, c, acc, want_lw=True, want_lse=False):
        """K1 with the running sum over time of importance sampling: lw = a + b - c (b, c optional),
        total = acc + lw, lse[b] = logsumexp_k total.  Returns (lw or None, total, lse or None)."""
        _require_hip(a, "log-prob term")
        tag = _tag(a, "log-prob term")
        if a.dim() != 2:
            raise ValueError("aesmc_amd: log-prob terms must be [batch_size, num_particles], got {}"
                             .format(tuple(a.shape)))
        for t in (b, c, acc):
            if t is not None:
                _require_hip(t, "log-prob term")
                if t.shape != a.shape or t.dtype != a.dtype or t.device != a.device:
                    raise ValueError("aesmc_amd: log-prob terms disagree: {} {} {} vs {} {} {}".format(
                        tuple(t.shape), t.dtype, t.device, tuple(a.shape), a.dtype, a.device))
        if acc is None:
            raise ValueError("aesmc_amd: logweight_accumulate needs the running sum")
        a, b, c, acc = [None if t is None else t.contiguous() for t in (a, b, c, acc)]
        B, K = a.shape
        need_lw = want_lw and not (b is None and c is None)
        lw = torch.empty_like(a) if need_lw else None
        total = torch.empty_like(a)
        lse = torch.empty(B, dtype=a.dtype, device=a.device) if want_lse else None
        if a.numel() == 0:
            return (lw if need_lw else (a if want_lw else None)), total, lse
        with _on_device(a.device):
            args = (tag, _ptr(a), _ptr(b), _ptr(c), _ptr(acc), _ptr(lw), _ptr(total), _ptr(lse), B, K,
                    self._stream(a))
            _lib.check(self._lib.aesmc_logweight_accumulate(*args), "aesmc_logweight_accumulate")
            if self.timer is not None:
                esz = a.element_size()
                terms = 3 + (b is not None) + (c is not None) + (lw is not None)
                self.timer.note("logweight_accumulate", (self._lib.aesmc_logweight_accumulate, args),
                                B * K * esz * terms + (B * esz if want_lse else 0), (a, b, c, acc, lw, total, lse))
        if want_lw and not need_lw:
            lw = a
        return lw, total, lse

    def logweight_lse_backward(self, lw, lse, grad_lw, grad_lse, want_neg=True):
        _require_hip(lw, "lw")
        tag = _tag(lw, "lw")
        B, K = lw.shape
        lw = lw.contiguous()
        lse = lse.contiguous()
        if lse.shape != (B,) or lse.dtype != lw.dtype or lse.device != lw.device:
            raise ValueError("aesmc_amd: lse must be [{}] {} on {}".format(B, lw.dtype, lw.device))
        if grad_lw is not None:
            if grad_lw.shape != lw.shape or grad_lw.dtype != lw.dtype or grad_lw.device != lw.device:
                raise ValueError("aesmc_amd: grad_lw does not match lw")
            grad_lw = grad_lw.contiguous()
        if grad_lse is not None:
            if grad_lse.shape != (B,) or grad_lse.dtype != lw.dtype or grad_lse.device != lw.device:
                raise ValueError("aesmc_amd: grad_lse does not match lse")
            grad_lse = grad_lse.contiguous()
        g = torch.empty_like(lw)
        ng = torch.empty_like(lw) if want_neg else None
        with _on_device(lw.device):
            args = (tag, _ptr(lw), _ptr(lse), _ptr(grad_lw), _ptr(grad_lse), _ptr(g), _ptr(ng), B, K,
                    self._stream(lw))
            _lib.check(self._lib.aesmc_logweight_lse_backward(*args), "aesmc_logweight_lse_backward")
            if self.timer is not None:
                terms = 2 + (grad_lw is not None) + (ng is not None)
                self.timer.note("logweight_lse_backward",
                                (self._lib.aesmc_logweight_lse_backward, args),
                                B * K * lw.element_size() * terms, (lw, lse, grad_lw, grad_lse, g, ng))
        return g, ng

    # ---- K2 ------------------------------------------------------------------------------------
    def ancestor_index(self, log_w, u):
        """log_w [B,K] float32/64, u [B] float64 (both on the HIP device) -> int64 [B,K]."""
        _require_hip(log_w, "log_weight")
        _require_hip(u, "uniforms")
        tag = _tag(log_w, "log_weight")
        if log_w.dim() != 2:
            raise ValueError("aesmc_amd: log_weight must be [batch_size, num_particles], got {}"
                             .format(tuple(log_w.shape)))
        B, K = log_w.shape
        if u.dtype != torch.float64 or u.numel() != B or u.device != log_w.device:
            raise ValueError("aesmc_amd: uniforms must be {} float64 values on {}".format(B, log_w.device))
        log_w = log_w.contiguous()
        u = u.contiguous()
        idx = torch.empty((B, K), dtype=torch.int64, device=log_w.device)
        ws_bytes = int(self._lib.aesmc_workspace_bytes(B, K))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=log_w.device) if ws_bytes else None
        with _on_device(log_w.device):
            flags = self.flags(log_w.device)
            args = (tag, _ptr(log_w), _ptr(u), _ptr(idx), _ptr(flags), B, K, _ptr(ws), ws_bytes,
                    self._stream(log_w))
            _lib.check(self._lib.aesmc_ancestor_index(*args), "aesmc_ancestor_index")
            if self.timer is not None:
                self.timer.note("ancestor_index", (self._lib.aesmc_ancestor_index, args),
                                B * K * (log_w.element_size() + 8) + 8 * B, (log_w, u, idx, ws))
        idx._aesmc_sorted = True  # systematic resampling is monotone in k: lets K3's backward skip atomics
        return idx

    # A workgroup owns a whole batch row: copying the payload inside the step pays while one row's
    # payload is small next to the chip (measured: 160 KB rows at B >= 128 gain, 8 MB rows lose 2x).
    STEP_PAYLOAD_MAX_ROW_BYTES = 1 << 20

    def step_covers(self, log_w, payload=None):
        """Host-only: can `resample_step` take this log-weight tensor (and payload)?"""
        if not (log_w.is_cuda and log_w.dim() == 2 and log_w.dtype in _DTYPE_TAG):
            return False
        K = log_w.size(1)
        if K > self.lds_max_particles or log_w.numel() == 0:
            return False
        if payload is None:
            return True
        if not (torch.is_tensor(payload) and payload.is_cuda and payload.device == log_w.device and
                payload.size()[:2] == log_w.size() and payload.numel() > 0 and _inner_dense(payload)):
            return False
        esz = payload.element_size()
        row_bytes = esz * (payload[0, 0].numel())
        if row_bytes % 4 or (K * row_bytes) % 16 or K * row_bytes > self.STEP_PAYLOAD_MAX_ROW_BYTES:
            return False
        return all((x * esz) % 4 == 0 for x in (payload.stride(0), payload.stride(1))) and \
            payload.data_ptr() % 4 == 0

    def resample_step(self, log_w, u, payload=None, want_lse=False, want_child_end=False):
        """The fused step: (idx, lse, resampled payload) from one launch — idx as `ancestor_index`,
        lse = logsumexp over particles [B] when `want_lse`, payload[b, idx[b,k]] as `gather` when a
        payload tensor [B,K,...] is given (else None).  Returns None when the launch does not cover
        the operands (more particles than one workgroup holds, payload rows not 4-byte multiples):
        the caller then runs `ancestor_index` / `gather` / `logweight_lse` separately."""
        _require_hip(log_w, "log_weight")
        _require_hip(u, "uniforms")
        tag = _tag(log_w, "log_weight")
        if log_w.dim() != 2:
            raise ValueError("aesmc_amd: log_weight must be [batch_size, num_particles], got {}"
                             .format(tuple(log_w.shape)))
        B, K = log_w.shape
        if u.dtype != torch.float64 or u.numel() != B or u.device != log_w.device:
            raise ValueError("aesmc_amd: uniforms must be {} float64 values on {}".format(B, log_w.device))
        if K > self.lds_max_particles:
            return None
        log_w = log_w.contiguous()
        u = u.contiguous()
        row_bytes = sb = sk = 0
        dst = None
        if payload is not None:
            _require_hip(payload, "value")
            assert payload.size()[:2] == log_w.size()
            if payload.device != log_w.device:
                raise RuntimeError("aesmc_amd: value on {} but log_weight on {}".format(payload.device,
                                                                                   log_w.device))
            if not _inner_dense(payload):
                payload = payload.contiguous()
            esz = payload.element_size()
            row_bytes = esz
            for size in payload.shape[2:]:
                row_bytes *= size
            sb, sk = payload.stride(0) * esz, payload.stride(1) * esz
            dst = torch.empty(payload.shape, dtype=payload.dtype, device=payload.device)
            if dst.numel() == 0:
                payload = dst = None
        idx = torch.empty((B, K), dtype=torch.int64, device=log_w.device)
        lse = torch.empty((B,), dtype=log_w.dtype, device=log_w.device) if want_lse else None
        if idx.numel() == 0:
            return None
        child_end = None
        with _on_device(log_w.device):
            flags = self.flags(log_w.device)
            if want_child_end and dst is None:
                # the children ranges ride along (aesmc_resample_step_ranges): where each particle's children end, for
                # the propagation's backward, which sums a particle's children itself (`idx._aesmc_child_end`)
                child_end = torch.empty((B, K), dtype=torch.int32, device=log_w.device)
                entry = self._lib.aesmc_resample_step_ranges
                args = (tag, _ptr(log_w), _ptr(u), _ptr(idx), _ptr(lse), _ptr(child_end), _ptr(flags), B, K,
                        self._stream(log_w))
            else:
                entry = self._lib.aesmc_resample_step
                args = (tag, _ptr(log_w), _ptr(u), _ptr(idx), _ptr(lse), _ptr(payload if dst is not None else None),
                        _ptr(dst), _ptr(flags), B, K, row_bytes, sb, sk, self._stream(log_w))
            status = entry(*args)
            if status == 2:
                return None
            _lib.check(status, "aesmc_resample_step")
            if self.timer is not None:
                nbytes = B * K * (log_w.element_size() + 8) + 8 * B
                if dst is not None:  # K2's 12 B + K3's (8 + 2 row_bytes) per particle
                    nbytes += B * K * (8 + 2 * row_bytes)
                if child_end is not None:
                    nbytes += 4 * B * K
                self.timer.note("resample_step", (entry, args), nbytes, (log_w, u, idx, lse, payload, dst, child_end))
        idx._aesmc_sorted = True
        if child_end is not None:
            idx._aesmc_child_end = child_end
        return idx, lse, dst

    # ---- K3 ------------------------------------------------------------------------------------
    @staticmethod
    def _check_index(value, idx):
        _require_hip(idx, "ancestral_index")
        if idx.dtype != torch.int64:
            raise TypeError("aesmc_amd: ancestral_index must be int64 (torch.LongTensor), got {}"
                            .format(idx.dtype))
        if idx.device != value.device:
            raise RuntimeError("aesmc_amd: ancestral_index on {} but value on {}".format(
                idx.device, value.device))

    def gather(self, src, idx):
        """dst[b,k,...] = src[b, idx[b,k], ...]; src [B,K,...] any dtype, idx int64 [B,K]."""
        _require_hip(src, "value")
        self._check_index(src, idx)
        assert idx.size() == src.size()[:2]
        B, K = idx.shape
        if not _inner_dense(src):
            src = src.contiguous()  # trailing dims must be dense; the leading two may be strided
        row_elems = 1
        for s in src.shape[2:]:
            row_elems *= s
        esz = src.element_size()
        dst = torch.empty(src.shape, dtype=src.dtype, device=src.device)
        if dst.numel() == 0:
            return dst
        idx = idx.contiguous()
        with _on_device(src.device):
            flags = self.flags(src.device)
            args = (_ptr(src), _ptr(idx), _ptr(dst), _ptr(flags), B, K, row_elems * esz,
                    src.stride(0) * esz, src.stride(1) * esz, self._stream(src))
            _lib.check(self._lib.aesmc_resample_gather(*args), "aesmc_resample_gather")
            if self.timer is not None:
                self.timer.note("resample_gather", (self._lib.aesmc_resample_gather, args),
                                B * K * (8 + 2 * row_elems * esz), (src, idx, dst))
        return dst

    def gather_backward(self, grad_out, idx, sorted_index=False):
        """grad_src[b,j,...] = sum over {k: idx[b,k]==j} of grad_out[b,k,...].  `sorted_index`
        promises idx non-decreasing along k (outputs of `ancestor_index`): segmented sum, no atomics."""
        _require_hip(grad_out, "grad")
        self._check_index(grad_out, idx)
        tag = _tag(grad_out, "gradient of a resampled value")
        B, K = idx.shape
        grad_out = grad_out.contiguous()
        idx = idx.contiguous()
        row_elems = 1
        for s in grad_out.shape[2:]:
            row_elems *= s
        # The sorted-index kernel writes every row of the result exactly once — if the promise holds.  Indices K2 wrote
        # itself keep it by construction; a tag that was only INHERITED (a lineage composed from tagged indices, or one a
        # caller set) starts from zeros, so that a false promise leaves zero rows behind its flag, not stale memory.
        grad_src = torch.zeros_like(grad_out) if sorted_index == "inherited" else torch.empty_like(grad_out)
        if grad_src.numel() == 0:
            return grad_src
        with _on_device(grad_out.device):
            flags = self.flags(grad_out.device)
            args = (tag, _ptr(grad_out), _ptr(idx), _ptr(grad_src), _ptr(flags), B, K, row_elems,
                    1 if sorted_index else 0, self._stream(grad_out))
            _lib.check(self._lib.aesmc_resample_gather_backward(*args),
                       "aesmc_resample_gather_backward")
            if self.timer is not None:
                self.timer.note("resample_gather_backward",
                                (self._lib.aesmc_resample_gather_backward, args),
                                B * K * (8 + 2 * row_elems * grad_out.element_size()),
                                (grad_out, idx, grad_src))
        return grad_src


    def gather_backward_ranges(self, child_grad, child_end):
        """grad_src[b,k] = sum of child_grad[b, child_end[b,k-1] : child_end[b,k]] — torch.gather's backward stated with
        the children ranges instead of the indices (only for shapes the fused backward declines)."""
        B, K = child_end.shape
        ends = child_end.to(torch.int64)
        position = torch.arange(K, device=child_end.device).unsqueeze(0).expand(B, K).contiguous()
        # the ancestor of position k is the number of particles whose children end at or before k
        index = torch.searchsorted(ends, position, right=True).clamp_(max=K - 1)
        return self.gather_backward(child_grad, index, sorted_index=True)

    # ---- K4 ------------------------------------------------------------------------------------
    @staticmethod
    def _view3(t):
        """Describes a [B,K,*] tensor as (tensor, (sb, sk, sd), D) with the trailing dims collapsed
        into one of extent D and element stride sd (0 = broadcast).  Copies only when the trailing
        dims cannot be described by a single stride."""
        rank = t.dim()
        if rank == 3:                     # the usual case: nothing to collapse
            return t, t.stride(), t.size(2)
        if rank == 2:
            return t, t.stride() + (0,), 1
        D = 1
        for size in t.shape[2:]:
            D *= size
        dims = [(size, stride) for size, stride in zip(t.shape[2:], t.stride()[2:]) if size != 1]
        for (_, outer), (size, inner) in zip(dims, dims[1:]):
            if outer != inner * size:
                t = t.contiguous()
                return t, (t.stride(0), t.stride(1), 1), D
        sd = dims[-1][1] if dims else 0
        return t, (t.stride(0), t.stride(1), sd), D

    @staticmethod
    def _unique_bytes(t):
        n = t.element_size()
        for size, stride in zip(t.shape, t.stride()):
            if stride != 0:
                n *= size
        return n

    def _normal_operands(self, value, loc, scale):
        _require_hip(value, "value")
        tag = _tag(value, "value")
        if value.dim() < 2:
            raise ValueError("aesmc_amd: value must be [batch_size, num_particles, ...]")
        for name, t in (("loc", loc), ("scale", scale)):
            _require_hip(t, name)
            if t.shape != value.shape or t.dtype != value.dtype or t.device != value.device:
                raise ValueError("aesmc_amd: {} must be a {} view of shape {} on {}, got {} {} on {}"
                                 .format(name, value.dtype, tuple(value.shape), value.device, t.dtype,
                                         tuple(t.shape), t.device))
        (value, sv, D), (loc, sm, _), (scale, ss, _) = [self._view3(t) for t in (value, loc, scale)]
        return tag, value, loc, scale, sv, sm, ss, D

    def normal_logprob_sum(self, value, loc, scale):
        """sum over trailing dims of Normal(loc, scale).log_prob(value) -> [B,K]; loc and scale are
        views already expanded to value's shape (stride 0 where broadcast)."""
        tag, value, loc, scale, sv, sm, ss, D = self._normal_operands(value, loc, scale)
        B, K = value.shape[:2]
        out = torch.empty((B, K), dtype=value.dtype, device=value.device)
        if out.numel() == 0:
            return out
        with _on_device(value.device):
            args = (tag, _ptr(value), _ptr(loc), _ptr(scale), _ptr(out), B, K, D) + sv + sm + ss + \
                (self._stream(value),)
            _lib.check(self._lib.aesmc_normal_logprob_sum(*args), "aesmc_normal_logprob_sum")
            if self.timer is not None:
                nbytes = sum(self._unique_bytes(t) for t in (value, loc, scale)) + out.numel() * out.element_size()
                self.timer.note("normal_logprob_sum", (self._lib.aesmc_normal_logprob_sum, args),
                                nbytes, (value, loc, scale, out))
        return out

    def normal_logprob_sum_backward(self, value, loc, scale, grad_out, need_value, need_loc, need_scale):
        """Dense [B,K,*] gradients w.r.t. value / loc / scale (None where not needed)."""
        shape = value.shape
        tag, value, loc, scale, sv, sm, ss, D = self._normal_operands(value, loc, scale)
        B, K = shape[:2]
        if grad_out.shape != (B, K) or grad_out.dtype != value.dtype or grad_out.device != value.device:
            raise ValueError("aesmc_amd: grad_out must be [{}, {}] {}".format(B, K, value.dtype))
        grad_out = grad_out.contiguous()
        outs = [torch.empty(shape, dtype=value.dtype, device=value.device) if need else None
                for need in (need_value, need_loc, need_scale)]
        if value.numel() == 0 or not any(o is not None for o in outs):
            return tuple(None if o is None else o.zero_() for o in outs)
        with _on_device(value.device):
            args = (tag, _ptr(value), _ptr(loc), _ptr(scale), _ptr(grad_out), _ptr(outs[0]),
                    _ptr(outs[1]), _ptr(outs[2]), B, K, D) + sv + sm + ss + (self._stream(value),)
            _lib.check(self._lib.aesmc_normal_logprob_sum_backward(*args),
                       "aesmc_normal_logprob_sum_backward")
            if self.timer is not None:
                nbytes = sum(self._unique_bytes(t) for t in (value, loc, scale, grad_out)) + \
                    sum(o.numel() * o.element_size() for o in outs if o is not None)
                self.timer.note("normal_logprob_sum_backward",
                                (self._lib.aesmc_normal_logprob_sum_backward, args), nbytes,
                                (value, loc, scale, grad_out) + tuple(outs))
        return tuple(outs)


    # ---- K5 ------------------------------------------------------------------------------------
    @staticmethod
    def normal_logweight_covers(x, scale_p, y, scale_g, scale_q):
        """Host-only pre-test of K5's preconditions: HIP float tensors of equal leading shape, at
        least one value per particle.  The launch itself may still decline (wide rows that are not
        whole 16-byte vectors, wide rows with tensor scales): `normal_logweight` then returns None."""
        if not (x.is_cuda and y.is_cuda and x.dtype in _DTYPE_TAG and y.dtype == x.dtype):
            return False
        if x.dim() < 2 or y.dim() < 2 or x.shape[:2] != y.shape[:2] or x.device != y.device:
            return False
        return x.numel() != 0 and y.numel() != 0

    def normal_logweight(self, x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q):
        """lw = logN(x; loc_p, scale_p) + logN(y; loc_g, scale_g) - logN(x; loc_q, scale_q), each
        summed over trailing dims -> [B,K]; loc / scale are views expanded to x's (resp. y's) shape.
        Returns None when the kernel does not cover the operands (wide rows that are not whole
        aligned 16-byte vectors, or wide rows with tensor scales):
        the caller then takes the K4 + K1 route, which gives bit-identical numbers."""
        tag, x, loc_p, scale_p, sx, sp, ssp, Dx = self._normal_operands(x, loc_p, scale_p)
        _, y, loc_g, scale_g, sy, sg, ssg, Dy = self._normal_operands(y, loc_g, scale_g)
        _, x2, loc_q, scale_q, sx2, sq, ssq, _ = self._normal_operands(x, loc_q, scale_q)
        if y.dtype != x.dtype or y.device != x.device or y.shape[:2] != x.shape[:2]:
            return None
        if Dx < 1 or Dy < 1:
            return None
        if x2 is not x and x2.data_ptr() != x.data_ptr():    # _view3 had to copy x: keep one copy
            x2, sx2 = x, sx
        B, K = x.shape[:2]
        out = torch.empty((B, K), dtype=x.dtype, device=x.device)
        if out.numel() == 0:
            return out
        views = (_lib.View3 * 8)(*[_lib.View3(_ptr(t), *st) for t, st in (
            (x, sx), (loc_p, sp), (scale_p, ssp), (y, sy), (loc_g, sg), (scale_g, ssg), (loc_q, sq),
            (scale_q, ssq))])
        with _on_device(x.device):
            args = (tag, views, _ptr(out), B, K, Dx, Dy, self._stream(x))
            status = self._lib.aesmc_normal_logweight(*args)
            if status == 2:
                return None
            _lib.check(status, "aesmc_normal_logweight")
            if self.timer is not None:
                nbytes = sum(self._unique_bytes(t) for t in (x, loc_p, y, loc_g, loc_q)) + \
                    out.numel() * out.element_size()
                self.timer.note("normal_logweight", (self._lib.aesmc_normal_logweight, args), nbytes,
                                (x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q, out, views))
        return out

    def normal_logweight_backward(self, x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q, grad_lw, need,
                                  lw=None, lse=None, grad_lse=None):
        """Gradients of `normal_logweight` in one launch: dense [B,K,*] tensors in the order
        (x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q), None where `need[i]` is false.
        With `lw`, `lse`, `grad_lse` the incoming gradient is K1's backward formed in place,
        grad_lse[b] * exp(lw - lse[b]) (+ grad_lw if given): the per-step softmax term never goes
        through HBM as a [B,K] tensor."""
        tag, x, loc_p, scale_p, sx, sp, ssp, Dx = self._normal_operands(x, loc_p, scale_p)
        _, y, loc_g, scale_g, sy, sg, ssg, Dy = self._normal_operands(y, loc_g, scale_g)
        _, _, loc_q, scale_q, _, sq, ssq, _ = self._normal_operands(x, loc_q, scale_q)
        if Dx < 1 or Dy < 1:
            return None
        B, K = x.shape[:2]
        fused_lse = grad_lse is not None
        if fused_lse:
            lw, lse, grad_lse = lw.contiguous(), lse.contiguous(), grad_lse.contiguous()
            if lw.shape != (B, K) or lse.shape != (B,) or grad_lse.shape != (B,) or \
                    any(t.dtype != x.dtype or t.device != x.device for t in (lw, lse, grad_lse)):
                raise ValueError("aesmc_amd: lw must be [{0}, {1}], lse and grad_lse [{0}], all {2} on {3}".format(
                    B, K, x.dtype, x.device))
        if grad_lw is not None:
            grad_lw = grad_lw.contiguous()
        elif not fused_lse:
            raise ValueError("aesmc_amd: normal_logweight_backward needs grad_lw or (lw, lse, grad_lse)")
        make = lambda like, wanted: torch.empty(like.shape, dtype=like.dtype, device=like.device) if wanted else None
        outs = [make(x, need[0]), make(x, need[1]), make(x, need[2]), make(y, need[3]), make(y, need[4]),
                make(y, need[5]), make(x, need[6]), make(x, need[7])]
        if x.numel() == 0:
            return outs
        gx, gp, gsp, gy, gg, gsg, gq, gsq = outs
        views = (_lib.View3 * 8)(*[_lib.View3(_ptr(t), *st) for t, st in (
            (x, sx), (loc_p, sp), (scale_p, ssp), (y, sy), (loc_g, sg), (scale_g, ssg), (loc_q, sq),
            (scale_q, ssq))])
        with _on_device(x.device):
            tail = (_ptr(gx), _ptr(gp), _ptr(gy), _ptr(gg), _ptr(gq), _ptr(gsp), _ptr(gsg), _ptr(gsq), B, K, Dx, Dy,
                    self._stream(x))
            if fused_lse:
                entry, name = self._lib.aesmc_normal_logweight_lse_backward, "aesmc_normal_logweight_lse_backward"
                args = (tag, views, _ptr(lw), _ptr(lse), _ptr(grad_lse), _ptr(grad_lw)) + tail
            else:
                entry, name = self._lib.aesmc_normal_logweight_backward, "aesmc_normal_logweight_backward"
                args = (tag, views, _ptr(grad_lw)) + tail
            status = entry(*args)
            if status == 2:
                return None
            _lib.check(status, name)
            if self.timer is not None:
                live = [t for t in outs if t is not None]
                reads = [t for t in (x, loc_p, scale_p, y, loc_g, scale_g, loc_q, scale_q, grad_lw, lw) if t is not None]
                nbytes = sum(self._unique_bytes(t) for t in reads) + sum(t.numel() * t.element_size() for t in live)
                self.timer.note(name[6:], (entry, args), nbytes,
                                tuple(reads) + (lse, grad_lse, views) + tuple(live))
        return outs

    def normal_rsample(self, eps, loc, scale):
        """loc + eps * scale (product rounded first, as eager PyTorch) -> DENSE tensor of eps's
        shape [B,K,*]; loc and scale are views already expanded to that shape.  eps may be the
        transposed view of a [K,B,*] draw: the result is written in [B,K,*] order directly."""
        tag = _DTYPE_TAG[eps.dtype]
        (eps, se, D), (loc, sm, _), (scale, ss, _) = [self._view3(t) for t in (eps, loc, scale)]
        B, K = eps.shape[:2]
        out = torch.empty(eps.shape, dtype=eps.dtype, device=eps.device)
        if out.numel() == 0:
            return out
        with _on_device(eps.device):
            for attempt in (0, 1):
                views = [_lib.View3(_ptr(t), *st) for t, st in ((eps, se), (loc, sm), (scale, ss))]
                args = (tag, ctypes.byref(views[0]), ctypes.byref(views[1]), ctypes.byref(views[2]), _ptr(out),
                        B, K, D, self._stream(eps))
                status = self._lib.aesmc_normal_rsample(*args)
                if status != 2 or attempt == 1:
                    break
                eps = eps.contiguous()   # a noise layout the kernel does not know: materialise it
                se = (eps.stride(0), eps.stride(1), 1)
            _lib.check(status, "aesmc_normal_rsample")
            if self.timer is not None:
                nbytes = sum(self._unique_bytes(t) for t in (eps, loc, scale)) + out.numel() * out.element_size()
                self.timer.note("normal_rsample", (self._lib.aesmc_normal_rsample, args), nbytes,
                                (eps, loc, scale, out, views))
        return out

    # Below this many elements the noise launch + K6 pair is as fast (both are launch-latency bound) and the drawn
    # kernel's 4-byte accesses buy nothing.
    RSAMPLE_DRAWN_MIN_ELEMENTS = int(settings.knob("AESMC_RSAMPLE_DRAWN_MIN", str(1 << 18)))

    def normal_rsample_drawn(self, noise, loc, scale, shape):
        """K6 with the noise formed in the launch: loc + n * scale -> dense float32 [B,K,*] of `shape`, n being the
        tensor `torch.empty(shape).normal_()` would have held for the reservation `noise` (an `_philox.NoiseStream`);
        loc and scale are views already expanded to `shape`.  None where the launch does not apply (the caller draws
        the noise with `philox_normal` and takes `normal_rsample`)."""
        if loc.dtype != torch.float32 or len(shape) < 2:
            return None
        numel = 1
        for extent in shape:
            numel *= int(extent)
        if numel < self.RSAMPLE_DRAWN_MIN_ELEMENTS or numel >= (1 << 32) or numel != noise.numel:
            return None
        (loc, sm, D), (scale, ss, _) = [self._view3(t) for t in (loc, scale)]
        B, K = int(shape[0]), int(shape[1])
        out = torch.empty(tuple(shape), dtype=torch.float32, device=loc.device)
        with _on_device(loc.device):
            views = [_lib.View3(_ptr(t), *st) for t, st in ((loc, sm), (scale, ss))]
            args = (_DTYPE_TAG[torch.float32], ctypes.byref(views[0]), ctypes.byref(views[1]), _ptr(out), B, K, D,
                    noise.seed, noise.offset, noise.threads, 0, _ptr(noise.state), self._stream(loc))
            status = self._lib.aesmc_normal_rsample_drawn(*args)
            if status == 2:
                return None
            _lib.check(status, "aesmc_normal_rsample_drawn")
            if self.timer is not None:
                nbytes = self._unique_bytes(loc) + self._unique_bytes(scale) + 4 * numel
                self.timer.note("normal_rsample_drawn", (self._lib.aesmc_normal_rsample_drawn, args), nbytes,
                                (loc, scale, out, views))
        return out

    # ---- K8 / K9 / K10: linear-Gaussian particle propagation -----------------------------------
    @property
    def affine_max_dim(self):
        if self._affine_max_dim is None:
            self._affine_max_dim = int(self._lib.aesmc_affine_max_dim())
        return self._affine_max_dim

    def affine_covers(self, source, weight, offset=None):
        """Host-only test of what K8 / K9 / K10 assume about one affine location
        `offset + source @ weight.T`: source [B,K,din] float32/64 on the HIP device, weight
        [dout,din] with din, dout <= aesmc_affine_max_dim(), offset None, [dout] or [B,dout]."""
        if not (torch.is_tensor(source) and torch.is_tensor(weight)):
            return False
        if not (source.is_cuda and source.dtype in _DTYPE_TAG and source.dim() == 3 and source.numel() > 0):
            return False
        if weight.dim() != 2 or weight.dtype != source.dtype or weight.device != source.device:
            return False
        dout, din = weight.shape
        if din != source.size(2) or not (1 <= din <= self.affine_max_dim and 1 <= dout <= self.affine_max_dim):
            return False
        if offset is not None:
            if not torch.is_tensor(offset) or offset.dtype != source.dtype or offset.device != source.device:
                return False
            if tuple(offset.shape) not in ((dout,), (source.size(0), dout)):
                return False
        return True

    @staticmethod
    def _dense16(t):
        """`t` laid out contiguously from a 16-byte aligned address (what the tile loads assume)."""
        t = t.contiguous()
        if t.data_ptr() % 16:
            t = t.clone(memory_format=torch.contiguous_format)
        return t

    def _affine_map(self, weight, offset, slot=None):
        """(aesmc_affine_map, tensors it borrows).  weight may be any 2-D view (a transpose costs
        nothing); offset is [dout] (shared by every batch row) or [B, dout].  `slot` (the hot per-timestep
        launchers pass the map's position in their argument list): the struct made for this weight tensor in
        this position is kept — a model hands the same parameters in at every timestep — and only its offset
        fields rewritten; the C entry points have read it by the time they return."""
        off_ptr, off_sb = 0, 0
        if offset is not None:
            if offset.stride(-1) != 1:
                offset = offset.contiguous()
            off_ptr = offset.data_ptr()
            off_sb = offset.stride(0) if offset.dim() == 2 else 0
        # (the timer keeps the structs of the launches it sampled; a view such as `A.t()` — a tensor with an autograd
        #  history — is never kept: a cache that outlives the step would keep that step's autograd graph alive, which
        #  a later hipGraph capture of a backward pass cannot tolerate)
        cached = slot is not None and self.timer is None and weight.grad_fn is None
        if cached:
            entry = self._map_cache.get((id(weight), slot))
            if entry is not None and entry[0] is weight and entry[1].weight == weight.data_ptr() and \
                    entry[2] == (weight.shape, weight.stride()):
                amap = entry[1]
                amap.offset, amap.offset_stride_b = off_ptr, off_sb
                return amap, (weight, offset)
        dout, din = weight.shape
        amap = _lib.AffineMap(weight.data_ptr(), weight.stride(0), weight.stride(1), off_ptr, off_sb, dout, din)
        if cached:
            if len(self._map_cache) > 64:
                self._map_cache.clear()
            self._map_cache[(id(weight), slot)] = (weight, amap, (weight.shape, weight.stride()))
        return amap, (weight, offset)

    def particle_affine(self, x1, w1, offset=None, x2=None, w2=None, base=None, through_tanh=False):
        """K8: base + (offset + x1 @ w1.T + x2 @ w2.T) -> dense [B,K,dout]; x2 / w2, offset, base optional.
        Every element is one fma chain (w1's terms, then w2's) started from the offset.  `through_tanh`: the launch stores
        tanh of that (aesmc_particle_affine_tanh: the bits `torch.tanh` of the plain result holds)."""
        if not self.affine_covers(x1, w1, offset) or (x2 is not None and not self.affine_covers(x2, w2)):
            raise ValueError("aesmc_amd: particle_affine operands outside what kernel K8 covers")
        tag = _DTYPE_TAG[x1.dtype]
        B, K = x1.shape[:2]
        dout = w1.size(0)
        if x2 is not None and (x2.shape[:2] != x1.shape[:2] or w2.size(0) != dout or x2.dtype != x1.dtype):
            raise ValueError("aesmc_amd: particle_affine inputs disagree")
        x1 = self._dense16(x1)
        x2 = None if x2 is None else self._dense16(x2)
        if base is not None:
            if base.shape != (B, K, dout) or base.dtype != x1.dtype or base.device != x1.device:
                raise ValueError("aesmc_amd: particle_affine base must be [{}, {}, {}]".format(B, K, dout))
            base = self._dense16(base)
        out = torch.empty((B, K, dout), dtype=x1.dtype, device=x1.device)
        m1, keep1 = self._affine_map(w1, offset)
        m2, keep2 = self._affine_map(w2, None) if x2 is not None else (None, ())
        with _on_device(x1.device):
            args = (tag, _ptr(x1), ctypes.byref(m1), _ptr(x2), ctypes.byref(m2) if m2 is not None else None,
                    _ptr(base), _ptr(out), B, K, self._stream(x1))
            entry = self._lib.aesmc_particle_affine_tanh if through_tanh else self._lib.aesmc_particle_affine
            _lib.check(entry(*args), "aesmc_particle_affine")
            if self.timer is not None:
                esz = x1.element_size()
                nbytes = esz * B * K * (x1.size(2) + (x2.size(2) if x2 is not None else 0) +
                                        dout * (2 if base is not None else 1))
                self.timer.note("particle_affine", (entry, args), nbytes,
                                (x1, x2, base, out, m1, m2, keep1, keep2))
        return out

    def affine_rsample(self, source, weight, offset, eps, scale, out=None):
        """K9: (offset + source @ weight.T) + eps * scale -> dense [B,K,dout]; `scale` holds one value.
        `out`: a dense, 16-byte aligned [B,K,dout] tensor to write instead of a new one."""
        if not self.affine_covers(source, weight, offset):
            raise ValueError("aesmc_amd: affine_rsample operands outside what kernel K9 covers")
        tag = _DTYPE_TAG[source.dtype]
        B, K = source.shape[:2]
        dout = weight.size(0)
        if eps.shape != (B, K, dout) or eps.dtype != source.dtype or eps.device != source.device:
            raise ValueError("aesmc_amd: affine_rsample noise must be [{}, {}, {}] {}".format(B, K, dout, source.dtype))
        if scale.numel() != 1 or scale.dtype != source.dtype or scale.device != source.device:
            raise ValueError("aesmc_amd: affine_rsample takes one scale value on the device")
        source, eps = self._dense16(source), self._dense16(eps)
        if out is None:
            out = torch.empty((B, K, dout), dtype=source.dtype, device=source.device)
        else:
            self._check_out(out, (B, K, dout), source, "affine_rsample")
        amap, keep = self._affine_map(weight, offset)
        with _on_device(source.device):
            args = (tag, _ptr(source), ctypes.byref(amap), _ptr(eps), _ptr(scale), _ptr(out), B, K,
                    self._stream(source))
            _lib.check(self._lib.aesmc_affine_normal_rsample(*args), "aesmc_affine_normal_rsample")
            if self.timer is not None:
                nbytes = source.element_size() * B * K * (source.size(2) + 2 * dout)
                self.timer.note("affine_normal_rsample", (self._lib.aesmc_affine_normal_rsample, args), nbytes,
                                (source, eps, scale, out, amap, keep))
        return out

    @staticmethod
    def _check_out(out, shape, like, what):
        if not (torch.is_tensor(out) and tuple(out.shape) == tuple(shape) and out.dtype == like.dtype and
                out.device == like.device and out.is_contiguous() and out.data_ptr() % 16 == 0):
            raise ValueError("aesmc_amd: {}: `out` must be a dense, 16-byte aligned {} {} tensor on {}".format(
                what, tuple(shape), like.dtype, like.device))

    def affine_propagate(self, x_prev, eps, y_rows, transition, emission, proposal, scales, out_x, checked=False,
                         ancestors=None):
        """K15: the proposal's draw and the step's log-weight in one launch.  Writes
        x = loc_q(x_prev) + eps * s_q into `out_x` (K9's bits) and returns K10's log-weight [B,K] of
        (x_prev, x, y_rows) (K10's bits).  `ancestors` (int64 [B,K]): `x_prev` is the UN-resampled latent and
        its rows are fetched through the indices inside the launch — x_prev[b, ancestors[b,k]] is never
        written; None comes back when the launch does not cover the shape (fewer than ~43 particles per
        batch row): the caller gathers, then calls again without `ancestors`."""
        # `checked`: the caller has just run affine_logweight_covers on these operands (x_t in eps's place)
        if not checked and not self.affine_logweight_covers(x_prev, eps, y_rows, transition, emission, proposal, scales):
            raise ValueError("aesmc_amd: affine_propagate operands outside what kernel K15 covers")
        if eps.shape != x_prev.shape or eps.dtype != x_prev.dtype or eps.device != x_prev.device:
            raise ValueError("aesmc_amd: affine_propagate noise must match x_prev")
        tag = _DTYPE_TAG[eps.dtype]
        B, K, dx = eps.shape
        self._check_out(out_x, (B, K, dx), eps, "affine_propagate")
        x_prev, eps = self._dense16(x_prev), self._dense16(eps)
        if out_x.data_ptr() == x_prev.data_ptr():
            raise ValueError("aesmc_amd: affine_propagate cannot write the draw over x_prev")
        if ancestors is not None:
            self._check_index(x_prev, ancestors)
            if ancestors.shape != (B, K):
                raise ValueError("aesmc_amd: affine_propagate ancestors must be [{}, {}]".format(B, K))
            ancestors = ancestors.contiguous()
        if y_rows.stride(1) != 1:
            y_rows = y_rows.contiguous()
        out = torch.empty((B, K), dtype=eps.dtype, device=eps.device)
        maps = [self._affine_map(*term, slot=slot) for slot, term in enumerate((transition, emission, proposal))]
        with _on_device(eps.device):
            tail = (_ptr(eps), _ptr(y_rows), y_rows.stride(0), ctypes.byref(maps[0][0]),
                    ctypes.byref(maps[1][0]), ctypes.byref(maps[2][0]), _ptr(scales[0]), _ptr(scales[1]),
                    _ptr(scales[2]), _ptr(out_x), _ptr(out))
            if ancestors is not None:
                entry, name = self._lib.aesmc_affine_normal_propagate_resampled, "affine_normal_propagate_resampled"
                args = (tag, _ptr(x_prev), _ptr(ancestors)) + tail + (_ptr(self.flags(eps.device)), B, K,
                                                                      self._stream(eps))
                status = entry(*args)
                if status == 2:
                    return None
            else:
                entry, name = self._lib.aesmc_affine_normal_propagate, "affine_normal_propagate"
                args = (tag, _ptr(x_prev)) + tail + (B, K, self._stream(eps))
                status = entry(*args)
            _lib.check(status, "aesmc_" + name)
            if self.timer is not None:
                nbytes = eps.element_size() * (B * K * (3 * dx + 1) + y_rows.numel())
                if ancestors is not None:
                    nbytes += 8 * B * K
                self.timer.note(name, (entry, args), nbytes, (x_prev, ancestors, eps, y_rows, out, out_x, maps, scales))
        return out

    # Particles below which the noise is materialised (aesmc_philox_normal_fill, the same values) and the step takes the
    # launches that read it instead of forming it in the propagation launch.  0 since round 5: with one work item per
    # workgroup (linear_gaussian_item.hip) the fused launch is ahead at every size measured, down to B=8 K=128 (5.6 us
    # against 3.0 + 7.3 for the fill followed by K15; B=256 K=1024: 14.6 against 6.2 + 13.8; profiles/r05_k16_small_shapes.txt).
    # Rounds 3 / 4 needed 1M / 0.5M particles for the persistent form to win.  Shapes the launch's item geometry does not
    # cover (fewer than 128 particles per row) are declined by the library and take the fill route as before.
    DRAWN_MIN_PARTICLES = int(settings.knob("AESMC_K16_MIN_PARTICLES", "0"))

    def philox_normal(self, stream_desc, shape, device):
        """The float32 tensor `torch.empty(shape).normal_()` would have held for the generator state
        `stream_desc` (an `_philox.NoiseStream`) — for a draw that was left to a kernel which then did not run."""
        out = torch.empty(shape, dtype=torch.float32, device=device)
        if out.numel() != stream_desc.numel:
            raise ValueError("aesmc_amd: noise of {} elements asked from a reservation of {}".format(
                out.numel(), stream_desc.numel))
        if out.numel() == 0:
            return out
        with _on_device(out.device):
            args = (_ptr(out), out.numel(), stream_desc.seed, stream_desc.offset, stream_desc.threads, 0,
                    _ptr(stream_desc.state), self._stream(out))
            _lib.check(self._lib.aesmc_philox_normal_fill(*args), "aesmc_philox_normal_fill")
            if self.timer is not None:
                self.timer.note("philox_normal_fill", (self._lib.aesmc_philox_normal_fill, args), 4 * out.numel(), (out,))
        return out

    # ---- K13: two-layer tanh net over the particles ---------------------------------------------
    def particle_mlp_covers(self, x, weight1, offset1, weight2, bias2=None):
        """Host-only test of K13's preconditions: x [B,K,din] (din <= 16), weight1 [H,din] (H <= 64),
        offset1 [H] or [B,H], weight2 [dout,H] (dout <= 16), bias2 [dout] or None; at least ~43
        particles per batch row (the kernel stages the rows' offsets per 256-particle tile)."""
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype in _DTYPE_TAG and x.dim() == 3 and x.numel() > 0):
            return False
        tensors = [weight1, offset1, weight2] + ([bias2] if bias2 is not None else [])
        if not all(torch.is_tensor(t) and t.dtype == x.dtype and t.device == x.device for t in tensors):
            return False
        if weight1.dim() != 2 or weight2.dim() != 2:
            return False
        H, din = weight1.shape
        dout = weight2.size(0)
        if din != x.size(2) or weight2.size(1) != H or not (1 <= din <= 16 and 1 <= dout <= 16 and 1 <= H <= 64):
            return False
        if tuple(offset1.shape) not in ((H,), (x.size(0), H)):
            return False
        if bias2 is not None and tuple(bias2.shape) != (dout,):
            return False
        return (256 - 1) // x.size(1) + 2 <= 8

    def particle_mlp(self, x, weight1, offset1, weight2, bias2=None):
        """K13: bias2 + tanh(offset1 + x @ weight1.T) @ weight2.T -> dense [B,K,dout]; None when the kernel
        declines the shape (the caller keeps the PyTorch expression)."""
        if not self.particle_mlp_covers(x, weight1, offset1, weight2, bias2):
            return None
        tag = _DTYPE_TAG[x.dtype]
        B, K = x.shape[:2]
        x = self._dense16(x)
        out = torch.empty((B, K, weight2.size(0)), dtype=x.dtype, device=x.device)
        m1, keep1 = self._affine_map(weight1, offset1)
        m2, keep2 = self._affine_map(weight2, bias2)
        with _on_device(x.device):
            args = (tag, _ptr(x), ctypes.byref(m1), ctypes.byref(m2), _ptr(out), B, K, self._stream(x))
            status = self._lib.aesmc_particle_mlp(*args)
            if status == 2:
                return None
            _lib.check(status, "aesmc_particle_mlp")
            if self.timer is not None:
                nbytes = x.element_size() * B * K * (x.size(2) + out.size(2))
                self.timer.note("particle_mlp", (self._lib.aesmc_particle_mlp, args), nbytes,
                                (x, out, m1, m2, keep1, keep2))
        return out

    def particle_mlp_backward(self, grad_out, x, weight1, offset1, weight2, need_x=True):
        """K13b: for grad_out [B,K,dout] -> (grad_x [B,K,din] or None, grad_weight1 [H,din], grad_offset1 summed per batch
        row [B,H], grad_weight2 [dout,H]) with the hidden layer RECOMPUTED from x and offset1; None when the kernel
        declines the shape (K not a multiple of 256, din = 16: the caller differentiates the PyTorch expression)."""
        if not self.particle_mlp_covers(x, weight1, offset1, weight2) or grad_out.dtype != x.dtype or \
                grad_out.device != x.device or tuple(grad_out.shape) != (x.size(0), x.size(1), weight2.size(0)):
            return None
        B, K, din = x.shape
        H, dout = weight1.size(0), weight2.size(0)
        records = int(self._lib.aesmc_particle_mlp_backward_records(B, K))
        if records <= 0 or din > 15:
            return None
        tag = _DTYPE_TAG[x.dtype]
        x, grad_out = self._dense16(x), self._dense16(grad_out)
        chunks = (H + 15) // 16
        grad_x = torch.empty_like(x) if need_x else None
        rec_w1 = torch.empty((records, chunks, 16, 16), dtype=x.dtype, device=x.device)
        rec_w2 = torch.empty((records, chunks, 16, 16), dtype=x.dtype, device=x.device)
        rows = torch.empty((B, K // 64, 16 * chunks), dtype=x.dtype, device=x.device)
        m1, keep1 = self._affine_map(weight1, offset1)
        m2, keep2 = self._affine_map(weight2, None)
        with _on_device(x.device):
            args = (tag, _ptr(x), _ptr(grad_out), ctypes.byref(m1), ctypes.byref(m2), _ptr(grad_x), _ptr(rec_w1),
                    _ptr(rec_w2), _ptr(rows), B, K, self._stream(x))
            status = self._lib.aesmc_particle_mlp_backward(*args)
            if status == 2:
                return None
            _lib.check(status, "aesmc_particle_mlp_backward")
            if self.timer is not None:
                nbytes = x.element_size() * B * K * (2 * din + dout)
                self.timer.note("particle_mlp_backward", (self._lib.aesmc_particle_mlp_backward, args), nbytes,
                                (x, grad_out, grad_x, rec_w1, rec_w2, rows, m1, m2, keep1, keep2))
        # the wavefronts' partials, added in record order; grad_W1's chunk c holds rows 16 c .. 16 c + 15 (hidden units) x
        # inputs, grad_W2's chunk c holds outputs x hidden units 16 c .. 16 c + 15
        grad_w1 = rec_w1.sum(0).reshape(16 * chunks, 16)[:H, :din]
        grad_w2 = rec_w2.sum(0).permute(1, 0, 2).reshape(16, 16 * chunks)[:dout, :H]
        grad_rows = rows.sum(1)[:, :H]
        return grad_x, grad_w1, grad_rows, grad_w2

    def _wide_limits(self):
        """(the extent with the noise in the launch and the backward pieces, smallest latent width, largest width) of the
        matrix-core step, from the library."""
        limits = self._wide_dim
        if limits is None:
            limits = self._wide_dim = (int(self._lib.aesmc_affine_wide_dim()), int(self._lib.aesmc_affine_wide_min_dim()),
                                       int(self._lib.aesmc_affine_wide_max_dim()))
        return limits

    def affine_wide_covers(self, source, weight, offset=None, scale=None):
        """Host-only test of what K17 / K18 (rows of 128 values) and K17g / K18g (every other width) assume about one
        location `offset + source @ weight.T`: source [B,K,din] float32 on the HIP device with 17 <= din <= 256, weight
        [dout,din] as an nn.Linear holds it with dout <= 256, offset None, [dout] or [B,dout], scale (if given) one value.
        (Maps of at most 16 x 16 are the item kernels': `affine_covers`.)"""
        if not (torch.is_tensor(weight) and weight.dim() == 2 and weight.dtype == torch.float32 and weight.is_cuda):
            return False
        _, smallest, largest = self._wide_limits()
        shape = tuple(source.shape)
        dout, din = weight.shape
        if len(shape) != 3 or shape[2] != din or source.dtype != torch.float32 or source.device != weight.device or \
                shape[0] * shape[1] == 0:
            return False
        if not (smallest <= din <= largest) or not (1 <= dout <= largest):
            return False
        if not weight.is_contiguous():
            return False
        if offset is not None:
            if not torch.is_tensor(offset) or offset.dtype != torch.float32 or offset.device != weight.device or \
                    tuple(offset.shape) not in ((dout,), (shape[0], dout)) or offset.stride(-1) != 1:
                return False
        if scale is not None and not (torch.is_tensor(scale) and scale.numel() == 1 and scale.dtype == torch.float32 and
                                      scale.device == weight.device):
            return False
        return True

    def affine_propagate_wide(self, x_src, eps, y_rows, transition, emission, proposal, scales, out_x, ancestors=None):
        """K17 + K18 (rows of 128 values) / K17g + K18g (any latent width from 17 to 256, any observation width up to 256,
        any K): a linear-Gaussian step on the fp32 matrix cores — the draw `loc_q(x_prev) + eps * s_q` into `out_x` (the C
        oracle's bits) and the step's log-weights [B,K] (its values to rounding), x_prev = x_src[b, ancestors[b,k]] when
        `ancestors` is given.
        `eps`: the noise [B,K,dx], or an `_philox.NoiseStream` — the reservation of the `normal_` call that did not
        happen: the launch then forms the noise itself (None when the shape does not allow it — anything but rows of 128
        values with K a multiple of 4 * threads / 128: the caller materialises it with `philox_normal`).
        None when the launch does not cover the operands (other extents, strided weights)."""
        if x_src.dtype != torch.float32 or x_src.dim() != 3:
            return None
        B, K, dx = x_src.shape
        wide, smallest, largest = self._wide_limits()
        if not (smallest <= dx <= largest) or B * K == 0 or not x_src.is_cuda:
            return None
        # the observation as K18 reads it: float32 [B, dy] on the latents' device (a float64 observation — torch.from_numpy
        # data against a float32 model — would be read as float32 bytes: declined, PyTorch's promotion applies instead)
        if not (torch.is_tensor(y_rows) and y_rows.dim() == 2 and y_rows.size(0) == B and
                y_rows.dtype == torch.float32 and y_rows.device == x_src.device):
            return None
        dy = y_rows.size(1)
        if not (1 <= dy <= largest):
            return None
        drawn = not torch.is_tensor(eps)      # an `_philox.NoiseStream`: the launch forms the noise itself
        if drawn:
            if eps.numel != x_src.numel():
                raise ValueError("aesmc_amd: affine_propagate_wide: the reservation does not match x_src")
            if dx != wide or dy != wide or K % 32 or eps.threads % wide or K % (4 * (eps.threads // wide)):
                return None
        elif eps.shape != x_src.shape or eps.dtype != x_src.dtype or eps.device != x_src.device:
            raise ValueError("aesmc_amd: affine_propagate_wide noise must match x_src")
        for (weight, offset), scale, shape in zip((transition, emission, proposal), scales, ((dx, dx), (dy, dx), (dx, dx))):
            # (what affine_wide_covers tests, for callers that did not ask it)
            if not (torch.is_tensor(weight) and tuple(weight.shape) == shape and weight.is_contiguous() and
                    weight.dtype == torch.float32 and weight.device == x_src.device):
                return None
            if offset is not None and not (torch.is_tensor(offset) and offset.dtype == torch.float32 and
                                           offset.device == x_src.device and
                                           tuple(offset.shape) in ((shape[0],), (B, shape[0])) and offset.stride(-1) == 1):
                return None      # (rows / bases that are not 16-byte pieces: the launch moves them element by element)
            if not (torch.is_tensor(scale) and scale.numel() == 1 and scale.dtype == torch.float32 and
                    scale.device == x_src.device):
                return None
        self._check_out(out_x, (B, K, dx), x_src, "affine_propagate_wide")
        x_src = self._dense16(x_src)
        if not drawn:
            eps = self._dense16(eps)
        if out_x.data_ptr() == x_src.data_ptr() or (not drawn and out_x.data_ptr() == eps.data_ptr()):
            raise ValueError("aesmc_amd: affine_propagate_wide cannot write the draw over its inputs")
        if ancestors is not None:
            self._check_index(x_src, ancestors)
            if ancestors.shape != (B, K):
                raise ValueError("aesmc_amd: affine_propagate_wide ancestors must be [{}, {}]".format(B, K))
            ancestors = ancestors.contiguous()
        if y_rows.stride(1) != 1 or y_rows.stride(0) % 4 or y_rows.data_ptr() % 16:
            y_rows = y_rows.contiguous()
        out = torch.empty((B, K), dtype=torch.float32, device=x_src.device)
        ws_bytes = int(self._lib.aesmc_affine_wide_workspace_bytes_for(B, K, dx, dy))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x_src.device)
        maps = [self._affine_map(*term, slot=slot) for slot, term in enumerate((transition, emission, proposal))]
        with _on_device(x_src.device):
            noise = (eps.seed, eps.offset, eps.threads, _ptr(eps.state)) if drawn else (0, 0, 256, 0)
            args = (_ptr(x_src), _ptr(ancestors), 0 if drawn else _ptr(eps), _ptr(y_rows), y_rows.stride(0),
                    ctypes.byref(maps[0][0]), ctypes.byref(maps[1][0]), ctypes.byref(maps[2][0]), _ptr(scales[0]),
                    _ptr(scales[1]), _ptr(scales[2]), _ptr(out_x), _ptr(out), _ptr(ws), ws_bytes,
                    _ptr(self.flags(x_src.device)), B, K) + noise + (self._stream(x_src),)
            status = self._lib.aesmc_affine_normal_propagate_wide(*args)
            if status == 2:
                return None
            _lib.check(status, "aesmc_affine_normal_propagate_wide")
            if self.timer is not None:
                nbytes = 4 * (B * K * (3 * dx + 1) + y_rows.numel()) + (8 * B * K if ancestors is not None else 0)
                self.timer.note("affine_normal_propagate_wide", (self._lib.aesmc_affine_normal_propagate_wide, args),
                                nbytes, (x_src, ancestors, eps, y_rows, out, out_x, ws, maps, scales))
        return out

    def affine_initial_step(self, eps, loc_q, scale_q, loc_p, scale_p, y, weight, offset, scale_g, out_x):
        """K20: the first timestep of a run in one launch — `out_x[b,k,:] = loc_q + eps[k,b,:] * scale_q` (the reference's
        transposed `rsample((K,))`), the emission's location `offset + out_x @ weight.T` and the three Normal
        log-densities — with the bits of K6 + K8 + K5 (aesmc_affine_normal_initial_step).  `eps` [K,B,dx] contiguous;
        `loc_q`, `scale_q`, `loc_p`, `scale_p` views of [B,K,dx] and `y`, `scale_g` of [B,K,dy] whose particle stride is 0;
        `weight` [dy,dx] (any strides), `offset` None, [dy] or [B,dy].  Returns the log-weights [B,K], or None where the
        launch does not apply (nothing has been launched then)."""
        B, K, dx = out_x.shape
        if not (torch.is_tensor(eps) and eps.dtype == torch.float32 and eps.is_cuda and tuple(eps.shape) == (K, B, dx) and
                eps.is_contiguous() and out_x.dtype == torch.float32 and out_x.is_contiguous() and out_x.device == eps.device):
            return None
        if not (torch.is_tensor(weight) and weight.dim() == 2 and weight.size(1) == dx and weight.dtype == torch.float32 and
                weight.device == eps.device and 1 <= dx <= 16 and 1 <= weight.size(0) <= 16):
            return None
        dy = weight.size(0)
        if offset is not None and not (torch.is_tensor(offset) and offset.dtype == torch.float32 and
                                       offset.device == eps.device and tuple(offset.shape) in ((dy,), (B, dy))):
            return None
        views = []
        for view, extent in ((loc_q, dx), (scale_q, dx), (loc_p, dx), (scale_p, dx), (y, dy), (scale_g, dy)):
            if not (torch.is_tensor(view) and view.dtype == torch.float32 and view.device == eps.device and
                    tuple(view.shape) == (B, K, extent) and (K == 1 or view.stride(1) == 0)):
                return None
            views.append(_lib.View3(_ptr(view), view.stride(0), 0, view.stride(2)))
        out = torch.empty((B, K), dtype=torch.float32, device=eps.device)
        amap, held = self._affine_map(weight, offset)
        with _on_device(eps.device):
            args = (_ptr(eps),) + tuple(ctypes.byref(v) for v in views[:5]) + (ctypes.byref(amap), ctypes.byref(views[5]),
                                                                                _ptr(out_x), _ptr(out), B, K, self._stream(eps))
            status = self._lib.aesmc_affine_normal_initial_step(*args)
            if status == 2:
                return None
            _lib.check(status, "aesmc_affine_normal_initial_step")
            if self.timer is not None:
                nbytes = 4 * (2 * B * K * dx + B * K)
                self.timer.note("affine_normal_initial_step", (self._lib.aesmc_affine_normal_initial_step, args), nbytes,
                                (eps, loc_q, scale_q, loc_p, scale_p, y, scale_g, out_x, out, views, amap, held))
        return out

    def begin_evaluation(self):
        """Called once per `infer`: weight pairs built for an earlier evaluation are dropped (the weights may have been
        stepped in between; inside a hipGraph capture the rebuild must be part of the captured work; and the entry holds
        references to the weight tensors — with their autograd history, if they are computed per evaluation — which must
        not outlive the evaluation they belong to)."""
        self.evaluation += 1
        self._pairs = None

    def _build_pairs(self, maps, scales, device):
        """One launch: the interleaved pairs, with the three densities' constants behind them when `scales` is given
        (aesmc_affine_weight_pairs_scaled) — else their tag cleared (the propagating launch forms them itself)."""
        pairs = torch.empty(int(self._lib.aesmc_affine_weight_pairs_floats()), dtype=torch.float32, device=device)
        if scales is not None:
            _lib.check(self._lib.aesmc_affine_weight_pairs_scaled(
                ctypes.byref(maps[0][0]), ctypes.byref(maps[1][0]), ctypes.byref(maps[2][0]), _ptr(scales[0]),
                _ptr(scales[1]), _ptr(scales[2]), _ptr(pairs), self._stream(pairs)), "aesmc_affine_weight_pairs_scaled")
        else:
            _lib.check(self._lib.aesmc_affine_weight_pairs(ctypes.byref(maps[0][0]), ctypes.byref(maps[1][0]),
                                                           ctypes.byref(maps[2][0]), _ptr(pairs), self._stream(pairs)),
                       "aesmc_affine_weight_pairs")
        return pairs

    # how often one evaluation may rebuild the pairs because the SCALES it was handed are other tensors than last time
    # (a model that computes them per timestep): after that the pairs carry no constants and fit any scales
    SCALED_PAIRS_REBUILDS = 2

    def _weight_pairs(self, maps, weights, device, scales=None):
        """The three maps' weights as interleaved pairs (aesmc_affine_weight_pairs[_scaled]) for the fused propagation
        launch: built by one small launch the first time an evaluation meets these weights, then reused by its other
        timesteps.  The entry keeps the weight (and scale) tensors alive for as long as it stands (until the next
        `begin_evaluation`), so a cached (address, version) can never be met again on a DIFFERENT tensor that the
        allocator placed where a freed one was.

        `scales` (three one-value float32 tensors): the densities' constants ride behind the pairs while the evaluation
        keeps handing in the SAME scale tensors, unchanged (`is` and `_version`: Python-number scales are cached device
        constants, parameters live as long as the model); other tensors rebuild — twice per evaluation at most, then
        the entry carries no constants."""
        w0, w1, w2 = weights
        if scales is not None and not all(torch.is_tensor(s) and s.dtype == torch.float32 and s.numel() == 1 and
                                          not s.is_inference() for s in scales):
            scales = None
        if any(w.is_inference() for w in weights):      # (no version counter to tell an in-place update by: never cached)
            return self._build_pairs(maps, scales, device)
        held = self._pairs
        key = (self.evaluation, w0._version, w1._version, w2._version)
        layout = None
        same = held is not None and held[0] == key and held[2] is w0 and held[3] is w1 and held[4] is w2
        # (the very tensors of this evaluation's last step, unchanged since: a model's next timestep)
        if not same and held is not None and held[0] == key:
            layout = tuple((w.data_ptr(), w.shape, w.stride()) for w in weights)
            same = held[5] == layout      # (new view objects of the same, still referenced, storage: `x @ W.t()` makes one per call)
        rebuilds = 0
        if same:
            for_scales = held[6]
            if for_scales is None:      # no constants behind these pairs: they fit any scales
                return held[1]
            if scales is not None and all(a is b for a, b in zip(for_scales[0], scales)) and \
                    for_scales[1] == tuple(s._version for s in scales):
                return held[1]
            rebuilds = held[7] + 1
            if rebuilds > self.SCALED_PAIRS_REBUILDS:
                scales = None
        if layout is None:
            layout = tuple((w.data_ptr(), w.shape, w.stride()) for w in weights)
        pairs = self._build_pairs(maps, scales, device)
        for_scales = None if scales is None else (tuple(scales), tuple(s._version for s in scales))
        self._pairs = (key, pairs, w0, w1, w2, layout, for_scales, rebuilds)
        return pairs

    def affine_propagate_drawn(self, x_src, noise, y_rows, transition, emission, proposal, scales, out_x,
                               ancestors=None):
        """K16: K15 with the resampling gather (`ancestors`, or None) AND the noise inside the launch — `noise`
        is an `_philox.NoiseStream`, the reservation of the `normal_` call that did not happen.  Returns the
        log-weights [B,K], or None when the launch does not cover the shape (the caller materialises the
        noise with `philox_normal` and takes `affine_propagate`)."""
        if x_src.dtype != torch.float32:
            return None
        B, K, dx = x_src.shape
        if B * K < self.DRAWN_MIN_PARTICLES:
            return None
        if noise.numel != B * K * dx:
            raise ValueError("aesmc_amd: affine_propagate_drawn: the reservation does not match x_src")
        self._check_out(out_x, (B, K, dx), x_src, "affine_propagate_drawn")
        x_src = self._dense16(x_src)
        if out_x.data_ptr() == x_src.data_ptr():
            raise ValueError("aesmc_amd: affine_propagate_drawn cannot write the draw over x_src")
        if ancestors is not None:
            self._check_index(x_src, ancestors)
            if ancestors.shape != (B, K):
                raise ValueError("aesmc_amd: affine_propagate_drawn ancestors must be [{}, {}]".format(B, K))
            ancestors = ancestors.contiguous()
        if y_rows.stride(1) != 1:
            y_rows = y_rows.contiguous()
        out = torch.empty((B, K), dtype=x_src.dtype, device=x_src.device)
        maps = [self._affine_map(*term, slot=slot) for slot, term in enumerate((transition, emission, proposal))]
        with _on_device(x_src.device):
            pairs = None
            if self.WEIGHT_PAIRS and 2 <= dx <= 16:
                pairs = self._weight_pairs(maps, (transition[0], emission[0], proposal[0]), x_src.device,
                                           scales if self.SCALED_PAIRS else None)
            args = (_ptr(x_src), _ptr(ancestors), _ptr(y_rows), y_rows.stride(0), ctypes.byref(maps[0][0]),
                    ctypes.byref(maps[1][0]), ctypes.byref(maps[2][0]), _ptr(scales[0]), _ptr(scales[1]),
                    _ptr(scales[2]), _ptr(out_x), _ptr(out), _ptr(self.flags(x_src.device)), B, K, noise.seed,
                    noise.offset, noise.threads, _ptr(noise.state), _ptr(pairs), self._stream(x_src))
            status = self._lib.aesmc_affine_normal_propagate_drawn_paired(*args)
            if status == 2:
                return None
            _lib.check(status, "aesmc_affine_normal_propagate_drawn")
            if self.timer is not None:
                nbytes = 4 * (B * K * (2 * dx + 1) + y_rows.numel()) + (8 * B * K if ancestors is not None else 0)
                self.timer.note("affine_normal_propagate_drawn",
                                (self._lib.aesmc_affine_normal_propagate_drawn_paired, args), nbytes,
                                (x_src, ancestors, None, y_rows, out, out_x, maps, scales, pairs))
        return out

    def affine_logweight_covers(self, x_prev, x, y_rows, transition, emission, proposal, scales):
        """Host-only test of K10's preconditions; each of transition / emission / proposal is
        (weight, offset or None), y_rows the observation [B, dy], scales three one-value tensors."""
        if not (torch.is_tensor(x) and torch.is_tensor(x_prev) and torch.is_tensor(y_rows)):
            return False
        last = self._covers_last
        if last is not None and last[0] is transition[0] and last[1] is emission[0] and last[2] is proposal[0] and \
                last[3] is scales[0] and last[4] is scales[1] and last[5] is scales[2]:
            # the very parameter tensors of the step accepted last (a model's next timestep): what can differ are
            # the per-step operands — shapes, dtypes and devices of x_{t-1}, x_t, the observation and the offsets
            shape, dtype, device = last[6]
            weight = transition[0]
            if x.shape == shape and x_prev.shape == shape and x.dtype == dtype and x_prev.dtype == dtype and \
                    x.device == device and x_prev.device == device and weight.dtype == dtype and weight.device == device \
                    and y_rows.dtype == dtype and y_rows.device == device and y_rows.shape == last[7] and \
                    self._offsets_signature(transition[1], emission[1], proposal[1], dtype, device) == last[8]:
                return True
        if x_prev.shape != x.shape or x_prev.dtype != x.dtype or x_prev.device != x.device:
            return False
        if not (self.affine_covers(x_prev, *transition) and self.affine_covers(x, *emission) and
                self.affine_covers(x_prev, *proposal)):
            return False
        dx = x.size(2)
        if transition[0].size(0) != dx or proposal[0].size(0) != dx:
            return False
        if y_rows.dim() != 2 or y_rows.shape != (x.size(0), emission[0].size(0)) or \
                y_rows.dtype != x.dtype or y_rows.device != x.device:
            return False
        if not all(torch.is_tensor(s) and s.numel() == 1 and s.dtype == x.dtype and s.device == x.device
                   for s in scales):
            return False
        signature = self._offsets_signature(transition[1], emission[1], proposal[1], x.dtype, x.device)
        kept = (transition[0], emission[0], proposal[0]) + tuple(scales)
        if any(t.grad_fn is not None for t in kept):      # views with an autograd history are not kept (see _affine_map)
            self._covers_last = None
        elif signature is not None:
            self._covers_last = (transition[0], emission[0], proposal[0], scales[0], scales[1], scales[2],
                                 (x.shape, x.dtype, x.device), y_rows.shape, signature)
        return True

    @staticmethod
    def _offsets_signature(off_p, off_g, off_q, dtype, device):
        """Shapes of the three offsets (None where absent) when they are tensors of `dtype` on `device`, else None."""
        out = []
        for offset in (off_p, off_g, off_q):
            if offset is None:
                out.append(None)
            elif torch.is_tensor(offset) and offset.dtype == dtype and offset.device == device:
                out.append(offset.shape)
            else:
                return None
        return tuple(out)

    def affine_logweight(self, x_prev, x, y_rows, transition, emission, proposal, scales):
        """K10: the step's log-weight [B,K] with the three locations affine in the particles (see
        `affine_logweight_covers` for the operands)."""
        if not self.affine_logweight_covers(x_prev, x, y_rows, transition, emission, proposal, scales):
            raise ValueError("aesmc_amd: affine_logweight operands outside what kernel K10 covers")
        tag = _DTYPE_TAG[x.dtype]
        B, K, dx = x.shape
        x_prev, x = self._dense16(x_prev), self._dense16(x)
        if y_rows.stride(1) != 1:
            y_rows = y_rows.contiguous()
        out = torch.empty((B, K), dtype=x.dtype, device=x.device)
        maps = [self._affine_map(*term) for term in (transition, emission, proposal)]
        with _on_device(x.device):
            args = (tag, _ptr(x_prev), _ptr(x), _ptr(y_rows), y_rows.stride(0), ctypes.byref(maps[0][0]),
                    ctypes.byref(maps[1][0]), ctypes.byref(maps[2][0]), _ptr(scales[0]), _ptr(scales[1]),
                    _ptr(scales[2]), _ptr(out), B, K, self._stream(x))
            _lib.check(self._lib.aesmc_affine_normal_logweight(*args), "aesmc_affine_normal_logweight")
            if self.timer is not None:
                nbytes = x.element_size() * (B * K * (2 * dx + 1) + y_rows.numel())
                self.timer.note("affine_normal_logweight", (self._lib.aesmc_affine_normal_logweight, args), nbytes,
                                (x_prev, x, y_rows, out, maps, scales))
        return out

    def particle_affine_backward(self, grad, x, weight, need_x=True, need_weight=True, need_offset=False):
        """K11, the adjoint of `offset + x @ weight.T` for grad [B,K,dout]: (grad_x = grad @ weight,
        grad_weight [dout,din] = sum over particles of grad (outer) x — on the matrix cores, summed in a
        fixed order —, grad_offset [B,dout] = sum over each row's particles of grad); entries not asked
        for are None."""
        if not (need_x or need_weight or need_offset):
            return None, None, None
        if not self.affine_covers(x, weight) or grad.shape != x.shape[:2] + (weight.size(0),) or \
                grad.dtype != x.dtype or grad.device != x.device:
            raise ValueError("aesmc_amd: particle_affine_backward operands outside what kernel K11 covers")
        tag = _DTYPE_TAG[x.dtype]
        B, K, din = x.shape
        dout = weight.size(0)
        grad, x = self._dense16(grad), self._dense16(x)
        in_kernel_offset = need_offset and (256 - 1) // K + 2 <= 8      # the kernel's row table: >= ~43 particles per row
        gx = torch.empty_like(x) if need_x else None
        gw = torch.empty((dout, din), dtype=x.dtype, device=x.device) if need_weight else None
        goff = torch.empty((B, dout), dtype=x.dtype, device=x.device) if in_kernel_offset else None
        reduces = need_weight or in_kernel_offset
        ws_bytes = int(self._lib.aesmc_affine_backward_workspace_bytes(tag, B, K)) if reduces else 0
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device) if ws_bytes else None
        amap, keep = self._affine_map(weight, None)
        with _on_device(x.device):
            args = (tag, _ptr(grad), _ptr(x), ctypes.byref(amap), _ptr(gx), _ptr(gw), _ptr(goff), _ptr(ws), ws_bytes,
                    B, K, self._stream(x))
            status = self._lib.aesmc_particle_affine_backward(*args)
            if status == 2 and goff is not None:     # the row sums do not fit this shape: the caller's reduction instead
                goff = None
                args = args[:6] + (0,) + args[7:]
                status = self._lib.aesmc_particle_affine_backward(*args)
            _lib.check(status, "aesmc_particle_affine_backward")
            if self.timer is not None:
                nbytes = x.element_size() * B * K * (dout + (din if need_weight else 0) + (din if need_x else 0))
                self.timer.note("particle_affine_backward", (self._lib.aesmc_particle_affine_backward, args), nbytes,
                                (grad, x, gx, gw, goff, ws, amap, keep))
        if need_offset and goff is None:
            goff = grad.sum(dim=1)
        return gx, gw, goff

    def outer_sum(self, g, x):
        """sum over all particles of g[b,k,:] (outer) x[b,k,:] -> [dout, din]: the weight gradient of an
        affine location (K11 without the input gradient).  The weight's values are not read."""
        placeholder = torch.empty((g.size(2), x.size(2)), dtype=x.dtype, device=x.device)
        return self.particle_affine_backward(g, x, placeholder, need_x=False, need_weight=True)[1]

    def affine_logweight_backward(self, x_prev, x, y_rows, transition, emission, proposal, scales, need,
                                  grad_lw=None, lw=None, lse=None, grad_lse=None):
        """K12: gradients of `affine_logweight` with respect to its twelve operands, in the order
        (x_prev, x, y_rows, A, off_p, C, off_g, Q, off_q, s_p, s_g, s_q); None where `need[i]` is false
        or the operand is absent.  The incoming gradient is `grad_lw` [B,K] and / or K1's softmax term
        grad_lse[b] * exp(lw - lse[b]) formed in place (`lw`, `lse`, `grad_lse`).  One pass over x_prev
        and x; the weight gradients are summed on the matrix cores in a fixed order."""
        if not self.affine_logweight_covers(x_prev, x, y_rows, transition, emission, proposal, scales):
            raise ValueError("aesmc_amd: affine_logweight_backward operands outside what kernel K12 covers")
        (A, off_p), (C, off_g), (Q, off_q) = transition, emission, proposal
        tag = _DTYPE_TAG[x.dtype]
        B, K, dx = x.shape
        dy = y_rows.size(1)
        fused_lse = grad_lse is not None
        check = lambda t, shape, what: None if (t.shape == shape and t.dtype == x.dtype and t.device == x.device) \
            else (_ for _ in ()).throw(ValueError("aesmc_amd: {} must be {} {} on {}".format(what, shape, x.dtype, x.device)))
        if fused_lse:
            lw, lse, grad_lse = lw.contiguous(), lse.contiguous(), grad_lse.contiguous()
            check(lw, (B, K), "lw"), check(lse, (B,), "lse"), check(grad_lse, (B,), "grad_lse")
        if grad_lw is not None:
            grad_lw = grad_lw.contiguous()
            check(grad_lw, (B, K), "grad_lw")
        elif not fused_lse:
            raise ValueError("aesmc_amd: affine_logweight_backward needs grad_lw or (lw, lse, grad_lse)")
        x_prev, x = self._dense16(x_prev), self._dense16(x)
        if y_rows.stride(1) != 1:
            y_rows = y_rows.contiguous()
        make = lambda shape, wanted: torch.empty(shape, dtype=x.dtype, device=x.device) if wanted else None
        gx_prev, gx = make((B, K, dx), need[0]), make((B, K, dx), need[1])
        rows_p = make((B, dx), need[4] and off_p is not None)      # location gradients summed over each row's particles
        rows_g = make((B, dy), (need[6] and off_g is not None) or need[2])
        rows_q = make((B, dx), need[8] and off_q is not None)
        gA, gC, gQ = make((dx, dx), need[3]), make((dy, dx), need[5]), make((dx, dx), need[7])
        gscales = make((3,), need[9] or need[10] or need[11])
        ws_bytes = int(self._lib.aesmc_affine_backward_workspace_bytes(tag, B, K))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        outs = _lib.AffineLogweightGrads(_ptr(gx_prev), _ptr(gx), 0, 0, 0, _ptr(gA), _ptr(gC), _ptr(gQ), _ptr(gscales),
                                         _ptr(rows_p), _ptr(rows_g), _ptr(rows_q))
        maps = [self._affine_map(*term) for term in (transition, emission, proposal)]
        with _on_device(x.device):
            args = (tag, _ptr(x_prev), _ptr(x), _ptr(y_rows), y_rows.stride(0), ctypes.byref(maps[0][0]),
                    ctypes.byref(maps[1][0]), ctypes.byref(maps[2][0]), _ptr(scales[0]), _ptr(scales[1]),
                    _ptr(scales[2]), _ptr(lw) if fused_lse else 0, _ptr(lse) if fused_lse else 0,
                    _ptr(grad_lse) if fused_lse else 0, _ptr(grad_lw), ctypes.byref(outs), _ptr(ws), ws_bytes, B, K,
                    self._stream(x))
            status = self._lib.aesmc_affine_normal_logweight_backward(*args)
            if status == 2:     # too few particles per batch row for the fused kernel's row table
                return self.affine_logweight_backward_unfused(
                    x_prev, x, y_rows, transition, emission, proposal, scales, need, grad_lw=grad_lw,
                    lw=lw if fused_lse else None, lse=lse if fused_lse else None,
                    grad_lse=grad_lse if fused_lse else None)
            _lib.check(status, "aesmc_affine_normal_logweight_backward")
            if self.timer is not None:
                dense = [t for t in (gx_prev, gx) if t is not None]
                nbytes = x.element_size() * (B * K * (2 * dx + 1)) + sum(t.numel() * t.element_size() for t in dense)
                self.timer.note("affine_normal_logweight_backward",
                                (self._lib.aesmc_affine_normal_logweight_backward, args), nbytes,
                                (x_prev, x, y_rows, lw, lse, grad_lse, grad_lw, outs, ws, maps, scales, gA, gC, gQ,
                                 gscales, rows_p, rows_g, rows_q) + tuple(dense))
        fold = lambda rows, off: rows if off.dim() == 2 else rows.sum(dim=0)     # a shared [d] offset: all rows
        grads = [gx_prev, gx, None, gA, None, gC, None, gQ, None, None, None, None]
        if need[2]:
            grads[2] = -rows_g
        if rows_p is not None:
            grads[4] = fold(rows_p, off_p)
        if need[6] and off_g is not None:
            grads[6] = fold(rows_g, off_g)
        if rows_q is not None:
            grads[8] = fold(rows_q, off_q)
        for slot, s in ((9, scales[0]), (10, scales[1]), (11, scales[2])):
            if need[slot]:
                grads[slot] = gscales[slot - 9].reshape(s.shape)
        return grads

    def affine_logweight_backward_unfused(self, x_prev, x, y_rows, transition, emission, proposal, scales, need,
                                          grad_lw=None, lw=None, lse=None, grad_lse=None):
        """The same gradients by the route K12 replaces — K8 x 3 to materialise the locations, K5's
        backward, K8 on the transposed weights, K11's outer sums — kept as the cross-check of K12."""
        (A, off_p), (C, off_g), (Q, off_q) = transition, emission, proposal
        s_p, s_g, s_q = scales
        B, K, dx = x.shape
        dy = y_rows.size(1)
        loc_p = self.particle_affine(x_prev, A, off_p)
        loc_g = self.particle_affine(x, C, off_g)
        loc_q = self.particle_affine(x_prev, Q, off_q)
        y_expanded = y_rows.unsqueeze(1).expand(B, K, dy)
        want = [True, True, bool(need[9]), bool(need[2]), True, bool(need[10]), True, bool(need[11])]
        outs = self.normal_logweight_backward(x, loc_p, s_p.expand_as(loc_p), y_expanded, loc_g,
                                              s_g.expand_as(loc_g), loc_q, s_q.expand_as(loc_q), grad_lw, want,
                                              lw=lw, lse=lse, grad_lse=grad_lse)
        if outs is None:
            raise RuntimeError("aesmc_amd: K5's backward declined operands of an affine step")
        gx, g_loc_p, g_sp, g_y, g_loc_g, g_sg, g_loc_q, g_sq = outs
        fold = lambda g, off: None if off is None else (g.sum(dim=1) if off.dim() == 2 else g.sum(dim=(0, 1)))
        grads = [None] * 12
        if need[0]:
            grads[0] = self.particle_affine(g_loc_p, A.t(), None, g_loc_q, Q.t())
        if need[1]:
            grads[1] = self.particle_affine(g_loc_g, C.t(), base=gx)
        if need[2]:
            grads[2] = g_y.sum(dim=1)
        if need[3]:
            grads[3] = self.outer_sum(g_loc_p, x_prev)
        if need[4] and off_p is not None:
            grads[4] = fold(g_loc_p, off_p)
        if need[5]:
            grads[5] = self.outer_sum(g_loc_g, x)
        if need[6] and off_g is not None:
            grads[6] = fold(g_loc_g, off_g)
        if need[7]:
            grads[7] = self.outer_sum(g_loc_q, x_prev)
        if need[8] and off_q is not None:
            grads[8] = fold(g_loc_q, off_q)
        for slot, g, s in ((9, g_sp, s_p), (10, g_sg, s_g), (11, g_sq, s_q)):
            if need[slot]:
                grads[slot] = g.sum().reshape(s.shape)
        return grads

    # ---- K14: the whole backward of a step whose x_t is the proposal's reparameterised draw ------
    def affine_step_backward(self, x_prev, x, y_rows, transition, emission, proposal, scales, need, lw, lse,
                             grad_lse=None, grad_x=None, grad_lw=None, ancestors=None, child_grad=None, child_end=None,
                             chain=None):
        """K14: gradients of one SMC step (log-weights `lw` of K10, their row log-sum-exp `lse`) whose x IS the
        draw  loc_q(x_prev) + s_q eps  of K9 from the same proposal operands, with respect to
        (x_prev, x, y_rows, A, off_p, C, off_g, Q, off_q, s_p, s_g, s_q) — x's own slot is always None: the
        gradient `grad_x` that arrives at x from later steps and the step's own gradient at x travel on
        through the draw inside the kernel (K12 + K11 + the accumulations between them, one pass).
        `chain` (a dict, only through ancestors): {"carry": (workspace, records[, pairs address, pairs' holder]) of the step run before this one with the
        SAME A, C, Q and scales, or None; "defer": leave this step's sums for those parameters as records}.  A
        deferring call returns None in their slots and sets chain["left"] = (workspace, records) for the next call
        to carry; a carrying call's gradients for them include everything carried.  Where the kernel declines the
        shape, what was carried is collected and added here and nothing is left (chain["left"] = None)."""
        if chain is not None:
            chain["left"] = None
            if ancestors is None:
                raise ValueError("aesmc_amd: affine_step_backward chains the weights' gradients only through ancestors")
        if x.dim() == 3 and x.size(2) > self.affine_max_dim:
            if chain is not None or child_grad is not None:
                raise ValueError("aesmc_amd: a wide step's backward takes the summed gradient (no children ranges, no chain)")
            return self.affine_step_backward_wide(x_prev, x, y_rows, transition, emission, proposal, scales, need, lw, lse,
                                                  grad_lse=grad_lse, grad_x=grad_x, grad_lw=grad_lw, ancestors=ancestors)
        if not self.affine_logweight_covers(x_prev, x, y_rows, transition, emission, proposal, scales):
            raise ValueError("aesmc_amd: affine_step_backward operands outside what kernel K14 covers")
        if need[1]:
            raise ValueError("aesmc_amd: affine_step_backward: x is the proposal's draw and has no gradient slot")
        (A, off_p), (C, off_g), (Q, off_q) = transition, emission, proposal
        tag = _DTYPE_TAG[x.dtype]
        B, K, dx = x.shape
        dy = y_rows.size(1)
        check = lambda t, shape, what: None if (t.shape == shape and t.dtype == x.dtype and t.device == x.device) \
            else (_ for _ in ()).throw(ValueError("aesmc_amd: {} must be {} {} on {}".format(what, shape, x.dtype, x.device)))
        fused_lse = grad_lse is not None
        if fused_lse:
            lw, lse, grad_lse = lw.contiguous(), lse.contiguous(), grad_lse.contiguous()
            check(lw, (B, K), "lw"), check(lse, (B,), "lse"), check(grad_lse, (B,), "grad_lse")
        if grad_lw is not None:
            grad_lw = grad_lw.contiguous()
            check(grad_lw, (B, K), "grad_lw")
        if grad_x is not None:
            grad_x = self._dense16(grad_x)
            check(grad_x, (B, K, dx), "grad_x")
        if not fused_lse and grad_lw is None and grad_x is None and child_grad is None:
            raise ValueError("aesmc_amd: affine_step_backward needs a gradient: (lw, lse, grad_lse), grad_lw or grad_x")
        if child_grad is not None:
            # the NEXT step's gradient of the rows it resampled from x (one row per child) and its children ranges: the
            # kernel sums each particle's children itself — torch.gather's backward, where it is consumed
            if ancestors is None:
                raise ValueError("aesmc_amd: affine_step_backward folds the children's gradient only through ancestors")
            check(child_grad, (B, K, dx), "child_grad")
            child_grad = self._dense16(child_grad)
            if child_end is None or child_end.shape != (B, K) or child_end.dtype != torch.int32 or \
                    child_end.device != x.device:
                raise ValueError("aesmc_amd: affine_step_backward child_end must be int32 [{}, {}] on {}".format(B, K, x.device))
            child_end = child_end.contiguous()
        if ancestors is not None:
            # `x_prev` is the un-resampled latent: the kernel fetches x_prev[b, ancestors[b,k]] itself and slot 0 of
            # the result is the gradient of those RESAMPLED rows (the caller sums children into ancestors)
            self._check_index(x_prev, ancestors)
            if ancestors.shape != (B, K):
                raise ValueError("aesmc_amd: affine_step_backward ancestors must be [{}, {}]".format(B, K))
            ancestors = ancestors.contiguous()
        x_prev, x = self._dense16(x_prev), self._dense16(x)
        if y_rows.stride(1) != 1:
            y_rows = y_rows.contiguous()
        make = lambda shape, wanted: torch.empty(shape, dtype=x.dtype, device=x.device) if wanted else None
        gx_prev = make((B, K, dx), need[0])
        rows_p = make((B, dx), need[4] and off_p is not None)
        rows_g = make((B, dy), (need[6] and off_g is not None) or need[2])
        rows_q = make((B, dx), need[8] and off_q is not None)
        carry = chain["carry"] if chain is not None else None
        defer = chain is not None and bool(chain["defer"])
        gA, gC, gQ = (make((dx, dx), need[3] and not defer), make((dy, dx), need[5] and not defer),
                      make((dx, dx), need[7] and not defer))
        gscales = make((3,), (need[9] or need[10] or need[11]) and not defer)
        ws_bytes = int(self._lib.aesmc_affine_backward_workspace_bytes(tag, B, K))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
        outs = _lib.AffineLogweightGrads(_ptr(gx_prev), 0, 0, 0, 0, _ptr(gA), _ptr(gC), _ptr(gQ), _ptr(gscales),
                                         _ptr(rows_p), _ptr(rows_g), _ptr(rows_q))
        maps = [self._affine_map(*term) for term in (transition, emission, proposal)]
        with _on_device(x.device):
            middle = (_ptr(x), _ptr(y_rows), y_rows.stride(0), ctypes.byref(maps[0][0]),
                      ctypes.byref(maps[1][0]), ctypes.byref(maps[2][0]), _ptr(scales[0]), _ptr(scales[1]),
                      _ptr(scales[2]), _ptr(lw) if fused_lse else 0, _ptr(lse) if fused_lse else 0,
                      _ptr(grad_lse) if fused_lse else 0, _ptr(grad_lw), _ptr(grad_x))
            tail = (ctypes.byref(outs), _ptr(ws), ws_bytes)
            link = None
            if chain is not None:
                # (carry[2], carry[3]: where the run's interleaved weight pairs lie and the workspace that holds them — built by
                #  the run's first call, handed on from step to step instead of one small launch per step)
                pairs_in = carry[2] if carry is not None and len(carry) > 2 and carry[2] else 0
                link = _lib.AffineChain(_ptr(carry[0]) if carry is not None else 0, carry[1] if carry is not None else 0,
                                        (2 if (need[9] or need[10] or need[11]) else 1) if defer else 0, 0, pairs_in, 0)
            if ancestors is not None:
                entry = self._lib.aesmc_affine_step_backward_resampled
                args = (tag, _ptr(x_prev), _ptr(ancestors)) + middle + (_ptr(child_grad), _ptr(child_end)) + tail + \
                    (_ptr(self.flags(x.device)), ctypes.byref(link) if link is not None else None, B, K, self._stream(x))
            else:
                entry = self._lib.aesmc_affine_step_backward
                args = (tag, _ptr(x_prev)) + middle + tail + (B, K, self._stream(x))
            status = entry(*args)
            if status == 2 and child_grad is not None:      # the unfused route takes the summed gradient as a tensor
                summed = self.gather_backward_ranges(child_grad, child_end)
                grad_x, child_grad, child_end = (summed if grad_x is None else grad_x + summed), None, None
            if status == 2 and ancestors is not None:       # the unfused route wants the resampled rows as a tensor
                x_prev, ancestors = self.gather(x_prev, ancestors), None
            if status == 2:     # too few particles per batch row for the fused kernel's row table
                grads = self.affine_step_backward_unfused(
                    x_prev, x, y_rows, transition, emission, proposal, scales, need, lw if fused_lse else None,
                    lse if fused_lse else None, grad_lse=grad_lse if fused_lse else None, grad_x=grad_x,
                    grad_lw=grad_lw)
                if carry is not None:      # what the steps before left: finished here, added to this step's own
                    carried = self.affine_backward_collect(carry, x.dtype, x.device, dx, dy, need, scales)
                    for slot, value in enumerate(carried):
                        if value is not None:
                            grads[slot] = value if grads[slot] is None else grads[slot] + value
                return grads
            _lib.check(status, "aesmc_affine_step_backward")
            if defer:
                handed_on = carry is not None and len(carry) > 2 and carry[2] and link.pairs_out == carry[2]
                chain["left"] = (ws, int(link.records), link.pairs_out or 0, carry[3] if handed_on else ws)
            if self.timer is not None:
                nbytes = x.element_size() * B * K * (2 * dx + 1 + (dx if grad_x is not None else 0) +
                                                     (dx if gx_prev is not None else 0))
                if ancestors is not None:
                    nbytes += 8 * B * K
                if child_grad is not None:
                    nbytes += x.element_size() * B * K * dx + 4 * B * K
                self.timer.note("affine_step_backward" + ("_resampled" if ancestors is not None else ""), (entry, args),
                                nbytes, (x_prev, ancestors, x, y_rows, lw, lse, grad_lse, grad_lw, grad_x, child_grad,
                                         child_end, outs, ws, maps, scales, gA, gC, gQ, gscales, rows_p, rows_g, rows_q,
                                         gx_prev, link, carry))
        fold = lambda rows, off: rows if off.dim() == 2 else rows.sum(dim=0)
        grads = [gx_prev, None, None, gA, None, gC, None, gQ, None, None, None, None]
        if need[2]:
            grads[2] = -rows_g
        if rows_p is not None:
            grads[4] = fold(rows_p, off_p)
        if need[6] and off_g is not None:
            grads[6] = fold(rows_g, off_g)
        if rows_q is not None:
            grads[8] = fold(rows_q, off_q)
        for slot, s in ((9, scales[0]), (10, scales[1]), (11, scales[2])):
            if need[slot] and gscales is not None:
                grads[slot] = gscales[slot - 9].reshape(s.shape)
        return grads

    @property
    def wide_dim(self):
        return self._wide_limits()[0]

    @property
    def wide_adjoint_tile(self):
        tile = getattr(self, "_wide_adjoint_tile", None)
        if tile is None:
            tile = self._wide_adjoint_tile = int(self._lib.aesmc_wide_adjoint_tile())
        return tile

    def _wide_adjoint_covers(self, u, weight, scale):
        """Do aesmc_wide_adjoint_scale / _merge take these operands (rows of 128 float32 values, dense, whole tiles)?"""
        return (u.dtype == torch.float32 and u.dim() == 3 and u.size(2) == self.wide_dim and u.is_contiguous() and
                u.size(1) % self.wide_adjoint_tile == 0 and u.numel() > 0 and weight.dtype == torch.float32 and
                weight.shape == u.shape[:2] and weight.is_contiguous() and scale.dtype == torch.float32 and
                scale.numel() == 1 and u.data_ptr() % 16 == 0)

    def wide_adjoint_scale(self, u, weight, scale, want_sq, want_rows, base=None):
        """`u` [B,K,128] (a residual d — or, `base` [B,128] given, the location, d = base[b] - u — overwritten by the
        location's adjoint weight d / scale^2) -> (sum_j d_j^2 [B,K] or None, the adjoint summed over each batch row's
        particles [B,128] or None): aesmc_wide_adjoint_scale."""
        B, K, d = u.shape
        sq = torch.empty((B, K), dtype=torch.float32, device=u.device) if want_sq else None
        tiles = K // self.wide_adjoint_tile
        rows = torch.empty((B, tiles, d), dtype=torch.float32, device=u.device) if want_rows else None
        with _on_device(u.device):
            args = (_ptr(u), _ptr(weight), _ptr(scale), _ptr(base), base.stride(0) if base is not None else 0, _ptr(sq),
                    _ptr(rows), B, K, self._stream(u))
            _lib.check(self._lib.aesmc_wide_adjoint_scale(*args), "aesmc_wide_adjoint_scale")
            if self.timer is not None:
                self.timer.note("wide_adjoint_scale", (self._lib.aesmc_wide_adjoint_scale, args), 8 * u.numel(),
                                (u, weight, scale, sq, rows, base))
        return sq, (rows.sum(1) if rows is not None else None)

    def wide_adjoint_merge(self, u_p, at_x, weight, scale, want_sq, want_rows_p, want_rows_x, value=None, base=None,
                           add=None):
        """The transition's residual d in `u_p` (or, `value` [B,K,128] given, its location: d = (value - base[b]) - u_p, base
        [B,128] or None) and the gradient arriving at x_t in `at_x` (plus `add`, if given), both [B,K,128] and overwritten:
        u_p <- weight d / scale^2, at_x <- (add + at_x) - u_p; returns (sum_j d_j^2 [B,K], u_p's and at_x's sums over each batch row's particles
        [B,128]), None where not wanted: aesmc_wide_adjoint_merge."""
        B, K, d = u_p.shape
        tiles = K // self.wide_adjoint_tile
        sq = torch.empty((B, K), dtype=torch.float32, device=u_p.device) if want_sq else None
        rows_p = torch.empty((B, tiles, d), dtype=torch.float32, device=u_p.device) if want_rows_p else None
        rows_x = torch.empty((B, tiles, d), dtype=torch.float32, device=u_p.device) if want_rows_x else None
        with _on_device(u_p.device):
            args = (_ptr(u_p), _ptr(at_x), _ptr(weight), _ptr(scale), _ptr(value), _ptr(base),
                    base.stride(0) if base is not None else 0, _ptr(add), _ptr(sq), _ptr(rows_p), _ptr(rows_x), B, K,
                    self._stream(u_p))
            _lib.check(self._lib.aesmc_wide_adjoint_merge(*args), "aesmc_wide_adjoint_merge")
            if self.timer is not None:
                self.timer.note("wide_adjoint_merge", (self._lib.aesmc_wide_adjoint_merge, args), 16 * u_p.numel(),
                                (u_p, at_x, weight, scale, sq, rows_p, rows_x, value, base, add))
        return sq, (rows_p.sum(1) if rows_p is not None else None), (rows_x.sum(1) if rows_x is not None else None)

    def affine_step_backward_wide(self, x_prev, x, y_rows, transition, emission, proposal, scales, need, lw, lse,
                                  grad_lse=None, grad_x=None, grad_lw=None, ancestors=None):
        """`affine_step_backward` for rows wider than the fused kernels take (BASELINE.json configs[4]: 128 values): the
        same twelve gradients, RECOMPUTED from what the forward launches (K17 / K18) left — x_{t-1}, the ancestors, x_t,
        the log-weights — instead of retained: the step keeps no location, no noise and no resampled latent for its
        backward (three [B,K,128] tensors per timestep in the GEMM route: what made training at this extent outgrow HBM).
        The adjoint's contractions — three locations, three transposed maps, three sums of outer products over the
        particles — are [B K, 128] x [128, 128] products on the matrix cores through the GEMM library; the element-wise
        parts between them are PyTorch's.  Reference: autograd of aesmc/state.py:114-155, :179 and
        aesmc/inference.py:108-130 for one timestep."""
        (A, off_p), (C, off_g), (Q, off_q) = transition, emission, proposal
        s_p, s_g, s_q = scales
        B, K, dx = x.shape
        dy = y_rows.size(1)
        with torch.no_grad():
            moved = x_prev if ancestors is None else self.gather(x_prev, ancestors)      # x_{t-1}[b, ancestors[b, k]]
            weight = None
            if grad_lse is not None:
                weight = grad_lse.unsqueeze(1) * torch.exp(lw - lse.unsqueeze(1))           # d L / d lw through the row lse
            if grad_lw is not None:
                weight = grad_lw if weight is None else weight + grad_lw
            # residual = base - source @ W.T with the map's offset folded into `base` ([B,1,d'] broadcast over the
            # particles, or the full [B,K,d'] tensor): ONE pass — the GEMM's epilogue — instead of a product, an offset add
            # and a subtraction
            def residual(base, source, W, offset):
                if offset is not None:
                    base = base - (offset.unsqueeze(1) if offset.dim() == 2 else offset)
                return torch.baddbmm(base, source, W.t().unsqueeze(0).expand(B, -1, -1), beta=1, alpha=-1)
            # sum over the particles of u ⊗ v: [d', N] x [N, d] with N = B K in the millions is a shape the GEMM library
            # picks poor tiles for (0.8 - 0.9 ms at N = 2^20); as a batch of N / 4096 products of depth 4096, summed: ~0.4 ms
            def outer_sum(u, v):
                n = B * K
                depth = 4096 if n % 4096 == 0 and n > 4096 else n
                parts = torch.bmm(u.reshape(n // depth, depth, u.size(2)).transpose(1, 2), v.reshape(n // depth, depth, v.size(2)))
                return parts.sum(0) if parts.size(0) > 1 else parts[0]
            rows = lambda u, off: u.sum(1) if off.dim() == 2 else u.sum((0, 1))
            incoming = grad_x      # what arrives at x_t from later steps
            grads = [None] * 12
            at_prev = None
            rows_x = None
            if weight is not None:
                g = weight.unsqueeze(2)
                # emission: u_g = g (y - loc_g) / s_g^2 — its gradient reaches C, its offset, y, s_g and x_t (C^T u_g)
                weight = weight.contiguous()
                want_rows_g = (need[6] and off_g is not None) or need[2]
                fused = dx == dy and x.is_contiguous() and moved.is_contiguous() and \
                    self._wide_adjoint_covers(x, weight, s_g) and self._wide_adjoint_covers(x, weight, s_p)
                rows_g = None
                if fused:
                    # the location by a plain product, then ONE pass (K19): d = (y - offset)[b] - location formed there (no
                    # [B,K,128] copy of the broadcast row for an epilogue to start from), the adjoint in place, sum_j d_j^2
                    # per particle, the rows' sums
                    base_g = y_rows if off_g is None else y_rows - off_g
                    base_g = base_g.contiguous()
                    u_g = torch.matmul(x, C.t())
                    sq_g, rows_g = self.wide_adjoint_scale(u_g, weight, s_g, need[10], want_rows_g, base=base_g)
                    if need[10]:
                        grads[10] = (weight * (sq_g / s_g ** 3 - dy / s_g)).sum().reshape(s_g.shape)
                else:
                    u_g = residual(y_rows.unsqueeze(1), x, C, off_g)
                    if need[10]:
                        grads[10] = (weight * (u_g.square().sum(2) / s_g ** 3 - dy / s_g)).sum().reshape(s_g.shape)
                    u_g.mul_(g / (s_g * s_g))
                if need[5]:
                    grads[5] = outer_sum(u_g, x)
                if want_rows_g:
                    if rows_g is None:
                        rows_g = u_g.sum(1)
                    if need[6] and off_g is not None:
                        grads[6] = rows_g if off_g.dim() == 2 else rows_g.sum(0)
                    if need[2]:
                        grads[2] = -rows_g
                # what arrives at x_t: later steps' gradient + C^T u_g - u_p — in K19's one pass where it runs, else
                # accumulated in the products' epilogues
                later = None
                if fused and incoming is not None and incoming.is_contiguous() and incoming.dtype == torch.float32 and \
                        incoming.data_ptr() % 16 == 0:
                    later, at_x = incoming, torch.matmul(u_g, C)
                else:
                    at_x = torch.matmul(u_g, C) if incoming is None else \
                        torch.baddbmm(incoming, u_g, C.unsqueeze(0).expand(B, -1, -1))
                del u_g
                # transition: u_p = g (x_t - loc_p) / s_p^2 — reaches A, its offset, s_p, x_{t-1} (A^T u_p) and x_t (- u_p)
                rows_x = None
                if fused and at_x.is_contiguous():
                    # the location by a plain product, then one pass over it, x_t and at_x (K19): d = (x_t - offset) -
                    # location, u_p and at_x in place, sum_j d_j^2, both tensors' row sums
                    u_p = torch.matmul(moved, A.t())
                    sq_p, rows_p, rows_x = self.wide_adjoint_merge(
                        u_p, at_x, weight, s_p, need[9], need[4] and off_p is not None, need[8] and off_q is not None,
                        value=x, base=None if off_p is None else (off_p if off_p.dim() == 2 else off_p.unsqueeze(0).expand(B, -1)).contiguous(),
                        add=later)
                    if need[9]:
                        grads[9] = (weight * (sq_p / s_p ** 3 - dx / s_p)).sum().reshape(s_p.shape)
                    if need[4] and off_p is not None:
                        grads[4] = rows_p if off_p.dim() == 2 else rows_p.sum(0)
                else:
                    if later is not None:
                        at_x.add_(later)
                    u_p = residual(x, moved, A, off_p)
                    if need[9]:
                        grads[9] = (weight * (u_p.square().sum(2) / s_p ** 3 - dx / s_p)).sum().reshape(s_p.shape)
                    u_p.mul_(g / (s_p * s_p))
                    at_x.sub_(u_p)
                    if need[4] and off_p is not None:
                        grads[4] = rows(u_p, off_p)
                if need[3]:
                    grads[3] = outer_sum(u_p, moved)
                if need[0]:
                    at_prev = torch.matmul(u_p, A)
                del u_p
                incoming = at_x
            # the draw x_t = loc_q(x_{t-1}) + s_q eps carries everything that arrived at x_t to Q, its offset, s_q, x_{t-1}
            if incoming is not None:
                if need[7]:
                    grads[7] = outer_sum(incoming, moved)
                if need[8] and off_q is not None:
                    if weight is not None and rows_x is not None:
                        grads[8] = rows_x if off_q.dim() == 2 else rows_x.sum(0)
                    else:
                        grads[8] = rows(incoming, off_q)
                if need[11]:
                    noise_times_scale = residual(x, moved, Q, off_q)       # s_q eps
                    value = (incoming * noise_times_scale).sum() / s_q
                    if weight is not None:
                        value = value + weight.sum() * (dx / s_q)      # - d log q / d s_q = + d / s_q per particle
                    grads[11] = value.reshape(s_q.shape)
                if need[0]:
                    # (at_prev is this function's own product: accumulated in place — an out-of-place baddbmm copies it first)
                    at_prev = torch.matmul(incoming, Q) if at_prev is None else \
                        at_prev.baddbmm_(incoming, Q.unsqueeze(0).expand(B, -1, -1))
            elif need[11] and weight is not None:
                grads[11] = (weight.sum() * (dx / s_q)).reshape(s_q.shape)
            if need[0]:
                grads[0] = at_prev if at_prev is not None else torch.zeros_like(x)
        return grads

    def affine_backward_collect(self, left, dtype, device, dx, dy, need, scales):
        """The weights' and scales' gradients (a 12-slot list like affine_step_backward's, None elsewhere) out of the
        records a deferring K14 call left — `left` = its chain["left"] — when no later call carried them on."""
        ws, records = left[0], left[1]      # (+ where the run's weight pairs lie and their holder: not needed here)
        make = lambda shape, wanted: torch.empty(shape, dtype=dtype, device=device) if wanted else None
        gA, gC, gQ = make((dx, dx), need[3]), make((dy, dx), need[5]), make((dx, dx), need[7])
        gscales = make((3,), need[9] or need[10] or need[11])
        outs = _lib.AffineLogweightGrads(0, 0, 0, 0, 0, _ptr(gA), _ptr(gC), _ptr(gQ), _ptr(gscales), 0, 0, 0)
        with _on_device(device):
            args = (_DTYPE_TAG[dtype], _ptr(ws), records, dx, dy, ctypes.byref(outs), self._stream(ws))
            _lib.check(self._lib.aesmc_affine_backward_collect(*args), "aesmc_affine_backward_collect")
            if self.timer is not None:
                self.timer.note("affine_backward_collect", (self._lib.aesmc_affine_backward_collect, args),
                                records * 1024 * (8 if dtype == torch.float64 else 4), (ws, outs, gA, gC, gQ, gscales))
        grads = [None, None, None, gA, None, gC, None, gQ, None, None, None, None]
        for slot, s in ((9, scales[0]), (10, scales[1]), (11, scales[2])):
            if need[slot]:
                grads[slot] = gscales[slot - 9].reshape(s.shape)
        return grads


    def affine_step_backward_unfused(self, x_prev, x, y_rows, transition, emission, proposal, scales, need, lw, lse,
                                     grad_lse=None, grad_x=None, grad_lw=None):
        """The same gradients by the launches K14 replaces — K12 (with x's gradient), the accumulation, K11 for
        the draw — the route for shapes K14 declines and its cross-check."""
        (Q, off_q) = proposal
        inner = list(need)
        inner[1] = True
        if grad_lse is None and grad_lw is None:      # only later steps' gradient arrives
            grad_lw = torch.zeros(x.shape[:2], dtype=x.dtype, device=x.device)
        grads = self.affine_logweight_backward(x_prev, x, y_rows, transition, emission, proposal, scales, inner,
                                               grad_lw=grad_lw, lw=lw, lse=lse, grad_lse=grad_lse)
        total = grads[1] if grad_x is None else grads[1] + grad_x
        grads[1] = None
        want_off = bool(need[8]) and off_q is not None
        gsrc, gw, rows = self.particle_affine_backward(total.contiguous(), x_prev, Q, bool(need[0]), bool(need[7]),
                                                       want_off)
        add = lambda a, b: b if a is None else a + b
        if need[0]:
            grads[0] = add(grads[0], gsrc)
        if need[7]:
            grads[7] = add(grads[7], gw)
        if want_off:
            grads[8] = add(grads[8], rows if off_q.dim() == 2 else rows.sum(dim=0))
        if need[11]:
            eps = (x - self.particle_affine(x_prev, Q, off_q)) / scales[2]
            grads[11] = add(grads[11], (total * eps).sum().reshape(scales[2].shape))
        return grads

    # ---- K7 ------------------------------------------------------------------------------------
    def particle_summary(self, log_w, value=None, want_log_ess=False, want_mean=False, want_second=False):
        """(log_ess [B], mean [B,...], second moment [B,...]) under w = softmax(log_w, dim=1); entries
        not asked for are None.  value [B,K,...] of log_w's dtype, may be strided."""
        _require_hip(log_w, "log_weight")
        tag = _tag(log_w, "log_weight")
        if log_w.dim() != 2:
            raise ValueError("aesmc_amd: log_weight must be [batch_size, num_particles], got {}"
                             .format(tuple(log_w.shape)))
        B, K = log_w.shape
        if K == 0:
            raise ValueError("aesmc_amd: particle summaries need at least one particle")
        log_w = log_w.contiguous()
        view, D, tail = None, 0, ()
        if want_mean or want_second:
            _require_hip(value, "value")
            if value.dtype != log_w.dtype or value.device != log_w.device:
                raise ValueError("aesmc_amd: value must be {} on {}".format(log_w.dtype, log_w.device))
            assert value.size()[:2] == log_w.size()
            tail = tuple(value.shape[2:])
            value, sv, D = self._view3(value)
            view = _lib.View3(_ptr(value), *sv)
        make = lambda shape: torch.empty(shape, dtype=log_w.dtype, device=log_w.device)
        log_ess = make((B,)) if want_log_ess else None
        mean = make((B,) + tail) if want_mean else None
        second = make((B,) + tail) if want_second else None
        if B == 0:
            return log_ess, mean, second
        if view is None or D == 0:
            view, D = None, 0
            if mean is not None:
                mean.zero_()        # rows without values: empty sums
            if second is not None:
                second.zero_()
        ws_bytes = int(self._lib.aesmc_particle_summary_workspace_bytes(tag, B, K, D))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=log_w.device) if ws_bytes else None
        with _on_device(log_w.device):
            args = (tag, _ptr(log_w), ctypes.byref(view) if view is not None else None,
                    _ptr(log_ess), _ptr(mean) if D > 0 else 0, _ptr(second) if D > 0 else 0, B, K, D,
                    _ptr(ws), ws_bytes, self._stream(log_w))
            _lib.check(self._lib.aesmc_particle_summary(*args), "aesmc_particle_summary")
            if self.timer is not None:
                nbytes = log_w.numel() * log_w.element_size() + (self._unique_bytes(value) if view is not None else 0)
                self.timer.note("particle_summary", (self._lib.aesmc_particle_summary, args), nbytes,
                                (log_w, value, view, log_ess, mean, second, ws))
        return log_ess, mean, second


_provider = None
_provider_lock = threading.Lock()


def get():
    """The active kernel provider (created on first use; raises if the library is missing)."""
    global _provider
    if _provider is None:
        with _provider_lock:
            if _provider is None:
                _provider = HipKernels()
    return _provider


def _swap_provider_for_tests(provider):
    """Test hook (used by tests/conftest.py only): substitute the kernel provider so the host
    logic can be exercised on CPU tensors against the oracle.  Returns the previous provider."""
    global _provider
    previous, _provider = _provider, provider
    return previous
