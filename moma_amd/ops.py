"""torch-facing wrappers of the C ABI (include/moma_hip.h): device pointers and the current HIP stream
are handed to libmoma_hip.so; autograd Functions wire the hand-written backward kernels in.

PyTorch is plumbing here (memory, streams, autograd graph); all arithmetic of the hot path happens in
the HIP library.  Every wrapper refuses CPU tensors: there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Sequence, Tuple

import torch

from . import _lib
from ._lib import DT_BF16, DT_F32, PREC_BF16, PREC_F32, EMA_BLOCK_ELEMS, MomaHipError, check

_PREC = {"fp32": PREC_F32, "f32": PREC_F32, "float32": PREC_F32, PREC_F32: PREC_F32,
         "bf16": PREC_BF16, "bfloat16": PREC_BF16, PREC_BF16: PREC_BF16}


def prec_code(p) -> int:
    try:
        return _PREC[p]
    except KeyError:
        raise ValueError(f"unknown precision {p!r}; use 'fp32' or 'bf16'")


def _qdtype(queue: torch.Tensor) -> int:
    if queue.dtype == torch.float32:
        return DT_F32
    if queue.dtype == torch.bfloat16:
        return DT_BF16
    raise TypeError(f"queue dtype must be float32 or bfloat16, got {queue.dtype}")


def _dev(t: torch.Tensor, name: str, dtype=torch.float32, contiguous=True) -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise MomaHipError(f"{name}: the MoMA hot path runs only on the GPU (HIP library); got a "
                           f"{'CPU tensor' if isinstance(t, torch.Tensor) else type(t)}. No CPU fallback exists.")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


def _dense_pair(p: torch.Tensor, e: torch.Tensor) -> None:
    """Element-wise kernels walk the raw storage: both tensors must be dense with identical strides
    (row-major or channels_last -- the memory format only permutes the walk order)."""
    for t, nm in ((p, "param"), (e, "param_ema")):
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            raise MomaHipError(f"{nm}: the MoMA hot path runs only on the GPU (HIP library). No CPU fallback exists.")
        if t.dtype != torch.float32:
            raise TypeError(f"{nm}: expected float32, got {t.dtype}")
    dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
    if not dense or p.stride() != e.stride():
        raise ValueError("momentum_update: parameter pair must be dense with identical strides "
                         f"(got {p.stride()} vs {e.stride()})")


# Optional instrumentation (bench.py): a callable (name) -> context manager recording HIP events on the
# current stream around a C-ABI call.  None in normal operation.
_EVENT_RECORDER = None


def set_event_recorder(rec) -> None:
    global _EVENT_RECORDER
    _EVENT_RECORDER = rec


# Optional (bench.py): a callable returning three raw hipEvent_t handles that the library records on the dispatches of the next
# moma_infonce_fused call: begin / end of its dominant kernel and end of its last kernel (see moma_infonce_fused_q).
_KERNEL_EVENTS = None


def set_kernel_event_provider(fn) -> None:
    global _KERNEL_EVENTS
    _KERNEL_EVENTS = fn


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# Optional tracing (MOMA_ROCTX=1): a roctx range (torch.cuda.nvtx = roctx on ROCm) around every C-ABI call of K1 - K4 and around the
# phases of the graph-served step (helper/step_graph.py), for `rocprofv3 --marker-trace --kernel-trace -- python ...`.  The
# reference has no tracing of its own; off by default: a range is two more host calls per library call.
_ROCTX = __import__("os").environ.get("MOMA_ROCTX", "0") == "1"


class _Range:
    def __init__(self, name, inner=None):
        self.name, self.inner = name, inner

    def __enter__(self):
        torch.cuda.nvtx.range_push(self.name)
        if self.inner is not None:
            self.inner.__enter__()
        return self

    def __exit__(self, *a):
        if self.inner is not None:
            self.inner.__exit__(*a)
        torch.cuda.nvtx.range_pop()
        return False


def trace_range(name):
    """context manager: a roctx range when MOMA_ROCTX=1, nothing otherwise"""
    return _Range(name) if _ROCTX else _NullCtx()


def _timed(name):
    # (no events inside a stream capture: an event recorded there becomes a graph node and carries no timestamp)
    if _EVENT_RECORDER is None or torch.cuda.is_current_stream_capturing():
        return _Range(name) if _ROCTX else _NullCtx()
    return _Range(name, _EVENT_RECORDER(name)) if _ROCTX else _EVENT_RECORDER(name)


_DEBUG = __import__("os").environ.get("MOMA_DEBUG", "0") == "1"
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)       # the handle without building a Stream object


def _stream() -> C.c_void_p:
    # (called once per library call, ~1000x per training step: torch.cuda.current_stream() costs ~12 us of host time each)
    if _RAW_STREAM is not None:
        return C.c_void_p(_RAW_STREAM(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


# ------------------------------------------------------------------------------------------------
# K4 multi-tensor EMA   (learning/contrast_trainer.py:207-211)
# ------------------------------------------------------------------------------------------------
class EmaTable:
    """Device-side pointer table for one (model, model_ema) pair; rebuilt if any tensor moves."""

    def __init__(self, params: Sequence[torch.Tensor], params_ema: Sequence[torch.Tensor]):
        params, params_ema = list(params), list(params_ema)
        # zip() semantics of the reference: stops at the shorter list; shapes must match pairwise
        n = min(len(params), len(params_ema))
        rows, first = [], 0
        self.keys = []
        for p, e in zip(params[:n], params_ema[:n]):
            if p.shape != e.shape:
                raise RuntimeError(f"The size of tensor a {tuple(e.shape)} must match the size of tensor b "
                                   f"{tuple(p.shape)} (momentum_update needs identical architectures)")
            _dense_pair(p, e)
            if e.numel() == 0:
                continue
            rows.append((e.data_ptr(), p.data_ptr(), e.numel(), first))
            first += (e.numel() + EMA_BLOCK_ELEMS - 1) // EMA_BLOCK_ELEMS
            self.keys.append((e.data_ptr(), p.data_ptr(), e.numel()))
        self.n = len(rows)
        self.total_blocks = first
        dev = params[0].device if n else torch.device("cuda")
        self.table = (torch.tensor(rows, dtype=torch.int64).reshape(-1, 4).to(dev) if rows
                      else torch.zeros(0, 4, dtype=torch.int64, device=dev))

    def matches(self, params, params_ema) -> bool:
        keys = [(e.data_ptr(), p.data_ptr(), e.numel()) for p, e in zip(params, params_ema) if e.numel()]
        return keys == self.keys


def ema_update_(table: EmaTable, m: float) -> None:
    """ema <- fma(fl32(1-m), p, ema*fl32(m)) for every tensor of the table, one launch."""
    lib = _lib.load()
    if table.n == 0:
        return
    with _timed("moma_ema_multi"):
        check(lib.moma_ema_multi(_ptr(table.table), table.n, table.total_blocks, float(m), float(1.0 - m), _stream()),
              "moma_ema_multi")


# ------------------------------------------------------------------------------------------------
# K3 ring-buffer enqueue   (MoMA/mem_moco.py:17-27)
# ------------------------------------------------------------------------------------------------
def enqueue_(queue: torch.Tensor, rows: torch.Tensor, index: int) -> None:
    lib = _lib.load()
    _dev(queue, "queue", None); _dev(rows, "rows")
    K, d = queue.shape
    if rows.dim() != 2 or rows.shape[1] != d:
        raise ValueError(f"rows must be [n,{d}], got {tuple(rows.shape)}")
    with trace_range("moma_enqueue"):
        check(lib.moma_enqueue(_ptr(queue), _ptr(rows), rows.shape[0], int(index), K, d, _qdtype(queue), _stream()),
              "moma_enqueue")


def queue_prefetch(queue: torch.Tensor, stream=None) -> None:
    """Cache hint (moma_queue_prefetch): sweep the queue into the Infinity Cache on `stream` (default: current)."""
    lib = _lib.load()
    _dev(queue, "queue", None)
    st = C.c_void_p(stream.cuda_stream) if stream is not None else _stream()
    check(lib.moma_queue_prefetch(_ptr(queue), queue.numel() * queue.element_size(), st), "moma_queue_prefetch")


def enqueue_mirror_(queue: torch.Tensor, mirror: torch.Tensor, rows: torch.Tensor, index: int) -> None:
    """fp32 queue + its bf16 mirror, one launch (moma_enqueue_mirror)."""
    lib = _lib.load()
    _dev(queue, "queue"); _dev(mirror, "mirror", torch.bfloat16); _dev(rows, "rows")
    K, d = queue.shape
    if mirror.shape != queue.shape or rows.dim() != 2 or rows.shape[1] != d:
        raise ValueError(f"rows must be [n,{d}] and mirror {tuple(queue.shape)}, got {tuple(rows.shape)} / {tuple(mirror.shape)}")
    with trace_range("moma_enqueue_mirror"):
        check(lib.moma_enqueue_mirror(_ptr(queue), _ptr(mirror), _ptr(rows), rows.shape[0], int(index), K, d, _stream()),
              "moma_enqueue_mirror")


# ------------------------------------------------------------------------------------------------
# K2 InfoNCE
# ------------------------------------------------------------------------------------------------
def _check_qk(q, k, queue):
    _dev(q, "q"); _dev(k, "k"); _dev(queue, "queue", None)
    if q.dim() != 2 or q.shape != k.shape or queue.dim() != 2 or queue.shape[1] != q.shape[1]:
        raise ValueError(f"shape mismatch: q {tuple(q.shape)} k {tuple(k.shape)} queue {tuple(queue.shape)}")


class _InfoNCELogits(torch.autograd.Function):
    """[B,K+1] logits of BaseMoCo._compute_logit (MoMA/mem_moco.py:29-49) with the backward w.r.t. q."""

    @staticmethod
    def forward(ctx, q, k, queue, T, prec):
        lib = _lib.load()
        q = q.contiguous(); k = k.contiguous()
        _check_qk(q, k, queue)
        B, d = q.shape
        K = queue.shape[0]
        out = torch.empty(B, K + 1, device=q.device, dtype=torch.float32)
        check(lib.moma_infonce_logits(_ptr(q), _ptr(k), _ptr(queue), _ptr(out), B, d, K, float(1.0 / T),
                                      _qdtype(queue), prec, _stream()), "moma_infonce_logits")
        ctx.save_for_backward(q, k, queue)
        ctx.T, ctx.prec = T, prec
        return out

    @staticmethod
    def backward(ctx, dlogits):
        lib = _lib.load()
        q, k, queue = ctx.saved_tensors
        dlogits = dlogits.contiguous()
        B, d = k.shape
        K = queue.shape[0]
        dq = dk = dqueue = None
        if ctx.needs_input_grad[0]:
            dq = torch.empty(B, d, device=k.device, dtype=torch.float32)
            # (the form with a workspace: the split-K partials are added in a fixed order -- bitwise reproducible)
            ws = torch.empty(lib.moma_infonce_logits_bwd_workspace_bytes(B, d, K), device=k.device, dtype=torch.uint8)
            check(lib.moma_infonce_logits_bwd_ws(_ptr(dlogits), _ptr(k), _ptr(queue), _ptr(dq), B, d, K,
                                                 float(1.0 / ctx.T), _qdtype(queue), ctx.prec, _ptr(ws), ws.numel(), _stream()),
                  "moma_infonce_logits_bwd_ws")
        # k / queue carry gradient only in the MoCoAtt cross-attention variants (they are attention outputs there)
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            if ctx.needs_input_grad[1]:
                dk = torch.empty(B, d, device=k.device, dtype=torch.float32)
            if ctx.needs_input_grad[2]:
                if queue.dtype != torch.float32:
                    raise TypeError("a queue that requires grad must be float32")
                dqueue = torch.empty(K, d, device=k.device, dtype=torch.float32)
            check(lib.moma_infonce_logits_bwd_kq(_ptr(dlogits), _ptr(q), _ptr(dk), _ptr(dqueue), B, d, K,
                                                 float(1.0 / ctx.T), ctx.prec, _stream()), "moma_infonce_logits_bwd_kq")
        return dq, dk, dqueue, None, None


def infonce_logits(q, k, queue, T: float, prec="fp32") -> torch.Tensor:
    return _InfoNCELogits.apply(q, k, queue, float(T), prec_code(prec))


class QPack:
    """The query of K2 in the packed bf16 MFMA-operand layout (moma_infonce_fused_q), written by the producer of q -- the proj
    epilogue of the attention module atts_q (moma_mha_fwd_fast) -- so that K2 runs no pre-pack launch.  One persistent buffer
    per (B, d): its pad rows stay zero.  `src` remembers which q tensor / temperature the image was made from; K2 ignores
    the image unless it is handed exactly that tensor.
    Caveat (as for the weight packs, Attention.invalidate_pack): the identity check is (data_ptr, autograd version, shape, T).  A
    write to q that bypasses the version counter -- arithmetic on `q.data`, a raw-pointer kernel -- between the producer and K2
    leaves a stale image that K2 would pair with the new fp32 q (positive logit and dq come from the fp32 tensor).  Nothing in
    the loop does that; a caller that does must drop the image (`qpack.src = None`) or pass `qpack=None` to K2.  MOMA_DEBUG=1
    re-packs q inside infonce_fused and compares (a test hook, costs a launch and a sync)."""

    def __init__(self):
        self.buf, self.shape, self.scale, self.src = None, None, None, None

    def prepare(self, B: int, d: int, T: float, device):
        """-> this holder, ready to be handed to Attention.forward(x, qpack=...) and then to infonce_fused(..., qpack=...); None
        when K2 takes no pre-packed query at this width (d not in {128, 256, 384, 512})."""
        lib = _lib.load()
        nbytes = lib.moma_infonce_qpack_bytes(B, d)
        if nbytes == 0:
            return None
        if self.buf is None or self.shape != (B, d) or self.buf.device != device:
            self.buf = torch.zeros(nbytes, device=device, dtype=torch.uint8)
            self.shape = (B, d)
        self.scale, self.T = (1.0 / T) * 1.4426950408889634, float(T)
        self.src = None
        return self

    def written_from(self, q: torch.Tensor, T: float):
        self.src = (q.data_ptr(), q._version, tuple(q.shape), float(T))

    def matches(self, q: torch.Tensor, T: float) -> bool:
        return self.buf is not None and self.src == (q.data_ptr(), q._version, tuple(q.shape), float(T))


class K2Buffers:
    """Caller-owned outputs + workspace of one moma_infonce_fused call at a fixed shape: what a step replayed from HIP graphs
    (helper/step_graph.py) hands to K2 every step -- the graphs on either side of the call read / write these addresses."""

    def __init__(self, B: int, d: int, K: int, queue_dtype, prec, device):
        lib = _lib.load()
        self.shape = (B, d, K)
        self.loss_rows = torch.empty(B, device=device, dtype=torch.float32)
        self.lse = torch.empty(B, device=device, dtype=torch.float32)
        self.top1 = torch.empty(B, device=device, dtype=torch.int32)
        self.dq = torch.empty(B, d, device=device, dtype=torch.float32)
        qd = DT_BF16 if queue_dtype == torch.bfloat16 else DT_F32
        self.ws = torch.empty(max(lib.moma_infonce_fused_workspace_bytes(B, d, K, qd, prec_code(prec)), 16), device=device,
                              dtype=torch.uint8)


def _infonce_fused_launch(q, k, queue, T, prec, qpack_buf, loss_rows, lse, top1, dq, ws, enq=None):
    """enq: None, or (rows [n,d] fp32, ring pointer, fp32 queue whose bf16 mirror `queue` is | None) -- the enqueue that follows
    the call (MoMA/mem_moco.py:97-99) rides on the call's last launch (moma_infonce_fused_enqueue)"""
    lib = _lib.load()
    B, d = q.shape
    K = queue.shape[0]
    ev0, ev1, ev2 = _KERNEL_EVENTS() if _KERNEL_EVENTS is not None else (None, None, None)
    with _timed("moma_infonce_fused"):
        if enq is None:
            check(lib.moma_infonce_fused_q(_ptr(q), _ptr(qpack_buf), _ptr(k), _ptr(queue), B, d, K, float(1.0 / T),
                                           _ptr(loss_rows), _ptr(lse), _ptr(top1), _ptr(dq), _ptr(ws), ws.numel(), _qdtype(queue), prec,
                                           _stream(), C.c_void_p(ev0), C.c_void_p(ev1), C.c_void_p(ev2)),
                  "moma_infonce_fused")
        else:
            rows, index, queue_f32 = enq
            _dev(rows, "rows")
            if rows.dim() != 2 or rows.shape[1] != d or not rows.is_contiguous():
                raise ValueError(f"rows must be contiguous [n,{d}], got {tuple(rows.shape)}")
            if queue_f32 is not None:
                _dev(queue_f32, "queue_f32")
                if queue_f32.shape != queue.shape or queue.dtype != torch.bfloat16:
                    raise ValueError("queue_f32 goes with its bf16 mirror of the same shape as `queue`")
            check(lib.moma_infonce_fused_enqueue(_ptr(q), _ptr(qpack_buf), _ptr(k), _ptr(queue), B, d, K, float(1.0 / T),
                                                 _ptr(loss_rows), _ptr(lse), _ptr(top1), _ptr(dq), _ptr(ws), ws.numel(), _qdtype(queue),
                                                 prec, _ptr(rows), rows.shape[0], int(index), _ptr(queue_f32), _stream(),
                                                 C.c_void_p(ev0), C.c_void_p(ev1), C.c_void_p(ev2)),
                  "moma_infonce_fused_enqueue")


def infonce_fused_into(q, k, queue, T: float, prec, qpack_buf, out: K2Buffers, enq=None) -> None:
    """moma_infonce_fused into caller-owned buffers, no autograd: loss_rows / lse / top1 / dq (= d sum(loss_rows) / dq) land in
    `out`.  qpack_buf: the packed image of q its producer wrote (or None: K2 packs q itself).  enq: see _infonce_fused_launch."""
    q = q.detach(); k = k.detach()
    _check_qk(_dev(q, "q"), _dev(k, "k"), queue)
    if (q.shape[0], q.shape[1], queue.shape[0]) != out.shape:
        raise ValueError(f"K2Buffers were made for (B, d, K) = {out.shape}, got {(q.shape[0], q.shape[1], queue.shape[0])}")
    pc = prec_code(prec)
    if qpack_buf is not None and not (queue.dtype == torch.bfloat16 and pc == PREC_BF16):
        qpack_buf = None
    _infonce_fused_launch(q, k, queue, float(T), pc, qpack_buf, out.loss_rows, out.lse, out.top1, out.dq, out.ws, enq)


class StaticK2Loss(torch.autograd.Function):
    """The autograd node that stands for a K2 call made OUTSIDE the captured graphs: forward hands out the loss rows K2 left in
    its static buffer, backward is K2's own (dq * upstream, as _InfoNCEFused.backward)."""

    @staticmethod
    def forward(ctx, q, loss_rows_buf, dq_buf):
        ctx.save_for_backward(dq_buf)
        return loss_rows_buf.view_as(loss_rows_buf)

    @staticmethod
    def backward(ctx, g_loss):
        (dq,) = ctx.saved_tensors
        return dq * g_loss.unsqueeze(1), None, None


class _InfoNCEFused(torch.autograd.Function):
    """One pass over the queue: per-row CE(label 0) loss, lse, top-1 flag and d(sum loss)/dq."""

    @staticmethod
    def forward(ctx, q, k, queue, T, prec, qpack=None, enq=None):
        lib = _lib.load()
        q = q.contiguous(); k = k.contiguous()
        _check_qk(q, k, queue)
        B, d = q.shape
        K = queue.shape[0]
        dev = q.device
        need_grad = ctx.needs_input_grad[0]
        loss_rows = torch.empty(B, device=dev, dtype=torch.float32)
        lse = torch.empty(B, device=dev, dtype=torch.float32)
        top1 = torch.empty(B, device=dev, dtype=torch.int32)
        dq = torch.empty(B, d, device=dev, dtype=torch.float32) if need_grad else None
        ws_bytes = lib.moma_infonce_fused_workspace_bytes(B, d, K, _qdtype(queue), prec)
        ws = torch.empty(max(ws_bytes, 16), device=dev, dtype=torch.uint8)
        _infonce_fused_launch(q, k, queue, T, prec, qpack, loss_rows, lse, top1, dq, ws, enq)
        if need_grad:
            ctx.save_for_backward(dq)
        ctx.mark_non_differentiable(lse, top1)
        return loss_rows, lse, top1

    @staticmethod
    def backward(ctx, g_loss, g_lse, g_top1):
        (dq,) = ctx.saved_tensors
        return dq * g_loss.unsqueeze(1), None, None, None, None, None, None


def debug_set_k2_target_wg(n: int) -> int:
    """Plan sweeps only (scripts/sweep_k2_plan.sh): cut K2's passes over the queue into about n workgroups (0 = the product's own
    plan).  Set it BEFORE the first call of the process: cached workspaces are sized under the plan in force when they were made.
    -> the previous value."""
    prev = lib.moma_debug_set_k2_target_wg(int(n))
    if prev < 0:
        raise ValueError(f"moma_debug_set_k2_target_wg({n}): refused (0, or 8 .. 1024)")
    return prev


def infonce_fused(q, k, queue, T: float, prec="fp32", qpack: "QPack | None" = None, enq=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """-> (loss_rows [B], lse [B], top1 [B] int32).  loss_kd = loss_rows.mean().
    qpack: a QPack that the producer of q filled (ignored unless it was written from exactly this q and T).
    enq: None, or (rows, ring pointer, fp32 queue | None): the enqueue behind the pass rides on the call (_infonce_fused_launch)."""
    buf = None
    if qpack is not None and q.is_contiguous() and qpack.matches(q, T) and queue.dtype == torch.bfloat16 \
            and prec_code(prec) == PREC_BF16:
        buf = qpack.buf
        if _DEBUG:
            # the image K2 is about to trust against one made from q as it stands now (own pre-pack: a forward-only call without it)
            with torch.no_grad():
                a = _InfoNCEFused.apply(q.detach(), k.detach(), queue, float(T), prec_code(prec), buf)[1]
                b = _InfoNCEFused.apply(q.detach(), k.detach(), queue, float(T), prec_code(prec), None)[1]
            if not torch.equal(a, b):
                raise MomaHipError("QPack: the packed image of q does not match q (written behind autograd's version counter?)")
    return _InfoNCEFused.apply(q, k, queue, float(T), prec_code(prec), buf, enq)


class _InfoNCEFusedMulti(torch.autograd.Function):
    """n InfoNCE terms (q_i, k_i, queue_i) in one sweep (moma_infonce_fused_multi): inputs q_0, k_0, queue_0, q_1, ...;
    outputs loss_rows_0, lse_0, top1_0, loss_rows_1, ..."""

    @staticmethod
    def forward(ctx, T, prec, *tensors):
        lib = _lib.load()
        n = len(tensors) // 3
        qs = [tensors[3 * i].contiguous() for i in range(n)]
        ks = [tensors[3 * i + 1].contiguous() for i in range(n)]
        queues = [tensors[3 * i + 2] for i in range(n)]
        for q, k, queue in zip(qs, ks, queues):
            _check_qk(q, k, queue)
        B, d = qs[0].shape
        K = queues[0].shape[0]
        dev = qs[0].device
        need_grad = any(ctx.needs_input_grad[2 + 3 * i] for i in range(n))
        qd = _qdtype(queues[0])
        ws = torch.empty(max(lib.moma_infonce_fused_multi_workspace_bytes(n, B, d, K, qd, prec), 16), device=dev, dtype=torch.uint8)
        terms = (_lib.InfoNCETerm * n)()
        outs, dqs = [], []
        for i in range(n):
            # (two terms with the same query tensor share its packed image: the library dedupes by pointer)
            loss_rows = torch.empty(B, device=dev, dtype=torch.float32)
            lse = torch.empty(B, device=dev, dtype=torch.float32)
            top1 = torch.empty(B, device=dev, dtype=torch.int32)
            dq = torch.empty(B, d, device=dev, dtype=torch.float32) if need_grad else None
            t = terms[i]
            t.q, t.k, t.queue = qs[i].data_ptr(), ks[i].data_ptr(), queues[i].data_ptr()
            t.loss_rows, t.lse, t.top1, t.dq = loss_rows.data_ptr(), lse.data_ptr(), top1.data_ptr(), (0 if dq is None else dq.data_ptr())
            outs += [loss_rows, lse, top1]
            dqs.append(dq)
        with _timed("moma_infonce_fused_multi"):
            check(lib.moma_infonce_fused_multi(C.cast(terms, C.c_void_p), n, B, d, K, float(1.0 / T), _ptr(ws), ws.numel(), qd,
                                               prec, _stream()), "moma_infonce_fused_multi")
        if need_grad:
            ctx.save_for_backward(*dqs)
        ctx.n = n
        ctx.mark_non_differentiable(*[o for i, o in enumerate(outs) if i % 3 != 0])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        dqs = ctx.saved_tensors
        res = [None, None]
        for i in range(ctx.n):
            g_loss = grads[3 * i]
            res += [dqs[i] * g_loss.unsqueeze(1) if (ctx.needs_input_grad[2 + 3 * i] and g_loss is not None) else None, None, None]
        return tuple(res)


def infonce_fused_multi(terms, T: float, prec="fp32"):
    """terms = [(q, k, queue), ...] over queues of one shape and dtype -> [(loss_rows, lse, top1), ...] through ONE library call
    (moma_infonce_fused_multi): one sweep where the one-pass bf16 kernel takes the configuration, term by term inside the
    library otherwise (wide rows, exact fp32).  Terms of different shapes: one moma_infonce_fused call each."""
    pc = prec_code(prec)
    q0, _, queue0 = terms[0]
    same = all(q.shape == q0.shape and queue.shape == queue0.shape and queue.dtype == queue0.dtype for q, _, queue in terms)
    if len(terms) > 1 and len(terms) <= 4 and same and q0.is_cuda:
        flat = _InfoNCEFusedMulti.apply(float(T), pc, *[t for term in terms for t in term])
        return [tuple(flat[3 * i:3 * i + 3]) for i in range(len(terms))]
    return [infonce_fused(q, k, queue, T, prec) for q, k, queue in terms]


# ------------------------------------------------------------------------------------------------
# K1 batch-token multi-head attention   (MoMA/criterion_moco_att.py:153-167)
# ------------------------------------------------------------------------------------------------
class MhaPack:
    """bf16 weight pack of one attention module for the fast path ([Wqkv | Wproj | Wqkv^T | Wproj^T], moma_mha_pack_weights):
    rebuilt -- into a NEW buffer, a pending backward keeps the one its forward used -- when the fp32 weights were replaced or
    modified in place (optimizer.step(), load_state_dict: the tensors' version counters move).  A writer that bypasses
    autograd's version counter (a raw-pointer kernel) must call invalidate()."""

    def __init__(self):
        self.buf, self.key, self.has_t = None, None, False

    def invalidate(self):
        self.buf = None

    def get(self, w_qkv: torch.Tensor, w_proj: torch.Tensor, need_t: bool) -> torch.Tensor:
        key = (w_qkv.data_ptr(), w_qkv._version, w_proj.data_ptr(), w_proj._version, w_qkv.device)
        if self.buf is None or key != self.key or (need_t and not self.has_t):
            lib = _lib.load()
            d = w_proj.shape[0]
            buf = torch.empty(lib.moma_mha_pack_bytes(d), device=w_qkv.device, dtype=torch.uint8)
            check(lib.moma_mha_pack_weights(_ptr(w_qkv), _ptr(w_proj), _ptr(buf), d, int(need_t), _stream()),
                  "moma_mha_pack_weights")
            self.buf, self.key, self.has_t = buf, key, bool(need_t)
        return self.buf


def _mha_check(x, w_qkv, b_qkv, w_proj, b_proj, H):
    _dev(x, "x", None)
    if x.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError(f"x: expected float32 or bfloat16, got {x.dtype}")
    for t, nm in ((w_qkv, "qkv.weight"), (w_proj, "proj.weight"), (b_proj, "proj.bias")):
        _dev(t, nm)
    if b_qkv is not None:
        _dev(b_qkv, "qkv.bias")
    N, d = x.shape
    if w_qkv.shape != (3 * d, d) or w_proj.shape != (d, d) or d % H:
        raise ValueError(f"bad attention shapes: x {tuple(x.shape)} Wqkv {tuple(w_qkv.shape)} H={H}")
    return N, d


def _mha_fast_fwd(items, H, want_lse, qpacks=None):
    """One grouped moma_mha_fwd_fast call.  items: [(x, pack, b_qkv, b_proj)], equal shapes.  -> [(y, qkv16, attn16, lse)]"""
    lib = _lib.load()
    N, d = items[0][0].shape
    dev = items[0][0].device
    mods = (_lib.MhaModule * len(items))()
    outs = []
    for i, (x, pack, b_qkv, b_proj) in enumerate(items):
        y = torch.empty(N, d, device=dev, dtype=torch.float32)
        qkv16 = torch.empty(N, 3 * d, device=dev, dtype=torch.bfloat16)
        attn16 = torch.empty(N, d, device=dev, dtype=torch.bfloat16)
        lse = torch.empty(H, N, device=dev, dtype=torch.float32) if want_lse else None
        qp = qpacks[i] if qpacks is not None else None       # a prepared QPack
        m = mods[i]
        m.x, m.pack, m.b_qkv, m.b_proj = x.data_ptr(), pack.data_ptr(), (0 if b_qkv is None else b_qkv.data_ptr()), b_proj.data_ptr()
        m.y, m.qkv16, m.attn16, m.lse = y.data_ptr(), qkv16.data_ptr(), attn16.data_ptr(), (0 if lse is None else lse.data_ptr())
        m.qpack, m.qpack_scale = (0, 0.0) if qp is None else (qp.buf.data_ptr(), float(qp.scale))
        m.x_dtype = DT_BF16 if x.dtype == torch.bfloat16 else DT_F32
        outs.append((y, qkv16, attn16, lse))
    with _timed("moma_mha_fwd" if len(items) == 1 else f"moma_mha_fwd_group{len(items)}"):
        check(lib.moma_mha_fwd_fast(C.cast(mods, C.c_void_p), len(items), N, d, H, _stream()), "moma_mha_fwd_fast")
    return outs


class _MHA(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_qkv, b_qkv, w_proj, b_proj, H, prec, grad_mode, holder, qpack):
        lib = _lib.load()
        x = x.contiguous()
        N, d = _mha_check(x, w_qkv, b_qkv, w_proj, b_proj, H)
        dev = x.device
        ctx.x_dtype = x.dtype
        need_bwd = grad_mode and any(ctx.needs_input_grad)      # (grad mode is always off inside forward)
        fast = lib.moma_mha_saved_state(N, d, H, prec) == _lib.MHA_SAVE_LSE
        ctx.H, ctx.prec, ctx.has_bqkv, ctx.fast = H, prec, b_qkv is not None, fast
        if fast:
            # bf16 fast path: the forward keeps qkv / attn_out as bf16 and the row log-sum-exp [H,N]; P is recomputed per tile
            pack = (holder if holder is not None else MhaPack()).get(w_qkv, w_proj, need_bwd)
            (y, qkv16, attn16, lse), = _mha_fast_fwd([(x, pack, b_qkv, b_proj)], H, need_bwd,
                                                     None if qpack is None else [qpack])
            if need_bwd:
                ctx.save_for_backward(x, pack, qkv16, attn16, lse)
            if qpack is not None:
                qpack.written_from(y, qpack.T)
            return y
        x = x.float()                       # (a bf16 x is consumed as it stands by the fast path only; a packed-q request is dropped)
        # staged path (exact fp32, odd head dims): keeps the probabilities [H,N,N]
        y = torch.empty(N, d, device=dev, dtype=torch.float32)
        qkv = torch.empty(N, 3 * d, device=dev, dtype=torch.float32)
        probs = torch.empty(H, N, N, device=dev, dtype=torch.float32)
        attn_out = torch.empty(N, d, device=dev, dtype=torch.float32)
        with _timed("moma_mha_fwd"):
            check(lib.moma_mha_fwd(_ptr(x), _ptr(w_qkv), _ptr(b_qkv), _ptr(w_proj), _ptr(b_proj), _ptr(y), _ptr(qkv),
                                   _ptr(probs), _ptr(attn_out), N, d, H, prec, _stream()), "moma_mha_fwd")
        if need_bwd:
            ctx.save_for_backward(x, w_qkv, w_proj, qkv, probs, attn_out)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = dy.contiguous()
        H = ctx.H
        need = ctx.needs_input_grad
        if ctx.fast:
            x, pack, qkv16, attn16, lse = ctx.saved_tensors
        else:
            x, w_qkv, w_proj, qkv, probs, attn_out = ctx.saved_tensors
        N, d = x.shape
        dev = x.device
        dx = torch.empty(N, d, device=dev, dtype=torch.float32) if need[0] else None
        dw_qkv = torch.empty(3 * d, d, device=dev, dtype=torch.float32) if need[1] else None
        db_qkv = torch.empty(3 * d, device=dev, dtype=torch.float32) if (need[2] and ctx.has_bqkv) else None
        dw_proj = torch.empty(d, d, device=dev, dtype=torch.float32) if need[3] else None
        db_proj = torch.empty(d, device=dev, dtype=torch.float32) if need[4] else None
        if ctx.fast:
            # a bias gradient rides on its weight gradient's launch: compute the weight gradient too if only the bias wants one
            t_wq = dw_qkv if (dw_qkv is not None or db_qkv is None) else torch.empty(3 * d, d, device=dev, dtype=torch.float32)
            t_wp = dw_proj if (dw_proj is not None or db_proj is None) else torch.empty(d, d, device=dev, dtype=torch.float32)
            ws = torch.empty(lib.moma_mha_bwd_fast_workspace_bytes(N, d, H), device=dev, dtype=torch.uint8)
            with _timed("moma_mha_bwd"):
                check(lib.moma_mha_bwd_fast(_ptr(pack), _ptr(x), DT_BF16 if x.dtype == torch.bfloat16 else DT_F32, _ptr(qkv16), _ptr(attn16), _ptr(lse), _ptr(dy), _ptr(dx),
                                            _ptr(t_wq), _ptr(db_qkv), _ptr(t_wp), _ptr(db_proj), _ptr(ws), ws.numel(), N, d, H,
                                            _stream()), "moma_mha_bwd_fast")
        else:
            ws = torch.empty(lib.moma_mha_bwd_workspace_bytes(N, d, H, ctx.prec), device=dev, dtype=torch.uint8)
            with _timed("moma_mha_bwd"):
                check(lib.moma_mha_bwd(_ptr(x), _ptr(w_qkv), _ptr(w_proj), _ptr(qkv), _ptr(probs), _ptr(attn_out),
                                       _ptr(dy), _ptr(dx), _ptr(dw_qkv), _ptr(db_qkv), _ptr(dw_proj), _ptr(db_proj), _ptr(ws),
                                       ws.numel(), N, d, H, ctx.prec, _stream()), "moma_mha_bwd")
        if dx is not None and dx.dtype != ctx.x_dtype:
            dx = dx.to(ctx.x_dtype)
        return dx, dw_qkv, db_qkv, dw_proj, db_proj, None, None, None, None, None


def mha(x, w_qkv, b_qkv, w_proj, b_proj, num_heads: int, prec="fp32", pack: "MhaPack | None" = None, qpack=None) -> torch.Tensor:
    """Attention.forward.  pack: the module's MhaPack (bf16 weight cache of the fast path; a throw-away one is built when
    omitted).  qpack: None, or a QPack prepared for (N, d, T): the fast path also writes y in K2's packed-Q layout (on the
    staged path the request is dropped and K2 packs q itself)."""
    return _MHA.apply(x, w_qkv, b_qkv, w_proj, b_proj, int(num_heads), prec_code(prec), torch.is_grad_enabled(), pack, qpack)


def mha_group(calls, num_heads: int, prec="fp32"):
    """Forward of several attention modules in ONE group of launches (no autograd): calls = [(x, w_qkv, b_qkv, w_proj,
    b_proj, pack)].  Falls back to one call per module when the fast path does not take the configuration, the shapes
    differ, or a gradient is wanted."""
    lib = _lib.load()
    pc = prec_code(prec)
    xs = [c[0].contiguous() for c in calls]
    same = all(x.shape == xs[0].shape for x in xs) and 1 < len(calls) <= 4
    wants_grad = torch.is_grad_enabled() and any(t is not None and t.requires_grad for c in calls for t in c[:5])
    if not same or wants_grad or lib.moma_mha_saved_state(xs[0].shape[0], xs[0].shape[1], int(num_heads), pc) != _lib.MHA_SAVE_LSE:
        return [mha(c[0], c[1], c[2], c[3], c[4], num_heads, prec, c[5]) for c in calls]
    items = []
    for x, c in zip(xs, calls):
        _mha_check(x, c[1], c[2], c[3], c[4], int(num_heads))
        holder = c[5] if c[5] is not None else MhaPack()
        items.append((x, holder.get(c[1], c[3], False), c[2], c[4]))
    return [o[0] for o in _mha_fast_fwd(items, int(num_heads), False)]


# ------------------------------------------------------------------------------------------------
# BatchNorm2d + fused activation on NCHW activations (backbone helper, include/moma_hip.h "BN")
# ------------------------------------------------------------------------------------------------
ACT_CODES = {None: 0, "none": 0, "silu": 1, "relu": 2}
_DT_CODES = {torch.float32: 0, torch.bfloat16: 1}


class _BNAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, act, want_mean):
        lib = _lib.load()
        _dev(x, "x", dtype=None, contiguous=False)
        if x.dim() < 2 or x.dtype not in _DT_CODES:
            raise ValueError(f"bn_act: expects [N, C, ...] float32 / bfloat16, got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        N, Cc = x.shape[0], x.shape[1]
        HW = x.numel() // (N * Cc)
        dev = x.device
        out = torch.empty_like(x)
        save_mean = torch.empty(Cc, device=dev, dtype=torch.float32)
        save_invstd = torch.empty(Cc, device=dev, dtype=torch.float32)
        ws = torch.empty(lib.moma_bn_workspace_bytes(Cc), device=dev, dtype=torch.uint8)
        pmean = torch.empty(N, Cc, 1, 1, device=dev, dtype=x.dtype) if want_mean else None
        check(lib.moma_bn_fwd(_ptr(x), _ptr(out), _ptr(weight), _ptr(bias), _ptr(running_mean), _ptr(running_var),
                              _ptr(save_mean), _ptr(save_invstd), _ptr(ws), ws.numel(), N, Cc, HW, _DT_CODES[x.dtype],
                              act, int(training), float(momentum), float(eps), _ptr(pmean), _stream()), "moma_bn_fwd")
        ctx.save_for_backward(x, weight, bias, save_mean, save_invstd)
        ctx.cfg = (N, Cc, HW, act, int(training))
        ctx.set_materialize_grads(False)
        return (out, pmean) if want_mean else out

    @staticmethod
    def backward(ctx, dout, dmean=None):
        lib = _lib.load()
        x, weight, bias, save_mean, save_invstd = ctx.saved_tensors
        N, Cc, HW, act, training = ctx.cfg
        if dout is None:
            dout = torch.zeros_like(x)
        dout = dout.contiguous()
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        if dmean is not None:
            dmean = dmean.to(x.dtype).contiguous()
        dev = x.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dgamma = torch.empty(Cc, device=dev, dtype=torch.float32) if (weight is not None and ctx.needs_input_grad[1]) else None
        dbeta = torch.empty(Cc, device=dev, dtype=torch.float32) if (bias is not None and ctx.needs_input_grad[2]) else None
        ws = torch.empty(lib.moma_bn_workspace_bytes(Cc), device=dev, dtype=torch.uint8)
        check(lib.moma_bn_bwd(_ptr(x), _ptr(dout), _ptr(weight), _ptr(bias), _ptr(save_mean), _ptr(save_invstd), _ptr(dx),
                              _ptr(dgamma), _ptr(dbeta), _ptr(ws), ws.numel(), N, Cc, HW, _DT_CODES[x.dtype], act,
                              training, _ptr(dmean), _stream()), "moma_bn_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


def bn_act(x, weight, bias, running_mean, running_var, training: bool, momentum: float, eps: float, act=None,
           want_mean: bool = False):
    """act(batch_norm(x)) on a contiguous NCHW tensor; running statistics are updated in place when training.
    want_mean: also return the [N,C,1,1] per-plane mean of the result (the squeeze of a squeeze-excite block)."""
    return _BNAct.apply(x, weight, bias, running_mean, running_var, bool(training), momentum, eps, ACT_CODES[act],
                        bool(want_mean))


# ------------------------------------------------------------------------------------------------
# Depthwise convolution on NCHW activations (backbone helper, include/moma_hip.h "DW")
# ------------------------------------------------------------------------------------------------
class _DWConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad_top, pad_left, out_h, out_w):
        lib = _lib.load()
        _dev(x, "x", dtype=None, contiguous=False)
        _dev(w, "weight")
        if x.dim() != 4 or x.dtype not in _DT_CODES or w.dim() != 4 or w.shape[1] != 1 or w.shape[0] != x.shape[1] \
                or w.shape[2] != w.shape[3]:
            raise ValueError(f"dwconv: x {tuple(x.shape)} {x.dtype}, weight {tuple(w.shape)}")
        x = x.contiguous()
        w = w.contiguous()
        N, Cc, H, W = x.shape
        K = w.shape[2]
        y = torch.empty(N, Cc, out_h, out_w, device=x.device, dtype=x.dtype)
        check(lib.moma_dwconv_fwd(_ptr(x), _ptr(w), _ptr(y), N, Cc, H, W, out_h, out_w, K, stride, pad_top, pad_left,
                                  _DT_CODES[x.dtype], _stream()), "moma_dwconv_fwd")
        ctx.save_for_backward(x, w)
        ctx.cfg = (N, Cc, H, W, out_h, out_w, K, stride, pad_top, pad_left)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        N, Cc, H, W, OH, OW, K, stride, pt, pl = ctx.cfg
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib.moma_dwconv_bwd_data(_ptr(dy), _ptr(w), _ptr(dx), N, Cc, H, W, OH, OW, K, stride, pt, pl,
                                           _DT_CODES[x.dtype], _stream()), "moma_dwconv_bwd_data")
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            ws = torch.empty(lib.moma_dwconv_workspace_bytes(Cc, K), device=x.device, dtype=torch.uint8)
            check(lib.moma_dwconv_bwd_weight(_ptr(x), _ptr(dy), _ptr(dw), _ptr(ws), ws.numel(), N, Cc, H, W, OH, OW, K,
                                             stride, pt, pl, _DT_CODES[x.dtype], _stream()), "moma_dwconv_bwd_weight")
        return dx, dw, None, None, None, None, None


def dwconv_supported(kernel: int, stride: int) -> bool:
    return kernel in (3, 5) and stride in (1, 2)


def dwconv(x, weight, stride: int, pad_top: int, pad_left: int, out_h: int, out_w: int):
    """Depthwise conv2d, weight [C,1,K,K] fp32, explicit (possibly asymmetric) zero padding given by the top/left pad
    and the output size."""
    return _DWConv.apply(x, weight, int(stride), int(pad_top), int(pad_left), int(out_h), int(out_w))


# ------------------------------------------------------------------------------------------------
# Squeeze-excite helpers (include/moma_hip.h "SE")
# ------------------------------------------------------------------------------------------------
class _PlaneMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        _dev(x, "x", dtype=None, contiguous=False)
        if x.dim() != 4 or x.dtype not in _DT_CODES:
            raise ValueError(f"plane_mean: expects [N,C,H,W] float32 / bfloat16, got {tuple(x.shape)} {x.dtype}")
        x = x.contiguous()
        N, Cc, H, W = x.shape
        out = torch.empty(N, Cc, 1, 1, device=x.device, dtype=x.dtype)
        check(lib.moma_plane_mean(_ptr(x), _ptr(out), N * Cc, H * W, _DT_CODES[x.dtype], _stream()), "moma_plane_mean")
        ctx.shape = (N, Cc, H, W)
        return out

    @staticmethod
    def backward(ctx, dmean):
        N, Cc, H, W = ctx.shape
        return (dmean / (H * W)).expand(N, Cc, H, W)          # a stride-0 view: nothing is materialised here


class _SEGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        lib = _lib.load()
        _dev(x, "x", dtype=None, contiguous=False)
        _dev(s, "s", dtype=None, contiguous=False)
        if x.dim() != 4 or x.dtype not in _DT_CODES or s.numel() != x.shape[0] * x.shape[1]:
            raise ValueError(f"se_gate: x {tuple(x.shape)} {x.dtype}, s {tuple(s.shape)}")
        x = x.contiguous()
        s = s.to(x.dtype).contiguous()
        N, Cc, H, W = x.shape
        out = torch.empty_like(x)
        check(lib.moma_se_gate_fwd(_ptr(x), _ptr(s), _ptr(out), N * Cc, H * W, _DT_CODES[x.dtype], _stream()), "moma_se_gate_fwd")
        ctx.save_for_backward(x, s)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, s = ctx.saved_tensors
        N, Cc, H, W = x.shape
        dout = dout.contiguous()
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        dx = torch.empty_like(x)
        ds = torch.empty_like(s)
        check(lib.moma_se_gate_bwd(_ptr(x), _ptr(s), _ptr(dout), _ptr(dx), _ptr(ds), N * Cc, H * W, _DT_CODES[x.dtype],
                                   _stream()), "moma_se_gate_bwd")
        return dx, ds


def plane_mean(x):
    """[N,C,H,W] -> [N,C,1,1] mean over each plane (F.adaptive_avg_pool2d(x, 1))."""
    return _PlaneMean.apply(x)


def se_gate(x, s):
    """x * sigmoid(s) with s [N,C,1,1]."""
    return _SEGate.apply(x, s)
