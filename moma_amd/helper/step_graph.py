"""The MoMA training step replayed from HIP graphs (the host out of the step).

A step of the reference loop (helper/loops_moma.py:256-361) is ~2400 kernel launches; issued one by one through Python / ATen /
MIOpen that is ~35 ms of host time against ~40 ms of GPU time at B = 256 -- host-bound at B <= 190, and the first thing eight
ranks sharing one host run out of.  The step has fixed shapes and a fixed launch sequence, so after a few eager steps it is
captured ONCE into four HIP graphs and replayed, on the same two streams and with the same joins as the eager loop:

    side stream : g_teacher  teacher forward #1, K4 EMA, Shuffle-BN key encoding, K1 key side (all no-grad)
    main stream : g_student  the student's forward                                 -- concurrent with g_teacher
                  (join)     main waits for the side stream (opt.prefetch_queue, off by default: a queue sweep on the side stream)
                  g_query    CE + KL, embed_s, K1 query side (leaves q packed for K2), the student's top-1
                  eager      K2 (one pass over the queue, into static buffers) + K3 (enqueue): the ring pointer stays the HOST
                             integer of the contract (a launch argument of a captured kernel is frozen, the pointer moves every
                             step), and the measurement events bench.py puts on K2's dispatches keep working (an event recorded
                             inside a capture is a graph node without a timestamp; external event nodes are refused by this
                             runtime -- scripts/diag_graph_events.py)
                  g_bwd      weighted loss, backward (K1 backward included); ops.StaticK2Loss is the autograd node that stands
                             for the K2 call
                  eager      the data-parallel gradient all-reduce (learning/ddp.py: ONE flat collective), optimizer.step()

Round 6, the widened loop paths: --attn self_mix / self_nomix run their key encoding (attention in front of the un-shuffle,
learning/contrast_trainer.py:_shuffle_bn_attn) inside g_query and let K2 pack q itself; --mem MoCoAtt has no one-pass K2 -- its
cross-attention variant, the logits against a snapshot of the queue and their CrossEntropy are part of g_query
(MoCoAtt.forward_logits), and the eager part between the graphs is the enqueue alone.

Why not ONE graph with the teacher side as a forked branch: measured (round 4, B = 256) a graph with the fork runs the step in
43.5 ms against 40.1 ms eager and 42.0 ms for either without the second stream -- the runtime does not overlap the branches of
one graph, it only adds their synchronisation.  Two graphs replayed on two streams overlap as the eager chains do.
g_student / g_query / g_bwd share one memory pool (the backward consumes the activations the forward graphs saved), g_teacher has
its own; each family is captured on a stream of its own (library workspaces are per stream: nothing eager ever runs on the
capture streams) and replayed on the loop's main / side stream only.  What changes from step to step enters through static
inputs: the batch (copied in), the Shuffle-BN permutation (drawn from the HOST generator exactly as the eager loop draws it --
one per step -- and copied in through pinned memory: PermFeed), the weights (read in place; the attention weight packs are
rebuilt inside the graphs), the queue (read in place by the eager K2).  Host-side state the captured Python would have advanced
is advanced per replay: the BatchNorm batch counters, the `.grad` attributes (the graph's static gradient tensors are
re-attached after every replay -- an eager step in between drops them with zero_grad(set_to_none=True)).

A (shape, train / eval flags, queue storage) variant is captured after `warmup` eager steps (MIOpen / hipBLASLt have selected
their kernels by then); the first step of every epoch (teacher still in eval mode, reference :227) is a variant of its own that
is seen once per epoch and stays eager, as does a ragged last batch.  Capture uses CUDAGraph.capture_begin / capture_end
directly: `torch.cuda.graph` empties the caching allocator's pool on entry, and the next EAGER step would pay for it with
hundreds of milliseconds of hipMalloc (seen in round 3 with the teacher graphs).
"""
from __future__ import annotations

import gc
import os
import time

import torch

from .. import ops


class PermFeed:
    """The Shuffle-BN permutation as a static graph input: `push()` draws the step's permutation from the host generator
    (torch.randperm -- the reference's stream, learning/contrast_trainer.py:108) into a ring of pinned buffers and queues the
    copy into the device tensor the captured gather reads."""

    SLOTS = 4

    def __init__(self, n: int, device):
        self.n = n
        self.static = torch.zeros(n, dtype=torch.int64, device=device)
        self.static.copy_(torch.arange(n))
        self.pinned = [torch.empty(n, dtype=torch.int64).pin_memory() for _ in range(self.SLOTS)]
        self.done = [None] * self.SLOTS
        self.i = 0

    def take(self, n, device):
        if n != self.n or device != self.static.device:
            raise RuntimeError(f"captured step asked for a permutation of {n} on {device}, feed holds {self.n} on {self.static.device}")
        return self.static

    def push(self):
        j = self.i % self.SLOTS
        self.i += 1
        if self.done[j] is not None:
            self.done[j].synchronize()                   # (four steps back: long done -- the loop's own pacing keeps three queued)
        torch.randperm(self.n, out=self.pinned[j])
        self.static.copy_(self.pinned[j], non_blocking=True)
        ev = torch.cuda.Event(blocking=True)      # (never waited on in practice, see above; a blocking-sync event SPINS on this runtime too)
        ev.record()
        self.done[j] = ev


class _Captured:
    pass


class StepGraphs:
    """Serves MomaStep steps from captured graphs where it can; `step()` returns None when the caller has to run the step
    eagerly (variant not captured yet / not capturable)."""

    def __init__(self, warmup: int = 3, max_graphs: int = 2):
        self.warmup, self.max_graphs = warmup, max_graphs
        self.enabled = os.environ.get("MOMA_GRAPH_STUDENT", "1") == "1"
        self.graphs, self.seen = {}, {}
        self.st = None
        self.capture_stream = None
        self.replays = 0

    # ---- binding to the objects of a training run -------------------------------------------------------------------------
    def same_objects(self, st) -> bool:
        o = self.st
        return o is not None and o.model_s is st.model_s and o.model_t is st.model_t and o.criterion_kd is st.criterion_kd \
            and o.contrast is st.contrast and o.optimizer is st.optimizer and o.trainer is st.trainer

    def bind(self, st) -> None:
        """`st`: the epoch's MomaStep (a new object per epoch over the same models).  The captured graphs stay valid -- they
        hold device addresses, not the Python object."""
        self.st = st
        from .loops_moma import _unwrap
        self._mods = list(_unwrap(st.model_s).modules()) + list(st.model_t.modules()) + list(st.criterion_kd.modules())
        self._bn = [m for m in self._mods if hasattr(m, "_nbt_pending")]
        # every tensor the captured graphs read or write in place: parameters and buffers of the three module trees
        self._state = [t for m in (_unwrap(st.model_s), st.model_t, st.criterion_kd) for t in list(m.parameters()) + list(m.buffers())]

    def _streamed_queue(self):
        """the tensor K2 streams: `memory`, or the bf16 mirror of an fp32 `memory` under the bf16 policy (made here if absent)"""
        c = self.st.contrast
        if c.memory.dtype == torch.float32 and ops.prec_code(c.precision) == ops.PREC_BF16 and hasattr(c, "_bf16_shadow"):
            return c._bf16_shadow()
        return c.memory                                       # (MoCoAtt reads `memory` itself: a snapshot per step)

    def _key(self, images, labels):
        st = self.st
        # (queue storage is part of the key: the captured prefetch-free graphs do not read it, but the static K2 buffers are sized
        #  for it, and a replaced `memory` -- .cuda(), load_state_dict -- must not meet buffers of another shape)
        return (tuple(images.shape), images.dtype, images.is_contiguous(), tuple(labels.shape), labels.dtype,
                hash(tuple(m.training for m in self._mods)), st.contrast.memory.data_ptr(), self._streamed_queue().data_ptr(),
                tuple(st.contrast.memory.shape), st.contrast.memory.dtype,       # (a new queue may land on a freed queue's address)
                # the graphs hold ADDRESSES of parameters and buffers: a tensor that moved (.to(), .half(), a re-created module
                # attribute) makes this another variant -- never a replay through a stale pointer (~0.1 ms of host time per step)
                hash(tuple(t.data_ptr() for t in self._state)),
                st.contrast.T, st.amp_dtype, st.overlap, float(st.opt.cls), float(st.opt.div), float(st.opt.beta),
                float(st.opt.alpha), float(getattr(st.criterion_div, "T", 0.0)))

    # ---- one step ---------------------------------------------------------------------------------------------------------
    def _switch(self, replayed: bool, device):
        """The loop changes between replayed and eagerly issued steps (the step that opens an epoch, a ragged last batch, a variant
        not captured yet): the device drains first.  Round 6: with the teacher side on its own stream and the host running ahead
        (batches already on the device: nothing in the step blocks the host), an EAGER step issued behind replays still in flight
        never finished -- the process sat in the epoch's closing read-back for good (scripts/diag_equivalences.py, HANG=1: the
        second run of a process, EfficientNet pair, reproducible; not with one stream, not with a synchronisation in front of the
        step).  Cause inside the runtime not established; a switch happens a few times per epoch, the drain costs nothing there."""
        if getattr(self, "_replayed_last", None) not in (None, replayed) and os.environ.get("MOMA_GRAPH_SWITCH_DRAIN", "1") == "1":
            torch.cuda.synchronize(device)                 # (MOMA_GRAPH_SWITCH_DRAIN=0: the diagnostic that reproduces the hang)
        self._replayed_last = replayed

    def step(self, images, labels):
        if not self.enabled or not images.is_cuda:
            return None
        res = self._step(images, labels)
        if res is None:
            self._switch(False, images.device)
        return res

    def _step(self, images, labels):
        key = self._key(images, labels)
        cap = self.graphs.get(key)
        last, self._last_key = getattr(self, "_last_key", None), key
        if cap is None:
            # CONSECUTIVE sightings count: the variant that opens every epoch (teacher still in eval mode) comes once per epoch and
            # must never earn a capture of its own -- a second set of graph pools (13 GB reserved at B = 256) for one step per epoch
            n = (self.seen.get(key, 0) if key == last else 0) + 1
            if len(self.seen) < 64 or key in self.seen:
                self.seen[key] = n
            if n <= self.warmup:
                return None
            if len(self.graphs) >= self.max_graphs:
                # at capacity: the variant that has gone unused longest makes room (a variant keyed on a queue storage that was
                # replaced since -- .cuda(), load_state_dict, a new `memory` -- can never match again; holding on to it would
                # leave every later variant eager for good).  Its graphs are idle by now (this variant has just run `warmup`
                # eager steps); the synchronize makes that a fact before they are destroyed.
                torch.cuda.synchronize(images.device)
                victim = min(self.graphs, key=lambda k_: self.graphs[k_].last_used)
                del self.graphs[victim]
                self.seen.pop(victim, None)
                gc.collect()
            cap = self._capture(key, images, labels)
            if cap is None:
                return None
        cap.last_used = self.replays
        self._switch(True, images.device)
        return self._replay(cap, images, labels)

    def _throttle(self):
        """The host issues a replayed step in ~6 ms against tens of ms of GPU time.  Left alone it runs ahead until the runtime
        has no room for another launch (kernel-argument pool, hardware queue) and waits INSIDE hipGraphLaunch -- and on this
        platform every host-side wait of the runtime is a spin: hipEventSynchronize with a blocking-sync event, hipStreamSynchronize
        and the launch path all burn wall time = CPU time (scripts/diag_blocking_event.py; ROC_ACTIVE_WAIT_TIMEOUT=0 changes nothing)
        -- a full core per rank, eight of them on a node (round 4's driver line: 44.5 ms of CPU time in a 39.8 ms step).
        So the loop paces itself: before issuing step k it waits, by POLLING an event between short sleeps, for the event it
        recorded when it STARTED issuing step k-2 -- i.e. for the end of step k-3 (the event sits behind everything issued for the
        step before).  Steps k-2 and k-1 may still be in flight while k is issued: THREE steps queued at most (PermFeed's four
        slots cover that): the GPU never runs dry, the launches find room, and the host's CPU time per step is what issuing costs."""
        evs = self.__dict__.setdefault("_step_done", [])
        if len(evs) >= 2:
            ev = evs.pop(0)
            while not ev.query():
                time.sleep(0.0005)
        ev = torch.cuda.Event()
        ev.record()                                        # (behind everything the loop issued for the previous step, optimizer included)
        evs.append(ev)

    def _replay(self, cap, images, labels):
        st = self.st
        main = torch.cuda.current_stream(images.device)
        if os.environ.get("MOMA_GRAPH_THROTTLE", "1") == "1":
            self._throttle()
        cap.images.copy_(images, non_blocking=True)
        cap.labels.copy_(labels, non_blocking=True)
        cap.perm.push()
        wrap = st.model_s
        if hasattr(wrap, "flat_buffer_broadcast"):
            wrap.flat_buffer_broadcast()                   # the wrap's per-forward collective: outside the graphs
        side = st.side_stream()
        if side is not None:
            side.wait_stream(main)                         # last step's optimizer, this step's inputs
            with torch.cuda.stream(side), ops.trace_range("moma_step/g_teacher"):
                cap.g_teacher.replay()
            with ops.trace_range("moma_step/g_student"):
                cap.g_student.replay()
            main.wait_stream(side)
            st.prefetch_queue()                            # (only with opt.prefetch_queue: on the side stream, under g_query)
        else:
            with ops.trace_range("moma_step/g_student"):
                cap.g_student.replay()
            with ops.trace_range("moma_step/g_teacher"):
                cap.g_teacher.replay()
        with ops.trace_range("moma_step/g_query"):
            cap.g_query.replay()
        fw = cap.fw
        with ops.trace_range("moma_step/K2_K3"):
            if cap.k2 is None:                             # --mem MoCoAtt: g_query holds the logits (from its own snapshot of the queue)
                st.contrast.enqueue_keys(fw["all_k"] if fw["all_k"] is not None else fw["k"])
            else:
                st.contrast.forward_fused_into(fw["f_s"], fw["k"], fw["all_k"], None if cap.qpack is None else cap.qpack.buf, cap.k2)
        with ops.trace_range("moma_step/g_bwd"):
            cap.g_bwd.replay()
        for m, dn in cap.bn_delta:
            m._nbt_pending += dn
        for p, g in cap.grads:
            p.grad = g
        self.replays += 1
        # (static outputs: the next replay overwrites them -- what is handed out is a copy)
        return cap.loss.clone(), cap.loss_kd.clone(), fw["acc"].clone()

    # ---- capture ----------------------------------------------------------------------------------------------------------
    @staticmethod
    def _record(graph, stream, fn, pool=None):
        """run fn() on `stream` under a capture into `graph` (CUDAGraph.capture_begin / capture_end: no empty_cache)"""
        with torch.cuda.stream(stream):
            if pool is None:
                graph.capture_begin(capture_error_mode="thread_local")
            else:
                graph.capture_begin(pool=pool, capture_error_mode="thread_local")
            try:
                return fn()
            finally:
                graph.capture_end()

    def _capture(self, key, images, labels):
        st = self.st
        from .graphs import replay_is_safe
        from .loops_moma import _unwrap
        if not replay_is_safe(images.device):               # (runtime hazard, ../hip_env.py: the loop stays eager)
            self.enabled = False
            return None
        dev = images.device
        trainer, contrast, kd = st.trainer, st.contrast, st.criterion_kd
        cap = _Captured()
        bn_before = [(m, m._nbt_pending) for m in self._bn]
        try:
            B = images.shape[0]
            d, K = contrast.memory.shape[1], contrast.memory.shape[0]
            cap.images, cap.labels = images.clone(), labels.clone()
            cap.perm = PermFeed(B, dev)
            cap.qpack = None
            if ops.prec_code(contrast.precision) == ops.PREC_BF16 and not st.attn_in_shuffle:
                # (--attn self_mix / self_nomix: q leaves another module -- `atts` over [q ; k] -- that does not pack it)
                cap.qpack = ops.QPack().prepare(B, d, contrast.T, dev)        # this graph's own packed-q image (None: K2 packs)
            # (--mem MoCoAtt: no one-pass K2 -- logits and CrossEntropy are part of g_query, only the enqueue is issued eagerly)
            cap.k2 = None if st.mocoatt else ops.K2Buffers(B, d, K, self._streamed_queue().dtype, contrast.precision, dev)
            if st.mocoatt:
                cap.qpack = None
            # the graphs own what they read besides parameters and buffers: weight packs of the attention modules are rebuilt
            # inside the graphs (into their pools) on every replay
            for m in kd.modules():
                if hasattr(m, "invalidate_pack"):
                    m.invalidate_pack()
            st.optimizer.zero_grad(set_to_none=True)                          # the captured backward CREATES the gradients
            if self.capture_stream is None:
                self.capture_stream = torch.cuda.Stream(device=dev)
                self.capture_stream_side = torch.cuda.Stream(device=dev)
            gc.collect()
            torch.clear_autocast_cache()
            main = torch.cuda.current_stream(dev)
            cs, cs_side = self.capture_stream, self.capture_stream_side
            cs.wait_stream(main)
            cs_side.wait_stream(main)
            trainer._perm_feed = cap.perm
            student = _unwrap(st.model_s)
            cap.g_student, cap.g_teacher = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            cap.g_query, cap.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            # (a capture executes nothing: the order of the captures only has to respect the data flow of the Python objects)
            feat_s, logit_s = self._record(cap.g_student, cs, lambda: st.student_forward(cap.images, student))
            torch.clear_autocast_cache()
            logit_t, k, all_k = self._record(cap.g_teacher, cs_side, lambda: st.teacher_side(cap.images, st.model_t))
            torch.clear_autocast_cache()
            pool = cap.g_student.pool()
            def query():
                f = st.losses_and_query(feat_s, logit_s, logit_t, k, all_k, cap.images, cap.labels, st.model_t, qpack=cap.qpack, prefetch=False)
                if st.mocoatt:
                    f["loss_kd"] = st.kd_logits_loss(f)
                return f
            fw = self._record(cap.g_query, cs, query, pool)
            del feat_s, logit_s

            def bwd():
                loss_kd = fw["loss_kd"] if st.mocoatt else ops.StaticK2Loss.apply(fw["f_s"], cap.k2.loss_rows, cap.k2.dq).mean()
                loss = st.backward_part(fw, loss_kd)
                return loss.detach(), loss_kd.detach()
            cap.loss, cap.loss_kd = self._record(cap.g_bwd, cs, bwd, pool)
            main.wait_stream(cs)
            main.wait_stream(cs_side)
            torch.clear_autocast_cache()
            cap.fw = {k_: (v.detach() if torch.is_tensor(v) else v) for k_, v in fw.items()}
            cap.teacher_out = (logit_t, k, all_k)
            params = [p for g in st.optimizer.param_groups for p in g["params"]]
            cap.grads = [(p, p.grad) for p in params]
            # the capture ran the Python side once WITHOUT executing anything: undo its host-side counting, remember the deltas
            cap.bn_delta = [(m, m._nbt_pending - before) for m, before in bn_before if m._nbt_pending != before]
            # device tables the captured K4 launches read (rebuilt -- and the old one freed -- only if a parameter moves)
            cap.keepalive = list(type(trainer)._ema_tables.values())
            self.graphs[key] = cap
            if getattr(st.opt, "rank", 0) == 0:
                mid = "losses + query + cross-attention logits | K3 eager" if st.mocoatt else "losses + query | K2 / K3 eager"
                print(f"[moma] step captured into HIP graphs (batch {B}, variant {len(self.graphs)}): student forward || teacher "
                      f"side | {mid} | backward")
            return cap
        except Exception as e:                                 # pragma: no cover - depends on the runtime
            print(f"[moma] HIP-graph capture of the training step failed ({type(e).__name__}: {e}); staying eager")
            self.enabled = False
            st.optimizer.zero_grad(set_to_none=True)
            return None
        finally:
            trainer._perm_feed = None
            for m, before in bn_before:
                m._nbt_pending = before
