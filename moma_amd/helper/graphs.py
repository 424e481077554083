"""HIP-graph replay of a no-grad module forward (the two teacher forwards of a MoMA step).

The train step launches ~2400 kernels; on the host that is ~35 ms of Python / ATen / MIOpen dispatch per step, two
thirds of the forward part of it in the two teacher passes (reference helper/loops_moma.py:270-272 and
learning/contrast_trainer.py:118-121), which are inference-only: fixed shapes, no autograd, weights updated in place by the
EMA kernel.  `GraphedInference` runs such a forward eagerly a few times (MIOpen compiles / selects its kernels), then
captures it once per (input shape, dtype, autocast state, train/eval flags) into a `torch.cuda.CUDAGraph` (= hipGraph) and
replays it: one launch instead of ~700.  Parameters and buffers are read and written through their own storage, so EMA
updates and BatchNorm running statistics behave exactly as in eager mode; dropout-like ops draw from the graph-safe
Philox stream.  Anything unusual (CPU tensors, grad mode on, a capture error) falls back to the eager call.
"""
import os

import torch

from ..hip_env import SWITCH as _SWITCH

_REPLAY_SAFE = {}


def replay_is_safe(device) -> bool:
    """Self-test of the HIP runtime's graph replay on `device`, once per process: a captured chain of ATen column sums (each a
    memset node + a reduce kernel) is replayed three times with eager kernels of the same kinds in between and compared with the
    eager result.  ROCm 7.2 with graph packet capture on fails it from the second replay on (../hip_env.py, which switches the
    packet capture off at import; this check covers a process whose HIP runtime was initialised before that).  Every capture of
    this package asks here first: no graph is recorded on a runtime that fails."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    if key in _REPLAY_SAFE:
        return _REPLAY_SAFE[key]
    ok = True
    try:
        with torch.no_grad():
            gen = torch.Generator(device="cpu").manual_seed(7)
            x = torch.randn(4096, 1024, generator=gen).to(device=device, dtype=torch.bfloat16)
            ref = x.float().sum(0)
            main = torch.cuda.current_stream(device)
            s = torch.cuda.Stream(device=device)
            s.wait_stream(main)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(s):
                x.sum(0)
                g.capture_begin(capture_error_mode="thread_local")
                try:
                    outs = [x.sum(0) for _ in range(24)]
                finally:
                    g.capture_end()
            main.wait_stream(s)
            worst = torch.zeros((), device=device)
            for _ in range(3):
                g.replay()
                got = torch.stack(outs).float()
                worst = torch.maximum(worst, torch.nan_to_num((got - ref).abs(), nan=1e30, posinf=1e30).max())
                z = torch.zeros(1 << 16, device=device)
                for _ in range(8):                        # eager work: column sums (memset + reduce), fills, elementwise
                    z = z * 2 + x[:2048].sum(0).float().mean()
            ok = float(worst) < 8.0                        # (bf16 output of a 4096-term sum: rounding stays below 2)
            del g, outs
    except Exception as e:                                 # pragma: no cover - depends on the runtime
        print(f"[moma] HIP-graph replay self-test could not run ({type(e).__name__}: {e}); graphs off")
        ok = False
    if not ok:
        print("[moma] this HIP runtime replays captured memset nodes incorrectly after eager work (ROCm graph packet capture; "
              f"set {_SWITCH}=0 before the first HIP call -- importing moma_amd first does it): HIP graphs are OFF, "
              "the step runs eagerly")
    _REPLAY_SAFE[key] = ok
    return ok


class GraphedInference:
    """Contract of the returned values: `model(x, is_feat=True)` gives `([pooled_feature], logits)` -- ONLY the last
    (pooled) feature, cloned, and the cloned logits; the intermediate feature maps live in the graph's static memory,
    are overwritten by the next replay and are therefore not handed out (the MoMA loop reads `feat[-1]` and the logits
    only; a caller that needs the whole list passes `full_feats=True` and gets an eager forward).  At most `max_graphs`
    (shape, autocast, train/eval, weight addresses, stream) variants are held -- each owns a private activation pool; at capacity
    the one unused longest makes room for a new one."""

    def __init__(self, module, warmup: int = 3, max_graphs: int = 4):
        self.module = module
        self.warmup = warmup
        self.max_graphs = max_graphs
        self.enabled = os.environ.get("MOMA_GRAPH_TEACHER", "1") == "1"
        self._seen = {}
        self._graphs = {}
        self._used, self._tick = {}, 0
        self._mods = list(module.modules())
        self._state = list(module.parameters()) + list(module.buffers())     # the graph reads these in place: their addresses are part of the key

    # the wrapped module stays reachable for everything that is not a forward call
    def __getattr__(self, name):
        return getattr(self.__dict__["module"], name)

    def _key(self, x, is_feat):
        ac = (torch.is_autocast_enabled(), torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else None)
        modes = hash(tuple(m.training for m in self._mods))
        # the stream is part of the key: a graph is only ever replayed on the stream that captured it (library workspaces --
        # hipBLASLt, MIOpen -- belong to the capturing stream; see prime())
        stream = torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0
        return (tuple(x.shape), x.dtype, x.device, x.is_contiguous(), bool(is_feat), ac, modes, hash(tuple(t.data_ptr() for t in self._state)), stream)

    def __call__(self, x, is_feat=False, full_feats=False):
        if full_feats:
            return self.module(x, is_feat=is_feat)
        if not (self.enabled and x.is_cuda and not torch.is_grad_enabled()):
            return self._trim(self.module(x, is_feat=is_feat), is_feat)       # one contract whatever path serves the call
        key = self._key(x, is_feat)
        entry = self._graphs.get(key)
        if entry is None:
            n = self._seen.get(key, 0) + 1
            if len(self._seen) < 64 or key in self._seen:
                self._seen[key] = n
            if n <= self.warmup:
                return self._trim(self.module(x, is_feat=is_feat), is_feat)
            if len(self._graphs) >= self.max_graphs:
                # at capacity: the variant unused longest makes room (one keyed on weights that have moved since can never match
                # again); its graph is idle -- this variant has just been served `warmup` eager calls -- and the synchronize makes
                # that a fact before it is destroyed
                torch.cuda.synchronize(x.device)
                victim = min(self._graphs, key=lambda k_: self._used.get(k_, 0))
                del self._graphs[victim]
                self._used.pop(victim, None)
                self._seen.pop(victim, None)
            entry = self._capture(key, x, is_feat)
            if entry is None:
                return self._trim(self.module(x, is_feat=is_feat), is_feat)
        self._tick += 1
        self._used[key] = self._tick
        graph, static_x, out, bn_train = entry
        static_x.copy_(x)
        graph.replay()
        for m in bn_train:                                    # host-side batch counters (see backbones' BatchNorm2d)
            m._nbt_pending += 1
        return self._detach_outputs(out, is_feat)

    def prime(self, x, is_feat=True):
        """Serve this (shape, autocast, train/eval) variant until it is captured, so that the next call is a replay.  Meant for
        variants without side effects -- the eval-mode forward that opens every epoch (reference helper/loops_moma.py:227,
        270-272) -- before a timed region; returns True when the variant is (now) graphed.
        Call it ON THE STREAM THE REPLAYS WILL RUN ON (the loop's side stream under overlap_teacher): library workspaces
        (hipBLASLt / MIOpen) belong to the capturing stream, and a replay that runs beside other work of that stream races on
        them (seen in round 3: a variant captured on the main stream and replayed on the side stream next to the student forward
        returned garbage logits -- possibly also an instance of the runtime hazard of ../hip_env.py, found a round later; the
        per-stream rule stays either way)."""
        if not (self.enabled and x.is_cuda):
            return False
        if any(m.training for m in self._mods if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)):
            # priming serves warmup + 1 real forwards: in training mode each of them moves the running statistics
            print("[moma] GraphedInference.prime: a BatchNorm layer is in training mode -- priming would advance its running "
                  "statistics; variant left to the regular warm-up")
            return False
        with torch.no_grad():
            for _ in range(self.warmup + 1):
                if self._key(x, is_feat) in self._graphs:
                    break
                # every call must see an EMPTY autocast weight-cast cache, as it does in the loop (one autocast context per call):
                # a cast cached by an earlier eager call would be reused during the capture -- the graph would then hold no cast
                # kernel and read a cached low-precision weight that is freed when the caller's autocast context exits
                torch.clear_autocast_cache()
                self(x, is_feat=is_feat)
            torch.clear_autocast_cache()
            return self._key(x, is_feat) in self._graphs

    def _capture(self, key, x, is_feat):
        if not replay_is_safe(x.device):
            self.enabled = False
            return None
        try:
            torch.clear_autocast_cache()        # the casts of this forward belong INSIDE the graph (see prime())
            static_x = x.clone()
            graph = torch.cuda.CUDAGraph()
            pending = [(m, m._nbt_pending) for m in self._mods if hasattr(m, "_nbt_pending")]
            # thread_local: other threads (RCCL's watchdog polls events) may keep calling the runtime during the capture
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out = self.module(static_x, is_feat=is_feat)
            # the capture ran the Python side once without executing kernels: undo its host-side counting
            bn_train = [m for m, before in pending if m._nbt_pending != before]
            for m, before in pending:
                m._nbt_pending = before
            torch.clear_autocast_cache()        # cached casts made during the capture live in the graph's pool: drop the references
            entry = (graph, static_x, out, bn_train)
            self._graphs[key] = entry
            return entry
        except Exception as e:                                 # pragma: no cover - depends on the runtime
            print(f"[moma] HIP-graph capture of the teacher forward failed ({type(e).__name__}: {e}); staying eager")
            self.enabled = False
            return None

    @staticmethod
    def _detach_outputs(out, is_feat):
        """Static graph outputs are overwritten by the next replay: what is handed out (pooled feature, logits) is cloned."""
        if is_feat:
            feats, logits = out
            return [feats[-1].clone()], logits.clone()
        return out.clone()

    @staticmethod
    def _trim(out, is_feat):
        """Eager calls return the same shape of result as replays (one contract whatever path served the call)."""
        if is_feat:
            feats, logits = out
            return [feats[-1]], logits
        return out
