"""Process environment of the HIP runtime that the replayed step depends on.

ROCm 7.2's graph executor pre-records the AQL packets of a graph at instantiation ("graph packet capture", on by default).  With
it, a captured `hipMemsetAsync` node -- ATen's column reductions zero their semaphores that way, e.g. every bias gradient of a wide
`nn.Linear` -- stops doing its work once EAGER kernels have run on the device between two replays: the reduce kernel behind it then
merges partial sums that were never written (seen round 4: garbage bias gradients in the ViT-S student of BASELINE configs[2] a few
steps after the first print of the loop; `scripts/diag_graph_memset.py` is the 40-line reproduction, clean with the switch below
and with no graph at all).  `DEBUG_CLR_GRAPH_PACKET_CAPTURE=0` makes the runtime dispatch graph nodes through its regular path:
measured cost 6.6 ms instead of 1.5 ms of host time per replayed step (against ~36 ms for the eager step), GPU time unchanged.

`configure()` must run before the process's first HIP call (the runtime reads its switches once): importing `moma_amd` does it.
Nothing is taken on trust -- `helper/graphs.py:replay_is_safe()` runs the reproduction on the live runtime before any graph of this
package is captured, and the loop stays eager (correct, slower) when it fails.
"""
import os

SWITCH = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def configure() -> bool:
    """-> True when the switch is (now) set to 0 in this process's environment.  A caller's own setting is left alone.
    Also defaults HSA_ENABLE_IPC_MODE_LEGACY to 0: the host driver of this platform supports dmabuf IPC only, and without it RCCL's
    intra-node transport (and any sharing of device tensors across processes) fails with `hipIpcGetMemHandle: invalid argument` --
    every rank the CLI spawns (train_student_moma.py --multiprocessing-distributed, bench.py --gpus N) inherits it from here."""
    os.environ.setdefault(SWITCH, "0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # room for the kernel arguments of the queued steps (up to three, ~2400 launches each: helper/step_graph.py:_throttle): with the runtime's default pool the host hits the
    # end of it inside hipGraphLaunch every other step and spins there (bench: host issue alternating 25 / 47 ms per 40 ms step)
    os.environ.setdefault("HSA_KERNARG_POOL_SIZE", str(64 << 20))
    return os.environ[SWITCH] == "0"
