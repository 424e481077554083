"""Counting the GPUs of this process tree WITHOUT touching them.

The parent of the ranks -- `train_student_moma.py --multiprocessing-distributed` in front of `mp.spawn` (the reference's own launch
mode, train_student_moma.py:207-224) and `bench.py --gpus N` -- must never open the device: on this platform a process that has
initialised HIP and then starts other programs is refused or takes the node down, and every process on the card counts against the
box's limit.  `torch.cuda.device_count()` is not that: where amdsmi does not initialise it falls back to `hipGetDeviceCount`.
"""
import os


def visible_gpu_count():
    """GPUs this process tree may use, counted WITHOUT any HIP / HSA call (the parent of the ranks must never open the device:
    torch.cuda.device_count() falls back to hipGetDeviceCount when amdsmi is unusable, as it is on this pool): KFD's topology in
    sysfs -- a node with simd_count > 0 is a GPU --, narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES
    when set.  None = unknown (no KFD sysfs): the ranks report a missing device themselves."""
    import glob
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    have = 0
    for f in nodes:
        try:
            props = dict(l.split()[:2] for l in open(f).read().splitlines() if len(l.split()) >= 2)
        except OSError:
            continue                                  # (a node this cgroup may not read is not ours)
        if int(props.get("simd_count", "0")) > 0:
            # (a container sees the host's whole topology; a GPU is ours when its render node is there and may be opened)
            minor = props.get("drm_render_minor")
            if minor is None or os.access(f"/dev/dri/renderD{minor}", os.R_OK | os.W_OK):
                have += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            have = min(have, len([x for x in v.split(",") if x.strip() != ""]))
    return have
