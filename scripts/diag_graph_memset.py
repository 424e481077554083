"""Round-4 diagnostic (ROCm 7.2 runtime, not this library): a captured ATen column sum -- a memset node (its semaphores) followed by
the reduce kernel -- replayed after EAGER work on the same device.  With the runtime's graph packet capture on (default) later
replays can return garbage; DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 is clean.  usage: python scripts/diag_graph_memset.py [rows cols [eager-work]]   eager-work: all | memset | colsum | fill | elementwise | sort | none"""
import sys
import torch
rows, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (50432, 1152)
mode = sys.argv[3] if len(sys.argv) > 3 else "all"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
x = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
ref = x.float().sum(0)
s = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        y = x.sum(0)
    g.capture_begin()
    outs = [x.sum(0) for _ in range(24)]
    g.capture_end()
torch.cuda.current_stream().wait_stream(s)
snaps = []
for it in range(12):
    g.replay()
    snaps.append(torch.stack(outs).float())          # (device-side copy: nothing is read back before the end)
    torch.cuda.synchronize()
    # eager work between the replays: reductions with their own memsets, fills, a sort
    if mode == "all":
        z = torch.zeros(1 << 16, device=dev)
        for _ in range(8):
            z = z + torch.isfinite(x[:1024]).all().float() + x[:4096].float().sum(0).mean()
        torch.arange(4096, device=dev).flip(0).sort()
    elif mode == "memset":
        z = torch.empty(1 << 16, device=dev)
        torch.cuda.current_stream().synchronize()
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        for _ in range(8):
            hip.hipMemsetAsync(ctypes.c_void_p(z.data_ptr()), 0, ctypes.c_size_t(64), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    elif mode == "colsum":
        for _ in range(8):
            x.sum(0)
    elif mode == "fill":
        for _ in range(8):
            torch.zeros(1 << 16, device=dev)
    elif mode == "elementwise":
        z = torch.empty(1 << 16, device=dev)
        for _ in range(8):
            z = z * 2 + 1
    elif mode == "sort":
        torch.arange(4096, device=dev).flip(0).sort()
    elif mode == "d2h":
        x[0, 0].item()
    elif mode == "d2h_async":
        hbuf = torch.empty(1, dtype=torch.bfloat16).pin_memory()
        hbuf.copy_(x[0, :1], non_blocking=True)
        torch.cuda.synchronize()
bad_total = 0
for it, sn in enumerate(snaps):
    err = float((sn - ref).abs().max())
    bad = int(((sn - ref).abs().amax(1) > 8.0).sum() + (~torch.isfinite(sn)).any(1).sum())
    bad_total += bad
    print(f"replay {it}: max |err| {err:.3g}, outputs off {bad} of {len(outs)}", flush=True)
print("RESULT", "garbage seen" if bad_total else "clean")
