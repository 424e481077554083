"""GPU-box experiment: do EXTERNAL event-record nodes inside a captured HIP graph give usable timestamps on replay?

    python scripts/diag_graph_events.py

Captures [ev0 (external record) -> a matmul -> ev1 (external record)] on a side stream with torch.cuda.graph, replays it a few
times and reads hipEventElapsedTime(ev0, ev1) after each replay; the same matmul timed eagerly with ordinary events is printed
beside it.  Used to decide how bench.py times K2 when the step runs from a graph (helper/step_graph.py)."""
import ctypes as C
import sys

import torch

hip = C.CDLL("libamdhip64.so")
hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
hip.hipEventRecordWithFlags.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
hip.hipEventSynchronize.argtypes = [C.c_void_p]
hip.hipEventQuery.argtypes = [C.c_void_p]
EXTERNAL = 1


def ev():
    e = C.c_void_p()
    assert hip.hipEventCreate(C.byref(e)) == 0
    return e


def main():
    dev = torch.device("cuda", 0)
    a = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    b = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); c = a @ b; e1.record(); torch.cuda.synchronize()
    print("eager matmul ms:", e0.elapsed_time(e1))

    g0, g1 = ev(), ev()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        raw = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        pre = a * 2                                   # something in front
        rc0 = hip.hipEventRecordWithFlags(g0, raw, EXTERNAL)
        if rc0 != 0:
            print("external record rc", rc0, "-> clearing, trying plain hipEventRecord"); hip.hipGetLastError()
            rc0 = hip.hipEventRecord(g0, raw)
            print("plain record rc", rc0); hip.hipGetLastError()
        c = a @ b
        rc1 = hip.hipEventRecordWithFlags(g1, raw, EXTERNAL)
        if rc1 != 0:
            hip.hipGetLastError()
            rc1 = hip.hipEventRecord(g1, raw); hip.hipGetLastError()
        post = c.float().sum()
    print("record rc inside capture:", rc0, rc1)
    for i in range(4):
        graph.replay()
        rc = hip.hipEventSynchronize(g1)
        ms = C.c_float(-1.0)
        rc2 = hip.hipEventElapsedTime(C.byref(ms), g0, g1)
        print(f"replay {i}: sync rc {rc}, elapsed rc {rc2}, ms {ms.value:.4f}")
    torch.cuda.synchronize()
    print("sum", float(post))
    # event sync WHILE later graph work is still queued: the host must come back before the tail finishes
    big = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    graph2 = torch.cuda.CUDAGraph()
    h0, h1 = ev(), ev()
    with torch.cuda.graph(graph2):
        raw = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        if hip.hipEventRecordWithFlags(h0, raw, EXTERNAL) != 0:
            hip.hipGetLastError(); hip.hipEventRecord(h0, raw); hip.hipGetLastError()
        c = a @ b
        if hip.hipEventRecordWithFlags(h1, raw, EXTERNAL) != 0:
            hip.hipGetLastError(); hip.hipEventRecord(h1, raw); hip.hipGetLastError()
        x = big
        for _ in range(20):
            x = (x @ big) * 1e-3
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter(); graph2.replay(); hip.hipEventSynchronize(h1); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ms = C.c_float(-1.0); hip.hipEventElapsedTime(C.byref(ms), h0, h1)
    print(f"graph2: host back after {1e3 * (t1 - t0):.3f} ms (whole graph {1e3 * (t2 - t0):.3f} ms), kernel ms {ms.value:.4f}")


if __name__ == "__main__":
    sys.exit(main())
