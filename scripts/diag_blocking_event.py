"""Does a blocking-sync HIP event let the waiting host thread sleep?  wall vs CPU time of Event.synchronize() behind ~0.3 s of GPU work."""
import time, torch
x = torch.randn(8192, 8192, device="cuda")
for blocking in (False, True):
    torch.cuda.synchronize()
    for _ in range(40):
        y = x @ x
    ev = torch.cuda.Event(blocking=blocking)
    ev.record()
    w0, c0 = time.perf_counter(), time.thread_time()
    ev.synchronize()
    w1, c1 = time.perf_counter(), time.thread_time()
    print(f"Event(blocking={blocking}).synchronize(): wall {1e3 * (w1 - w0):.1f} ms, thread CPU {1e3 * (c1 - c0):.1f} ms")
torch.cuda.synchronize()
for _ in range(40):
    y = x @ x
w0, c0 = time.perf_counter(), time.thread_time()
torch.cuda.synchronize()
w1, c1 = time.perf_counter(), time.thread_time()
print(f"torch.cuda.synchronize(): wall {1e3 * (w1 - w0):.1f} ms, thread CPU {1e3 * (c1 - c0):.1f} ms")
