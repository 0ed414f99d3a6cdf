"""Does hipGraph replay shorten the back-to-back cost of small dependent kernels on this chip?  python tools/graph_probe.py
Chain of N tiny dependent kernels (x += 1 on 4 KiB), launched on a stream vs replayed from a captured graph."""
import torch, time
dev = "cuda"; x = torch.zeros(1024, device=dev); N = 400
def chain():
    for _ in range(N): x.add_(1.0)
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3 / N, (time.perf_counter() - t0) / reps * 1e6 / N
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    chain(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side): chain()
torch.cuda.synchronize()
print("stream launches: %.2f us per kernel on the GPU timeline (host wall %.2f us per kernel)" % timeit(chain))
print("graph replay:    %.2f us per kernel on the GPU timeline (host wall %.2f us per kernel)" % timeit(g.replay))
y = torch.zeros(1 << 22, device=dev)
def chain2():
    for _ in range(N): y.add_(1.0)
with torch.cuda.stream(side):
    chain2(); torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side): chain2()
torch.cuda.synchronize()
print("16 MiB kernels, stream: %.2f us (host %.2f)" % timeit(chain2))
print("16 MiB kernels, graph:  %.2f us (host %.2f)" % timeit(g2.replay))
# the same chain QUEUED behind a long kernel (the host is ahead, as in the GPU-bound parts of an update)
big = torch.zeros(1 << 28, device=dev)
def queued(fn):
    res = []
    for _ in range(5):
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        for _ in range(6): big.add_(1.0)            # ~2.5 ms of GPU work: the chain below is fully enqueued before it starts
        s.record(); fn(); e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) * 1e3 / N)
    return min(res)
print("queued behind a long kernel, stream launches: %.2f us per tiny kernel" % queued(chain))
print("queued behind a long kernel, graph replay:    %.2f us per tiny kernel" % queued(g.replay))
