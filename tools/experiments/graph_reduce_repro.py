"""Minimal PyTorch-only probe (no codenet_amd code) for the cause of the GraphedTrainStep wrong-loss report: inside a
captured HIP graph, several multi-block ATen reductions back to back.  Each `mean()` of a large tensor takes ATen's
global-reduce path: a staging buffer + a SEMAPHORE word from the caching allocator, `hipMemsetAsync(semaphore, 0)` (a
MEMSET NODE under capture), then the reduce kernel whose last-arriving block writes the result and resets the
semaphore.  The allocations are freed when the launch returns, so the NEXT reduction gets the same semaphore address:
its memset node must not run before the previous reduce kernel has finished.  If it does, the previous kernel's
arrival count is wiped, no block sees itself as the last one, and that reduction's output is never written -- the
static output tensor keeps the value of the previous replay.
    python tools/experiments/graph_reduce_repro.py [replays]"""
import sys

import torch

n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
xs = [torch.randn(8, 20, 64, 64, generator=g).to(dev), torch.randn(8, 2, 64, 64, generator=g).to(dev),
      torch.randn(8, 2, 64, 64, generator=g).to(dev)]
scale = torch.ones(1, device=dev)


def work():
    terms = [(x * scale).square().mean() for x in xs]
    return terms, terms[0] + terms[1] + terms[2]


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        work()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    terms, loss = work()
# other work in the process between replays
net = torch.nn.Sequential(torch.nn.Conv2d(3, 24, 3, 2, 1), torch.nn.BatchNorm2d(24), torch.nn.ReLU(),
                          torch.nn.Conv2d(24, 58, 3, 2, 1)).to(dev)
xt = torch.randn(8, 3, 256, 256, device=dev)
bad = stale = 0
first = None
for it in range(n_rep):
    scale.fill_(1.0 + 0.001 * (it % 977))           # every replay has its own expected values
    if it % 3 == 1:
        net(xt).square().mean().backward()
    elif it % 3 == 2:
        torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    want = [(x * scale).double().square().mean().item() for x in xs]
    got = [t.item() for t in terms]
    ok = all(abs(a - b) <= 1e-5 * abs(b) for a, b in zip(got, want)) and abs(loss.item() - sum(want)) <= 1e-5 * sum(want)
    if not ok:
        bad += 1
        if first is None:
            first = (it, got, want, loss.item())
print("GRAPH_REDUCE replays %d wrong %d first %r" % (n_rep, bad, first))
