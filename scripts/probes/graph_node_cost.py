"""device time per node of a replayed hipGraph made of N dependent tiny kernels, against the same kernels enqueued on a
stream; run under different DEBUG_HIP_GRAPH_* / DEBUG_CLR_GRAPH_PACKET_CAPTURE settings (see profiles/r03_graph_node_cost.txt)"""
import os, sys, time
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
x = torch.zeros(256, device="cuda")
def body():
    for _ in range(n):
        x.add_(1.0)
for _ in range(3):
    body()
torch.cuda.synchronize()
t0 = time.perf_counter(); body(); th = time.perf_counter() - t0; torch.cuda.synchronize(); te = time.perf_counter() - t0
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    body()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
reps = 10
t0 = time.perf_counter()
for _ in range(reps):
    g.replay()
thg = time.perf_counter() - t0
torch.cuda.synchronize()
tg = time.perf_counter() - t0
print("env", {k: v for k, v in os.environ.items() if "GRAPH" in k}, "nodes", n,
      "eager: host %.2f us/launch, total %.2f us/kernel | graph: host %.2f us/node, total %.2f us/node" %
      (1e6 * th / n, 1e6 * te / n, 1e6 * thg / n / reps, 1e6 * tg / n / reps))
