import os, sys, time, threading
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from reart_amd.utils.lap import cdist, linear_sum_assignment_batch
dev = torch.device("cuda:0")
g = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/structure.npz"))
pred, pcs = torch.from_numpy(g["pred"]).to(dev), torch.from_numpy(g["pc_list"]).to(dev)
cost = cdist(pred, pcs)
costs = [cost.clone() for _ in range(3)]
streams = [torch.cuda.Stream() for _ in range(3)]
linear_sum_assignment_batch(cost)
def run(k):
    with torch.cuda.stream(streams[k]):
        linear_sum_assignment_batch(costs[k])
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(3): run(k)
torch.cuda.synchronize(); print("sequential 3 x (9 x 4096^2): %.0f ms" % ((time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(k,)) for k in range(3)]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize(); print("3 threads / 3 streams: %.0f ms" % ((time.perf_counter() - t0) * 1e3))
big = torch.cat(costs)
torch.cuda.synchronize(); t0 = time.perf_counter()
linear_sum_assignment_batch(big)
torch.cuda.synchronize(); print("one batch of 27: %.0f ms" % ((time.perf_counter() - t0) * 1e3))
