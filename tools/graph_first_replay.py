"""How much does the FIRST replay of a captured 20-iteration graph cost over the following ones (the driver's round-end run
times exactly one replay after 5 warm-up iterations), and does hipGraphUpload ahead of it help?
    python tools/graph_first_replay.py [upload]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
eng, seq, model = bench.build_instance(dev, 20, 4096, 10, 2, n_iter=15000)
used = eng.capture(steps_per_graph=20)
eng.step(4)
torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "upload":
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipGraphUpload.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    rc = hip.hipGraphUpload(ctypes.c_void_p(eng._graph.raw_cuda_graph_exec()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    print("hipGraphUpload rc", rc)
if len(sys.argv) > 1 and sys.argv[1] == "spin":
    # keep the GPU busy for ~100 ms right before the first replay: does the clock state matter?
    x = torch.randn(4096, 4096, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        y = x @ x
    torch.cuda.synchronize()
    del x, y
for i in range(6):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step(20)
    torch.cuda.synchronize()
    print(f"replay {i}: {(time.perf_counter() - t0) * 1e6:8.1f} us for 20 iterations")
