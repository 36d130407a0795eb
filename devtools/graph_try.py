"""Does the inference rollout capture into a HIP graph (torch.cuda.graph) and replay bit-exactly?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pl_convlstm_gan_amd import Generator  # noqa: E402

dev = torch.device("cuda:0")
for (B, T, H, W) in [(2, 10, 64, 64), (8, 10, 128, 128), (4, 10, 256, 256)]:
    torch.manual_seed(0)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=1).to(dev)
    rain = torch.rand(B, T, 1, H, W, device=dev)
    dem = torch.rand(B, 1, H, W, device=dev)
    lu = torch.rand(B, 5, H, W, device=dev)
    with torch.no_grad():
        for _ in range(3):
            ref = m(rain, dem, lu).clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        m(rain, dem, lu)
    torch.cuda.current_stream().wait_stream(s)
    with torch.no_grad(), torch.cuda.graph(g):
        out = m(rain, dem, lu)
    g.replay()
    torch.cuda.synchronize()
    print((B, T, H, W), "graph == eager:", torch.equal(out, ref))

    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def eager():
        with torch.no_grad():
            m(rain, dem, lu)
    print("   eager %.3f ms   graph %.3f ms" % (timeit(eager), timeit(g.replay)))
