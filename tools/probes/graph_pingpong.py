"""Does re-launching ONE graph exec back to back leave the GPU idle between replays?  Times K steps replayed from
one exec, from two execs launched alternately, and from one exec that holds two steps.  GPU only."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from ilps_amd import ops
from ilps_amd.smpl_model import synthetic_smpl_model

dev = torch.device("cuda", 0)
B, W, K = 128, 48, 200
consts = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
pt = ops.get_part_table(1, dev, consts.V)
x = torch.tensor(bench.make_x(B, W, 1000), device=dev)
dseg = torch.randn(B, W, W, 32, device=dev)


def step():
    xg = x.detach().requires_grad_(True)
    verts, proj, mask, seg, silh, jt, _ls = ops.DecoderFn.apply(xg, consts, 4, W, 1, pt, 64, True, False, 1)
    seg.backward(dseg)
    return xg.grad


def capture(nsteps):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(nsteps):
            out = step()
    return g, out


def timeit(fn, launches, steps_per_launch):
    for _ in range(10):
        fn(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(launches):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (launches * steps_per_launch) * 1e6


g1, _ = capture(1)
g2, _ = capture(1)
g4, _ = capture(2)
gs = [g1, g2]
print("one exec               : %.1f us/step" % timeit(lambda i: g1.replay(), K, 1))
print("two execs, alternating : %.1f us/step" % timeit(lambda i: gs[i & 1].replay(), K, 1))
print("one exec of two steps  : %.1f us/step" % timeit(lambda i: g4.replay(), K // 2, 2))
print("one exec (again)       : %.1f us/step" % timeit(lambda i: g1.replay(), K, 1))
