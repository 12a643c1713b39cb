#!/usr/bin/env python3
"""Probe: training-mode BatchNorm2d fwd+bwd on ENet's activation sizes, MIOpen vs torch's native kernels."""
import time, torch, torch.nn.functional as F
dev = torch.device("cuda:0")
def run(shape, native):
    x = torch.randn(*shape, device=dev, requires_grad=True)
    bn = torch.nn.BatchNorm2d(shape[1], eps=1e-3).to(dev).train()
    g = torch.randn(*shape, device=dev)
    def step():
        with torch.backends.cudnn.flags(enabled=not native):
            y = bn(x)
        y.backward(g)
        x.grad = None
    for _ in range(3): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 10 * 1e3
for shape in [(256, 16, 128, 128), (256, 64, 64, 64), (256, 16, 64, 64), (256, 128, 32, 32), (256, 32, 32, 32), (256, 256, 32, 32)]:
    mb = 4 * shape[0] * shape[1] * shape[2] * shape[3] / 1e6
    a, b = run(shape, False), run(shape, True)
    print("%-22s %6.0f MB  miopen %.3f ms (%.2f TB/s of 8 passes)  native %.3f ms (%.2f TB/s)" % (shape, mb, a, 8 * mb / a / 1e6, b, 8 * mb / b / 1e6))
