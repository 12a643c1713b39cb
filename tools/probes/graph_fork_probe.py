#!/usr/bin/env python3
"""Probe: do forked branches of a captured HIP graph overlap?  A chain of N small latency-bound kernels
(pose_fwd on 128 meshes: 32 workgroups) serial on one stream vs split over two forked streams, replayed from a graph."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import ilps_amd
from ilps_amd import ops
from ilps_amd.smpl_model import synthetic_smpl_model
sys.argv = ["x"]
import bench
dev = torch.device("cuda:0")
consts = ops.SMPLConstants.from_model(synthetic_smpl_model(1234), dev)
x = torch.tensor(bench.make_x(128, 48, 1), device=dev)
outs = [(None, torch.empty(128, 24, 9, device=dev), torch.empty(128, 24, 3, device=dev), torch.empty(128, 24, 12, device=dev),
         torch.empty(128, 24, 3, device=dev)) for _ in range(2)]
side = torch.cuda.Stream()

def serial(n):
    for i in range(n):
        ops._pose_fwd(x, 4, consts, out=outs[0])

def forked(n):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for i in range(n // 2):
            ops._pose_fwd(x, 4, consts, out=outs[1])
    for i in range(n // 2):
        ops._pose_fwd(x, 4, consts, out=outs[0])
    cur.wait_stream(side)

def one_side(n):
    """main: n-1 kernels; side: ONE kernel forked at the start, joined at the end"""
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        ops._pose_fwd(x, 4, consts, out=outs[1])
    for i in range(n - 1):
        ops._pose_fwd(x, 4, consts, out=outs[0])
    cur.wait_stream(side)

def timed(fn, n):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(n)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn(n)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 200 * 1e6

for n in (2, 8):
    print("n=%d kernels: serial %.1f us, two forked branches %.1f us, one side kernel %.1f us"
          % (n, timed(serial, n), timed(forked, n), timed(one_side, n)))
