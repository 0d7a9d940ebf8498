"""Do small kernels on a second HIP stream run beside large ones?  A = 20 large 3x3 convs (main stream), B = 600 tiny
launches (side stream).  Eager two-stream issue, two HIP graphs replayed on two streams, and one HIP graph holding both
as parallel branches (how the training step holds its frozen-OCR branch).  GPU box: python tools/exp_stream_concurrency.py"""
import sys, time; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
dev = torch.device('cuda:0')
x = torch.randn(16, 128, 64, 256, device=dev)
wp = ops.pack_filter(torch.randn(3, 3, 128, 128, device=dev), False, False)
xs = torch.randn(16, 256, 2, 25, device=dev)
ws = ops.pack_filter(torch.randn(3, 3, 256, 256, device=dev), False, False)

def A():
    for _ in range(20):
        ops.conv2d_raw(x, wp, 128, 3, 3, (64, 256), (1, 1), (1, 1))

def B():
    for _ in range(200):
        ops.conv2d_raw(xs, ws, 256, 3, 3, (2, 25), (1, 1), (1, 1))  # split-K: conv + slab epilogue = 2 launches

def wall(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

PRIO = int(sys.argv[1]) if len(sys.argv) > 1 else 0
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(priority=PRIO)
print("side stream priority", PRIO, "(range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?", ")")
def graph_of(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            fn()
    return g
gA, gB = graph_of(A, s1), graph_of(B, s2)

def both_branches():
    side = s2
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        B()
    A()
    torch.cuda.current_stream().wait_stream(side)
gAB = graph_of(both_branches, s1)

def two_graphs():
    with torch.cuda.stream(s1): gA.replay()
    with torch.cuda.stream(s2): gB.replay()

def serial_graphs():
    with torch.cuda.stream(s1):
        gA.replay(); gB.replay()

print(f"graph A alone            {wall(gA.replay):7.2f} ms")
print(f"graph B alone            {wall(gB.replay):7.2f} ms")
print(f"A then B, one stream     {wall(serial_graphs):7.2f} ms")
print(f"A || B, two graphs on two streams {wall(two_graphs):7.2f} ms")
print(f"A || B, ONE graph with two branches {wall(gAB.replay):7.2f} ms")
