"""Can the frozen-OCR branch be hidden by giving it a few CUs of its own?  Streams created with
hipExtStreamCreateWithCUMask (disjoint masks), graph A = 20 large convs on the big partition, graph B = 400 tiny launches
on the small one.  GPU box: python tools/exp_cu_mask.py [side CUs]"""
import ctypes as C, sys, time; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
dev = torch.device('cuda:0')
torch.cuda.init(); torch.zeros(1, device=dev)
hip = C.CDLL("libamdhip64.so")
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 32
NCU = torch.cuda.get_device_properties(0).multi_processor_count

def masked_stream(bits):
    words = (C.c_uint32 * ((NCU + 31) // 32))()
    for b in bits: words[b // 32] |= (1 << (b % 32))
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), len(words), words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)

# every 256/NS-th CU for the side partition (spread over the XCDs whatever the bit order is), the rest for the main one
step = NCU // NS
side_bits = [i for i in range(NCU) if i % step == 0][:NS]
main_bits = [i for i in range(NCU) if i not in set(side_bits)]
s_main, s_side = masked_stream(main_bits), masked_stream(side_bits)
s_full = torch.cuda.Stream()
print(f"{NCU} CUs: side partition {len(side_bits)}, main partition {len(main_bits)}")

x = torch.randn(16, 128, 64, 256, device=dev)
wp = ops.pack_filter(torch.randn(3, 3, 128, 128, device=dev), False, False)
xs = torch.randn(16, 256, 2, 25, device=dev)
ws = ops.pack_filter(torch.randn(3, 3, 256, 256, device=dev), False, False)
def A():
    for _ in range(20): ops.conv2d_raw(x, wp, 128, 3, 3, (64, 256), (1, 1), (1, 1))
def B():
    for _ in range(200): ops.conv2d_raw(xs, ws, 256, 3, 3, (2, 25), (1, 1), (1, 1))
def graph_of(fn, stream):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(stream):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=stream):
            fn()
    return g
def wall(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
gA, gB = graph_of(A, s_full), graph_of(B, s_full)
def on(stream, g):
    def f():
        with torch.cuda.stream(stream): g.replay()
    return f
def both():
    with torch.cuda.stream(s_main): gA.replay()
    with torch.cuda.stream(s_side): gB.replay()
print(f"A on all CUs             {wall(on(s_full, gA)):7.2f} ms")
print(f"B on all CUs             {wall(on(s_full, gB)):7.2f} ms")
print(f"A on the main partition  {wall(on(s_main, gA)):7.2f} ms")
print(f"B on the side partition  {wall(on(s_side, gB)):7.2f} ms")
print(f"A (main) || B (side)     {wall(both):7.2f} ms")
