"""Does a HIP graph replay run parallel branches (fork/join captured from two streams) concurrently?"""
import torch, time


def main():
    dev = torch.device('cuda:0')
    a = torch.zeros(1024, device=dev); b = torch.zeros(1024, device=dev)
    big = torch.randn(8192, 8192, device=dev)
    _ = big @ big; torch.cuda.synchronize()
    def chain(t, n):
        for _ in range(n): t.add_(1.0)
    def build(two_streams, n=300, heavy=False):
        g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            chain(a, 3); chain(b, 3); big @ big
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                if two_streams:
                    side.wait_stream(s)
                    with torch.cuda.stream(side): chain(b, n)
                    (big @ big) if heavy else chain(a, n)
                    s.wait_stream(side)
                else:
                    chain(b, n); (big @ big) if heavy else chain(a, n)
        return g
    def timeit(g, reps=20):
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): g.replay()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    for heavy in (False, True):
        g1 = build(False, heavy=heavy); g2 = build(True, heavy=heavy)
        print(f"heavy={heavy}: one stream {timeit(g1):.3f} ms   fork/join {timeit(g2):.3f} ms")
    # eager streams
    side = torch.cuda.Stream()
    def eager(two):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            if two:
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side): chain(b, 300)
                big @ big
                torch.cuda.current_stream().wait_stream(side)
            else:
                chain(b, 300); big @ big
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / 5 * 1e3
    print(f"eager heavy: one stream {eager(False):.3f} ms   two streams {eager(True):.3f} ms")


if __name__ == "__main__":
    main()
