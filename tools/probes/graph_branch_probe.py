"""Do two branches of ONE captured hipGraph overlap on this stack?  Branch A: an HBM-bound stream (copy of `--mb` MB); branch B: a chain of K small dependent
kernels (each ~5 us).  Reports: A alone, B alone, A then B in one stream (captured), A || B as two branches of one captured graph, and the same pair
launched eagerly on two streams.  (The material step's backward has such a pair: the albedo Adam against the roughness gradient chain.)"""
import argparse
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mb", type=int, default=1200)
    ap.add_argument("--k", type=int, default=8)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    n = a.mb * (1 << 20) // 4
    src, dst = torch.empty(n, device=dev), torch.empty(n, device=dev)
    small = [torch.rand(98304 * 3, device=dev) for _ in range(2)]

    def A():
        dst.copy_(src)

    def B():
        x = small[0]
        for _ in range(a.k):
            x = x * 1.0001 + 0.5            # dependent elementwise launches of a 1.2 MB tensor
        small[1].copy_(x)

    def timeit(f, reps=20):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    def capture(body):
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            body()                              # warm-up
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=s):
                body()
        torch.cuda.current_stream().wait_stream(s)
        return g

    side = torch.cuda.Stream()

    def forked():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            B()
        A()
        cur.wait_stream(side)

    gA, gB = capture(A), capture(B)
    gAB = capture(lambda: (A(), B()))
    gF = capture(forked)
    print("A alone (graph)            %7.1f us" % timeit(gA.replay))
    print("B alone (graph, %d launches) %7.1f us" % (a.k * 2 + 1, timeit(gB.replay)))
    print("A then B, one stream (graph) %7.1f us" % timeit(gAB.replay))
    print("A || B, two graph branches   %7.1f us" % timeit(gF.replay))
    print("A || B, eager on two streams %7.1f us" % timeit(forked))


if __name__ == "__main__":
    main()
