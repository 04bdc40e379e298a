"""Does a captured hipGraph run two independent branches concurrently?  Branch A: a chain of
kernels that does not fill the GPU for long (a 10 us splice-like copy); branch B: a long chain
(VALU-heavy elementwise work standing in for the conv backward).  Time of the graph with both
branches against B alone and A alone (HIP events around 200 replays)."""
import torch
dev = torch.device("cuda:0")
a_in = torch.randn(256, 4, 5000, device=dev); a_out = torch.empty_like(a_in)
b = torch.randn(64 * 1024 * 1024 // 4, device=dev)


def branch_a():
    a_out.copy_(a_in)
    a_out.mul_(1.0001)


def branch_b():
    x = b
    for _ in range(6):
        x = torch.sin(x) * 1.0001 + 0.1
    return x


def capture(fn):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    return g


side = torch.cuda.Stream()


def both():
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)              # fork
    with torch.cuda.stream(side):
        branch_a()
    branch_b()
    cur.wait_stream(side)              # join


def time_graph(g, n=200):
    for _ in range(10):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


ga, gb, gab = capture(branch_a), capture(branch_b), capture(both)
ta, tb, tab = time_graph(ga), time_graph(gb), time_graph(gab)
print(f"graph A alone {ta:.1f} us, graph B alone {tb:.1f} us, graph with both branches {tab:.1f} us "
      f"(sum {ta + tb:.1f}, max {max(ta, tb):.1f})")
