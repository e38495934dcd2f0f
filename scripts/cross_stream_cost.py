#!/usr/bin/env python3
"""What a cross-stream dependency costs on this GPU / runtime (the price of every halo / interior overlap): a chain of
tiny kernels that alternates between two streams through events, against the same chain on one stream.

    python scripts/cross_stream_cost.py [hops=2000]"""
import sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
x = torch.zeros(64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
ev = [torch.cuda.Event() for _ in range(2)]


def one_stream():
    with torch.cuda.stream(s1):
        for _ in range(2 * n):
            x.add_(1.0)


def two_streams():
    for _ in range(n):
        with torch.cuda.stream(s1):
            x.add_(1.0)
            ev[0].record(s1)
        s2.wait_event(ev[0])
        with torch.cuda.stream(s2):
            x.add_(1.0)
            ev[1].record(s2)
        s1.wait_event(ev[1])


for name, fn in (("one stream", one_stream), ("two streams, event hand-over after every kernel", two_streams)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-50s %.2f us per kernel" % (name, 1e6 * dt / (2 * n)))
