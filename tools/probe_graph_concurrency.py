#!/usr/bin/env python
"""Does a captured hipGraph run its parallel branches concurrently?  Two (or four) independent single-workgroup spin kernels on separate streams,
forked from and joined to the capture stream: eager submission against graph replay.  (Measurement helper; needs the MI355X.)"""
import time, sys, json
import torch
dev = torch.device("cuda:0")
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000

def body(streams):
    main = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(main)
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    for s in streams:
        main.wait_stream(s)

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n

out = {}
for nb in (1, 2, 4):
    streams = [torch.cuda.Stream() for _ in range(nb)]
    out["eager_%d_branches_ms" % nb] = timed(lambda: body(streams))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body(streams)
    out["graph_%d_branches_ms" % nb] = timed(g.replay)
print(json.dumps(out))
