import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lidarseg3d_amd import ops
dev = torch.device("cuda:0")
for n, bits in ((65864, 20), (141896, 20), (141896, 27), (86569, 27)):
    g = torch.Generator().manual_seed(1)
    keys = torch.randint(0, 1 << bits, (n,), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
    want = torch.sort(keys.long(), stable=True)[1].int()
    got = ops.radix_argsort(keys, bits)
    assert torch.equal(got, want)
    for _ in range(5):
        ops.radix_argsort(keys, bits)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for _ in range(20):
            ops.radix_argsort(keys, bits)
    gph.replay(); torch.cuda.synchronize()
    s.record(); gph.replay(); e.record(); torch.cuda.synchronize()
    print("n %6d bits %2d: %.1f us per sort (%d passes)" % (n, bits, s.elapsed_time(e) * 1e3 / 20, (bits + 7) // 8))
