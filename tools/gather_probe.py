"""What slows gather_fm_fwd in situ?  (a) table size / TLB reach, (b) cold caches between launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recsys_benchmark_amd import _lib
from recsys_benchmark_amd.profiling import KernelTimer
dev = torch.device("cuda"); lib = _lib.load()
B, F, D = 4096, 26, 16
def run(N, thrash, tag):
    per = N // F
    W = torch.rand(per * F, D, device=dev); w1 = torch.rand(per * F, device=dev); bias = torch.zeros(1, device=dev)
    off = (torch.arange(F) * per).to(dev)
    xs = [torch.randint(0, per, (B, F), device=dev) for _ in range(8)]
    emb = torch.empty(B, F, D, device=dev); y = torch.empty(B, device=dev); rows = torch.empty(B, F, dtype=torch.int64, device=dev)
    junk = torch.empty(128 << 20, device=dev) if thrash else None       # 512 MB
    s = _lib.stream_ptr(dev)
    def go(i):
        x = xs[i % 8]
        lib.mi_gather_fm_fwd(x.data_ptr(), off.data_ptr(), W.data_ptr(), w1.data_ptr(), bias.data_ptr(), emb.data_ptr(),
                             y.data_ptr(), rows.data_ptr(), B, F, D, per * F, None, s)
    for i in range(5): go(i)
    torch.cuda.synchronize()
    with KernelTimer(256) as kt:
        for i in range(40):
            if thrash: junk.fill_(1.0)
            go(i)
        torch.cuda.synchronize()
    v = kt.summary()["gather_fm_fwd"]
    print(f"{tag:44s} avg {v['avg_us']:.2f} us  min {v['min_us']:.2f} us")
run(260000, False, "table 16 MB, back-to-back")
run(33762577, False, "table 2.16 GB, back-to-back")
run(260000, True, "table 16 MB, 512 MB fill between launches")
run(33762577, True, "table 2.16 GB, 512 MB fill between launches")
