import torch, time
dev = torch.device("cuda:0")
for H in (8, 16, 32, 64, 256):
    g = torch.randn(518000, H, device=dev); b = torch.randn(518000, H, device=dev)
    for _ in range(3): r = g.t() @ b
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): r = g.t() @ b
    e1.record(); torch.cuda.synchronize()
    t1 = e0.elapsed_time(e1) / 10
    e0.record()
    for _ in range(10): z = torch.zeros(518000, H, device=dev)
    e1.record(); torch.cuda.synchronize()
    t2 = e0.elapsed_time(e1) / 10
    # chunked: [C, n/C, H] batched then summed
    C_ = 256
    n = (518000 // C_) * C_
    e0.record()
    for _ in range(10): r2 = torch.bmm(g[:n].view(C_, -1, H).transpose(1, 2), b[:n].view(C_, -1, H)).sum(0)
    e1.record(); torch.cuda.synchronize()
    t3 = e0.elapsed_time(e1) / 10
    print(H, "gemm ms", round(t1, 4), "zeros ms", round(t2, 4), "bmm+sum ms", round(t3, 4), flush=True)
