import torch, time
for mb in (132, 528, 2112):
    n = mb * 1024 * 1024 // 4
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    for _ in range(5): out.fill_(1.0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): out.fill_(1.0)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(mb, "MB fill:", us, "us", mb * 1.048576e6 / us / 1e6, "GB/s")
    src = torch.empty_like(out)
    for _ in range(3): out.copy_(src)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20): out.copy_(src)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(mb, "MB copy:", us, "us", 2 * mb * 1.048576e6 / us / 1e6, "GB/s (r+w)")
