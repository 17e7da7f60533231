#!/usr/bin/env python3
"""Does a buffer written by one kernel get read back from the Infinity Cache by the next?  fill -> sum on buffers of
16 MiB .. 2 GiB (torch kernels, HIP events): GB/s of the write and of the read that follows it."""
import torch
dev = torch.device("cuda:0")
for mib in (16, 32, 64, 96, 128, 192, 256, 512, 2048):
    n = mib * (1 << 20) // 4
    x = torch.empty(n, dtype=torch.float32, device=dev)
    y = torch.empty(n, dtype=torch.float32, device=dev)
    for _ in range(3):
        x.fill_(1.0); x.sum()
    torch.cuda.synchronize()
    reps = 20
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3 * reps)]
    for i in range(reps):
        ev[3 * i].record(); x.fill_(float(i)); ev[3 * i + 1].record(); s = x.sum(); ev[3 * i + 2].record()
    torch.cuda.synchronize()
    w = sum(ev[3 * i].elapsed_time(ev[3 * i + 1]) for i in range(reps)) / reps
    r = sum(ev[3 * i + 1].elapsed_time(ev[3 * i + 2]) for i in range(reps)) / reps
    # read of a buffer not written just before: alternate two buffers so x was evicted by y's traffic when large
    ev2 = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps)]
    for i in range(reps):
        y.fill_(0.0); ev2[2 * i].record(); s = x.sum(); ev2[2 * i + 1].record()
    torch.cuda.synchronize()
    r2 = sum(ev2[2 * i].elapsed_time(ev2[2 * i + 1]) for i in range(reps)) / reps
    gb = n * 4 / 1e9
    print(f"{mib:5d} MiB  fill {gb / w * 1e3:8.0f} GB/s ({w * 1e3:7.1f} us)   sum right after fill {gb / r * 1e3:8.0f} GB/s ({r*1e3:7.1f} us)   "
          f"sum after another buffer's fill {gb / r2 * 1e3:8.0f} GB/s")
