import time, torch, numpy as np
dev = torch.device("cuda:0")
data = bytes(np.random.default_rng(0).integers(0, 256, 687096, dtype=np.uint8))
def t(name, fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter()-t0)/n*1e6:9.1f} us"); return r
t("bytearray concat", lambda: bytearray(data) + bytearray(8))
t("frombuffer(...).to(dev)", lambda: torch.frombuffer(bytearray(data) + bytearray(8), dtype=torch.uint8).to(dev))
pin = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
def pinned():
    n = len(data)
    pin[:n].numpy()[:] = np.frombuffer(data, np.uint8)
    return pin[:n + 8].to(dev, non_blocking=True)
t("pinned staging + non_blocking", pinned)
d = torch.empty(687096, dtype=torch.uint8, device=dev)
t("d.cpu().numpy().tobytes()", lambda: d.cpu().numpy().tobytes())
pin2 = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
def d2h():
    pin2[:d.numel()].copy_(d, non_blocking=True); torch.cuda.synchronize(); return pin2[:d.numel()].numpy().tobytes()
t("pinned d2h + tobytes", d2h)
x = torch.zeros(1, dtype=torch.int64, device=dev)
t("x.item()", lambda: x.item())
t("torch.zeros(1) on dev", lambda: torch.zeros(1, dtype=torch.int64, device=dev))
t("torch.empty big", lambda: torch.empty((14864, 128), dtype=torch.int32, device=dev))
