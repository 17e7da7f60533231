import sys, os, time
sys.path.insert(0, os.getcwd())
import torch, bench
dev = torch.device("cuda:0")
orig = torch.optim.Adam
for fused in (False, True, False, True):
    if fused:
        torch.optim.Adam = lambda params, lr: orig(params, lr=lr, fused=True)
    else:
        torch.optim.Adam = orig
    one, info = bench.train_step_setup(dev)
    for _ in range(6): one()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): last = one()
    torch.cuda.synchronize()
    print("fused", fused, round((time.time() - t0) / 20 * 1e3, 3), last)
