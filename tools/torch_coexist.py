import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "channel-pruning_amd"))
order = sys.argv[1]
import numpy as np
def maps():
    libs = set()
    for l in open("/proc/self/maps"):
        if "amdhip64" in l or "hsa-runtime" in l:
            libs.add(l.split()[-1])
    return sorted(libs)
if order == "torch_first":
    import torch
    t = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
    print("torch ok", maps())
    from cpmi355 import capi
    ctx = capi.Context(0)
    print("mfma probe", ctx.probe_mfma_f64(), maps())
    b = torch.ones(8, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
    print(ctx.to_host(b, (8,), np.float64))
else:
    from cpmi355 import capi
    ctx = capi.Context(0)
    print("mfma probe", ctx.probe_mfma_f64(), maps())
    import torch
    t = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
    print("torch ok", maps())
