#!/usr/bin/env python3
"""Development aid: many GAN iterations on changing batches — finite losses, no growth of device memory outside torch's allocator."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.argv = [sys.argv[0], "--steps", "2"]
import gan_bench as GB  # noqa: E402

t = GB.trainer
data = GB.SyntheticPairs(256, 60, 13, 80, seed=1)
col = GB.WindowCollater(2000, 80, 512, np.random.default_rng(1))
free0 = None
for it in range(int(os.environ.get("SOAK_ITERS", "300"))):
    idx = np.random.default_rng(it).integers(0, 256, 64)
    log = t.train_step(col([data[i] for i in idx]))
    if it % 50 == 0 or it == 10:
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        other = total - free - torch.cuda.memory_reserved()
        if it == 10:
            free0 = other
        print(f"iter {it}: " + ", ".join(f"{k.split('/')[1]} {float(v):.3f}" for k, v in sorted(log.items())) + f"; non-torch device memory {other / 2**20:.0f} MiB, "
              f"torch reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB", flush=True)
        assert all(np.isfinite(float(v)) for v in log.values())
torch.cuda.synchronize()
free, total = torch.cuda.mem_get_info()
other = total - free - torch.cuda.memory_reserved()
print(f"non-torch device memory grew by {(other - free0) / 2**20:.1f} MiB over the run")
assert other - free0 < 64 * 2**20
print("soak ok")
