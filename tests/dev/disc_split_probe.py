"""Is the discriminator step bound by its longest sub-network (critical path) or by total throughput?  D-step time of the scale subs
alone, the period subs alone, single subs, and all eight (streams on).  python tests/dev/disc_split_probe.py"""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from articulatory_amd.models import HiFiGANMultiScaleMultiPeriodDiscriminator  # noqa: E402
from articulatory_amd.utils.recipes import recipe_train_config  # noqa: E402
from articulatory_amd.utils.synth import synth_disc_state_dict  # noqa: E402

base = recipe_train_config("car")["discriminator_params"]
real = torch.rand(64, 1, 2512, device="cuda") - 0.5
fake = torch.rand(64, 1, 2512, device="cuda") - 0.5
for tag, over in (("all 8", {}), ("3 scale subs", {"periods": []}), ("5 period subs", {"scales": 0}), ("scale 0 only", {"scales": 1, "periods": []}),
                  ("period 2 only", {"scales": 0, "periods": [2]}), ("period 11 only", {"scales": 0, "periods": [11]}),
                  ("scales + period 2", {"periods": [2]})):
    p = dict(base, **over)
    d = HiFiGANMultiScaleMultiPeriodDiscriminator(**p)
    d.load_state_dict({k: torch.from_numpy(v) for k, v in synth_disc_state_dict(p, seed=1).items()})
    d = d.cuda()

    def step():
        d.zero_grad(set_to_none=True)
        t, _, _ = d.discriminator_loss(fake, real, average_by_discriminators=False)
        t.backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    print(f"{tag:20s} D step {(time.perf_counter() - t0) * 100:.2f} ms   ({2.0 * d.macs(64, 2512) * 6 / 1e12:.3f} TFLOP)")
    del d
