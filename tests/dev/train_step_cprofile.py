"""cProfile of the host side of Trainer.train_step at the recipe's size (where do 44 ms of host time per iteration go?)."""
import cProfile
import os
import pstats
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import torch  # noqa: E402

from articulatory_amd.bin.train import Trainer  # noqa: E402
from articulatory_amd.utils.recipes import recipe_train_config  # noqa: E402
from articulatory_amd.utils.synth import synth_train_batch  # noqa: E402

cfg = recipe_train_config("car", fused_optimizers="--fused" in sys.argv)
t = Trainer(cfg, torch.device("cuda"))
batch = {k: torch.from_numpy(v) for k, v in synth_train_batch(cfg, 1, 64).items()}
t.steps = 2
for _ in range(3):
    t.train_step(batch)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    t.train_step(batch)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
