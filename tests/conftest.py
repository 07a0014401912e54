import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")

# generator_params of the reference's egs/ema/voc1/conf/e2w_hifigan.yaml:35-56 (values, not the file)
E2W_PARAMS = dict(
    in_channels=141, out_channels=1, channels=512, kernel_size=7,
    upsample_scales=[5, 4, 2, 2], upsample_kernel_sizes=[10, 8, 4, 4],
    resblock_kernel_sizes=[3, 7, 11], resblock_dilations=[[1, 3, 5], [1, 3, 5], [1, 3, 5]],
    use_additional_convs=True, bias=True, nonlinear_activation="LeakyReLU",
    nonlinear_activation_params={"negative_slope": 0.1}, use_weight_norm=True,
    use_ar=True, ar_input=512, ar_hidden=256, ar_output=128,
)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def e2w_params():
    return dict(E2W_PARAMS)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_err(a, b):
    """max|a-b| / max|b| — the normalisation BASELINE.md §4 states for the 1e-3 parity gate."""
    import numpy as np

    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def same_across_shapes(a, b, tol=5e-6):
    """Waveforms of the same utterance from launches of DIFFERENT shapes (batch size, number of utterances in flight).

    With the split-K conv form off (HIFICAR_KSPLIT=0) every launch shape uses the same accumulation order and the results are
    bit-identical (tests/test_gpu_parity.py::test_batch_invariance_is_bitwise_without_split_k).  By default small launches use the
    split-K form, whose partial sums are added in a different (fixed) order: equal to fp32 rounding."""
    import os
    import numpy as np
    import torch

    if os.environ.get("HIFICAR_KSPLIT") == "0":
        return torch.equal(torch.as_tensor(a), torch.as_tensor(b))
    a = np.asarray(a.detach().cpu() if hasattr(a, "detach") else a, dtype=np.float64)
    b = np.asarray(b.detach().cpu() if hasattr(b, "detach") else b, dtype=np.float64)
    if a.shape != b.shape:
        return False
    if a.size == 0:
        return True
    return float(np.abs(a - b).max()) <= tol * max(float(np.abs(b).max()), 1e-3)
