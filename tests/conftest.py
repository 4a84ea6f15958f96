import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["sharedbottom_ml", "mmoe_kuairec", "ple_ijcai", "mmoe_ae30", "mmoe_ae30d", "star_amazon",
                "pepnet_amazon", "mlp_ml", "mlp_ae", "esmm_ml",
                "cross_stitch_ae", "hmoe_ml", "aitm_ml", "snr_trans_ae",
                "mssm_ml", "sharedbottom_bn", "mmoe_bn",
                "mssm_bn", "cross_stitch_bn", "ple_l2", "escm_ml", "apg_ae", "star_dbn"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return request.param, load_golden(request.param)


def bn_noise_keys(keys):
    """Keys whose comparison is noise-driven in a model with BatchNorm: the bias of a Linear that feeds a BatchNorm has
    a mathematically ZERO gradient (the batch mean is subtracted again), so the reference's own gradient is fp32 noise
    (~1e-7) -- and Adam / Adagrad turn the SIGN of that noise into full lr-sized steps.  The model output does not
    depend on these biases; the BatchNorm's running_mean follows them (it is checked strictly after the first step,
    whose forward still saw the initial bias)."""
    keys = set(keys)
    bias = {k for k in keys if ".linears." in k and k.endswith(".bias")
            and k.replace(".linears.", ".bn.").replace(".bias", ".weight") in keys}
    rmean = {k.replace(".linears.", ".bn.").replace(".bias", ".running_mean") for k in bias}
    return bias, rmean & keys
