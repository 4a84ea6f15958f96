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
                "mssm_ml"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


@pytest.fixture(params=GOLDEN_CASES)
def golden(request):
    return request.param, load_golden(request.param)
