import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# The sample-major / streaming forms of the first layer's backward (round 6) are the default from batch 32768 on (below it a
# step is bound by its launches, not their work: EmbeddingLayer._smp_tables).  The model-level tests run batches of a few
# thousand samples: they take the round-6 forms too, so that bit-identity / parity / capture tests cover them end to end;
# the C-ABI tests call every form directly, and bench.full_size_parity runs the default choice at B = 65536.
os.environ.setdefault("RP_SMP_MIN_BATCH", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def load_golden(name):
    """npz -> {group: {key: torch tensor}} split on the first '/'."""
    z = np.load(os.path.join(GOLDEN, name))
    out = {}
    for k in z.files:
        grp, _, rest = k.partition("/")
        t = torch.from_numpy(z[k])
        if rest:
            out.setdefault(grp, {})[rest] = t
        else:
            out[grp] = t
    return out


ENC_ORDER = ["I1", "C1", "C2", "I2", "C3", "I3", "C4", "C5"]
VOCAB = {"C1": 7, "C2": 3, "C3": 50, "C4": 11, "C5": 2}


def small_enc_dict():
    """Same ordered enc_dict tests/golden/make_golden.py fed the reference."""
    return {k: ({"min": 0.0, "max": 1.0} if k.startswith("I") else {"vocab_size": VOCAB[k]}) for k in ENC_ORDER}


# the ordered enc_dict of tests/golden/adam_long.npz (make_golden_r4.py feeds it to the reference's DeepFM)
ADAM_LONG_ENC = {"I1": {"min": 0.0, "max": 1.0}, "I2": {"min": 0.0, "max": 1.0},
                 "C1": {"vocab_size": 3000}, "C2": {"vocab_size": 700}, "C3": {"vocab_size": 40}, "C4": {"vocab_size": 5}}


@pytest.fixture(scope="session")
def enc_dict():
    return small_enc_dict()


def require_gpu():
    if not torch.cuda.is_available():
        pytest.fail("this test is marked gpu and needs a visible MI355X")
