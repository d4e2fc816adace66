import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle runs small networks: on the GPU box's 128 host threads oneDNN spends its time synchronising them (the Xception
    # mIoU trajectory: 180 s there against 30 s on 8 cores)
    import torch
    if torch.get_num_threads() > 16:
        torch.set_num_threads(16)


@pytest.fixture(scope='session')
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    return torch.device('cuda:0')


def needs_experimental():
    """Skip marker for the tests of what is built only with `make -C pylc_amd/csrc EXPERIMENTAL=1` (persistent 1x1 kernels, the dgrad epilogue
    with BatchNorm-backward sums, CU-masked streams: measured neutral / negative, absent from the product library)."""
    from pylc_amd import lib as L
    return pytest.mark.skipif(not L.HAS_EXPERIMENTAL, reason='needs a library built with EXPERIMENTAL=1 (off in the product)')
