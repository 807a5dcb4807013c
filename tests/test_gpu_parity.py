"""-m gpu: parity of the HIP path (through the C ABI) against the oracle / ABI-contract emulation / reference goldens."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _checks():
    from tests import gpu_checks
    return gpu_checks.ALL


def _names():
    import importlib
    try:
        return [f.__name__ for f in importlib.import_module("tests.gpu_checks").ALL]
    except Exception:      # collection on a box without torch.cuda must still work
        return []


@pytest.mark.parametrize("name", _names())
def test_gpu(name):
    assert torch.cuda.is_available(), "needs the MI355X"
    from ipercore_amd import _lib
    _lib.lib()          # fail loudly if the HIP library is missing: no fallback exists
    from tests import gpu_checks
    metrics = getattr(gpu_checks, name)()
    print(name, metrics)
