"""`pytest -m gpu`: the same parity suite through libtnn_hip.so on the MI355X (HIP kernels, pool
allocator, real hipGraph capture)."""

import pytest

import parity_suite
import tinynn_autograd_amd as tn


@pytest.mark.gpu
def test_backend_is_hip():
    assert tn.backend_name() == "hip-gfx950"


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(parity_suite.SUITE))
def test_gpu_parity(name):
    assert tn.backend_name() == "hip-gfx950"
    parity_suite.SUITE[name]()
