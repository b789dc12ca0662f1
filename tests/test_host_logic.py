"""`pytest -m "not gpu"`: the parity suite on the CPU twin of the C-ABI — covers the host logic
(DeviceArray protocol, Tensor/ops graph + backward schedule, layers, optimizers, Model arenas, trainer
host code, hipGraph-style capture/replay) against the reference's golden fixtures without a GPU."""

import pytest

import parity_suite


@pytest.mark.parametrize("name", sorted(parity_suite.HOST_SUITE))
def test_host_logic(name):
    parity_suite.HOST_SUITE[name]()
