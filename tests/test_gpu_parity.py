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


# The host layer runs from ahead-of-time compiled copies of its modules when they are fresh (_host_build.py); the cases that
# exercise the autograd bookkeeping hardest run once more with the interpreter executing the same sources.
INTERPRETED = ["op_cases_float32", "op_cases_float64", "traj_A_adam_fused", "traj_A_adam_generic_ops", "traj_A_adam_no_arena",
               "fused_classifier_head_matches_generic_chain", "fused_dense_relu_node_matches_generic_chain",
               "dense_vjp_writes_arena_views_and_survives_weight_sharing", "epoch_loop_ops_path_matches_reference",
               "op_level_step_captured_in_graph", "error_behaviour", "views_indexing_and_numpy_protocol"]


@pytest.mark.gpu
def test_gpu_parity_with_the_host_modules_interpreted():
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import parity_suite, tinynn_autograd_amd as tn\n"
            "assert tn.host_modules_compiled() == [], tn.host_modules_compiled()\n"
            "assert tn.backend_name() in ('hip-gfx950', 'unloaded')\n"
            "for name in %r:\n"
            "    parity_suite.SUITE[name]()\n"
            "assert tn.backend_name() == 'hip-gfx950'\n"
            "print('INTERPRETED-OK')\n") % (here, os.path.dirname(here), INTERPRETED)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TNN_HOST_COMPILED="0"),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "INTERPRETED-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
