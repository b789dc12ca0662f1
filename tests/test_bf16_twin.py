"""The bf16 trainer's host logic (csrc/tnn_mlp.cpp: the three launch sequences of the single-GPU step, the lazily re-derived
first-layer bf16 copy) on whatever backend the suite runs on — in the build container that is the CPU twin of the C-ABI, so
the sequencing code is exercised without a GPU.  The test body is the GPU suite's."""
import test_gpu_bf16


def test_bf16_trainer_step_forms_on_this_backend(monkeypatch):
    test_gpu_bf16.test_bf16_trainer_fused_step_equals_separate_launches(monkeypatch)
