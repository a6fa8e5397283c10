"""Host-side sanitizer pass over the C entry points: the same .hip sources, built for x86 against the
SIMT emulator with AddressSanitizer + UndefinedBehaviorSanitizer (tests/hipemu/build_emu.build(sanitize=True)), and a
subset of the emulator parity tests run against that library in a child Python with clang's ASan runtime preloaded.  Every
extern "C" function's argument handling, the host-side layout / launch logic and the kernels' index arithmetic (LDS tiles,
row maps with padding and ragged tails, the DMA swizzle, the destination-sorted gather) execute under the sanitizers; an
out-of-bounds access, a misaligned vector access or signed overflow aborts the child."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))

SUBSET = [
    "tests/test_emu_nce.py::test_nce_forward_backward_emulated[2-21-7-32-3000.0-0-1]",
    "tests/test_emu_nce.py::test_nce_forward_backward_emulated[2-21-7-32-3000.0-0-0]",
    "tests/test_emu_encoder.py::test_encoder_forward_backward_emulated[3-1290-0-734]",
    "tests/test_emu_gru.py::test_persistent_recurrence_in_chunks_of_batch_tiles_emulated",
    "tests/test_emu_nce.py::test_nce_scores_of_foreign_predictions_emulated[2-21-7-32]",
    "tests/test_emu_nce.py::test_out_of_range_negative_indices_are_clamped_and_flagged",
    "tests/test_emu_nce.py::test_more_than_sixteen_heads_walked_in_groups_emulated[1-41-35-24-1]",
    "tests/test_emu_nce.py::test_criterion_at_the_edges_of_its_shapes_emulated[1-13-12-1-1]",
    "tests/test_emu_nce.py::test_criterion_at_the_edges_of_its_shapes_emulated[1-6-2-1040-1]",
    "tests/test_emu_nce.py::test_criterion_at_the_edges_of_its_shapes_emulated[3-5-1-3-1]",
    "tests/test_emu_train_step.py::test_composite_step_matches_oracle_and_the_stagewise_step_emulated[1-800-2-1-False]",
    "tests/test_emu_encoder.py::test_encoder_forward_backward_emulated[1-170-0-734]",
    "tests/test_emu_adam.py",
    "tests/test_emu_encoder.py::test_encoder_forward_backward_emulated[2-1280-0-3]",
    "tests/test_emu_encoder.py::test_encoder_forward_backward_emulated[1-1370-64-1]",
    "tests/test_emu_encoder.py::test_encoder_forward_backward_emulated[2-1280-0-34]",
    "tests/test_emu_train_step.py::test_prefetched_index_lists_give_the_same_step_emulated",
    "tests/test_emu_train_step.py::test_composite_step_argument_errors",
    "tests/test_emu_encoder.py::test_dma_conv_kernel_matches_the_register_staged_kernel_emulated[2-131-4-2-1-256-False-0]",
    "tests/test_emu_encoder.py::test_dma_conv_kernel_matches_the_register_staged_kernel_emulated[1-70-8-4-2-256-True-1]",
    "tests/test_emu_encoder.py::test_dma_conv_kernel_matches_the_register_staged_kernel_emulated[1-300-8-4-2-256-True-4]",
    "tests/test_emu_encoder.py::test_dma_conv_kernel_matches_the_register_staged_kernel_emulated[1-300-8-4-2-256-True-6]",
    "tests/test_emu_gru.py::test_gru_forward_backward_emulated[3-6-2-False]",
    "tests/test_emu_gru.py::test_gru_persistent_equals_stepwise_emulated[3-6-False]",
    "tests/test_emu_transformer.py::test_transformer_layer_forward_backward_emulated[1-40-False]",
]


def test_c_entry_points_under_asan_and_ubsan():
    import build_emu
    try:
        build_emu.build(sanitize=True)
        rt = build_emu.asan_runtime()
    except FileNotFoundError as e:
        pytest.skip(f"no host clang: {e}")
    if rt is None:
        pytest.skip("clang's shared ASan runtime not found")
    env = dict(os.environ)
    env.update({"LD_PRELOAD": rt, "CPC_EMU_SANITIZE": "1",
                # the emulator switches between its own fiber stacks: no fake stacks; python itself leaks by design
                "ASAN_OPTIONS": "detect_leaks=0:detect_stack_use_after_return=0:abort_on_error=1",
                "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1"})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", *SUBSET], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    assert " passed" in r.stdout, tail
