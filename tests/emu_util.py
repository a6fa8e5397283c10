"""Helpers for the emulator-backed kernel tests (CPU, no GPU needed).

The kernels are compiled for x86 against tests/hipemu (a SIMT interpreter) and called
through the same C ABI and the same ctypes signature table as the product library."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))

# Resident workgroup slots of the emulated device (OS worker threads).  The persistent GRU kernels need
# all of their 32 * ceil(B/16) workgroups resident at once; 64 covers the B <= 32 cases tested here.
os.environ.setdefault("HIPEMU_THREADS", "64")

_emu = None


def emu():
    global _emu
    if _emu is None:
        import build_emu
        from cpc_audio_amd import _lib
        try:
            # CPC_EMU_SANITIZE=1 (tests/test_emu_sanitized.py runs a subset that way, with clang's ASan runtime preloaded):
            # the AddressSanitizer + UBSan build of the same sources
            path = build_emu.build(sanitize=os.environ.get("CPC_EMU_SANITIZE") == "1")
        except FileNotFoundError as e:  # no host clang: cannot emulate (a compile ERROR still fails the test)
            pytest.skip(f"emulator build unavailable: {e}")
        _emu = _lib.bind(path)
    return _emu


def P(t):
    return None if t is None else t.data_ptr()


def rel_err(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()
