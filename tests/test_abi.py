"""The C-ABI library loads and exports every symbol include/sr_hip.h declares (no compute: no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sr_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-zA-Z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("sr_create", "sr_destroy", "sr_load_tensor", "sr_commit_weights", "sr_step", "sr_step_dev",
                 "sr_clip_begin", "sr_clip_push", "sr_clip_push_dev", "sr_step_hr_warp", "sr_fnet", "sr_srnet",
                 "sr_op_conv3x3", "sr_op_convT3x3s2", "sr_op_maxpool2", "sr_op_resize_bilinear", "sr_op_warp",
                 "sr_op_space_to_depth", "sr_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    import sr_hip
    assert os.path.exists(sr_hip.LIB_PATH), "libsr_hip.so missing: run __graft_entry__.build()"
    lib = ctypes.CDLL(sr_hip.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in sr_hip.h but not exported"


def test_binding_covers_every_declared_symbol():
    import sr_hip
    assert sorted(sr_hip.SIGNATURES) == declared_symbols()


def test_no_cpu_fallback_and_argument_errors_without_gpu():
    """Argument / shape validation happens before any device work; with no HIP device the
    library fails loudly instead of computing on the CPU."""
    import sr_hip
    if sr_hip.device_count() > 0:
        pytest.skip("a GPU is present; this test documents the no-GPU behaviour")
    with pytest.raises(ValueError):
        sr_hip.Context(20, 16)                 # not a multiple of 8
    with pytest.raises(ValueError):
        sr_hip.Context(16, 16, scale=2)
    with pytest.raises(ValueError):
        sr_hip.Context(16, 16, ch=1)
    with pytest.raises(RuntimeError, match="no HIP device"):
        sr_hip.Context(16, 16)
    import numpy as np
    with pytest.raises(RuntimeError, match="no HIP device"):
        sr_hip.maxpool2(np.zeros((1, 4, 4, 4), np.float32))
