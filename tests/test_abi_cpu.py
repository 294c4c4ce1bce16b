"""CPU: the C-ABI library loads and exports every symbol declared in include/icgan_hip.h (no compute calls)."""
import ctypes
import os

import pytest


def test_library_built_and_exports_every_declared_symbol():
    from ic_gan_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    protos = _lib.parse_header()
    assert len(protos) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in protos if not hasattr(lib, n)]
    assert not missing, missing
    assert _lib.lib().icg_version() >= 100
    assert _lib.lib().icg_strerror(-3).decode().startswith("workspace")


def test_descriptor_structs_match_header_layout():
    from ic_gan_amd import _lib
    assert ctypes.sizeof(_lib.AdamTensor) == 40 and ctypes.sizeof(_lib.EmaTensor) == 24
    assert ctypes.sizeof(_lib.SnLayer) == 13 * 8 + 8 + 4 * 4


def test_product_fails_loudly_without_gpu():
    """no CPU fallback: an operator on a CPU tensor raises instead of computing something else."""
    import torch
    from ic_gan_amd import layers
    conv = layers.SNConv2d(4, 4, 3, padding=1)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="AMD GPU only|no CPU path"):
        conv(torch.randn(1, 4, 8, 8))


def test_workspace_queries_are_pure_host_functions():
    from ic_gan_amd import _lib
    assert _lib.query("icg_conv2d_wgrad_workspace_bytes", 64, 256, 256, 96, 96, 3) > 16      # split-K plan
    assert _lib.query("icg_conv2d_wgrad_workspace_bytes", 64, 4, 4, 1536, 1536, 3) == 16     # no split needed
    assert _lib.query("icg_bn_workspace_bytes", 64 * 256 * 256, 96) > 0
    assert _lib.query("icg_sn_scratch_bytes", 1536, 1536, 3) > 0
