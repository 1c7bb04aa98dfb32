"""The C-ABI library loads on a CPU-only host and exports every symbol include/nerf_hip.h declares
(no compute calls here: there is no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'nerf_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(nerf_[a-z0-9_]+)\s*\(', text)))


def test_header_symbols_exported():
    from nerf_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), 'libnerf_hip.so does not export ' + n
        assert n in _lib.SIGNATURES, 'no ctypes signature for ' + n
    assert sorted(_lib.SIGNATURES) == names


def test_host_side_config_helpers():
    """nerf_grid_cfg_init is pure host code: check it against the oracle's level table."""
    from nerf_amd import _lib
    from oracle.ref_cpu import grid_levels
    l = _lib.lib()
    assert l.nerf_abi_version() == 4
    for L, T in ((20, 24), (16, 19), (8, 15), (5, 14)):
        cfg = _lib.NerfGridCfg()
        total = l.nerf_grid_cfg_init(ctypes.byref(cfg), L, 2, T, 16, 1.3819)
        levels, total_o = grid_levels(L, T)
        assert total == total_o
        for i, lv in enumerate(levels):
            assert cfg.res[i] == lv.res and cfg.entries[i] == lv.entries and cfg.offset[i] == lv.offset
            assert bool(cfg.hashed[i]) == lv.hashed
            assert cfg.scale[i] == lv.scale      # bit-equal: the cell a point falls in depends on it
    net = _lib.NerfNetCfg(64, 2, 15, 3, 4, 48)
    cfg = _lib.NerfGridCfg()
    l.nerf_grid_cfg_init(ctypes.byref(cfg), 20, 2, 24, 16, 1.3819)
    assert l.nerf_sigma_param_count(ctypes.byref(cfg), ctypes.byref(net)) == 40 * 64 + 64 * 16
    assert l.nerf_color_param_count(ctypes.byref(net)) == 79 * 64 + 64 * 64 + 64 * 3
    bad = l.nerf_grid_cfg_init(ctypes.byref(cfg), 0, 2, 24, 16, 1.3819)
    assert bad < 0 and b'invalid' in l.nerf_last_error()


def test_product_path_refuses_cpu_tensors():
    """No CPU fallback: handing the product path a CPU tensor must fail loudly."""
    import torch
    from nerf_amd import _lib
    with pytest.raises(RuntimeError):
        _lib.ptr(torch.zeros(4))
