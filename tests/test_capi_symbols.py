"""CPU: libmappo_hip.so loads and exports every symbol include/mappo_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mappo_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mappo_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from mappo_amd import _lib
    names = declared_symbols()
    assert len(names) >= 15
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in mappo_hip.h but not exported by libmappo_hip.so"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in mappo_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)


def test_library_loads_and_reports_version():
    from mappo_amd import _lib, ops
    lib = _lib.load()
    assert lib.mappo_abi_version() >= 1
    # host-only helpers are callable without a GPU
    assert ops.net_param_count(ops.net_desc(18, 5)) == 10281 - 4288       # reference actor minus the unused fc_h
    assert ops.net_param_count(ops.net_desc(54, 1)) == 12397 - 4288
    assert ops.mlp_backward_slabs(76800) == 256 and ops.mlp_backward_slabs(40) == 2


def test_bad_arguments_return_error_codes_not_crashes():
    from mappo_amd import _lib
    lib = _lib.load()
    rc = lib.mappo_gae_scan(None, None, None, None, None, None, None, 0, 0, 0.99, 0.95, 1, 0, None)
    assert rc == -1 and b"gae_scan" in lib.mappo_last_error()
    d = _lib.NetDesc(18, 128, 5, 1, 1, 1, 0)            # hidden 128 is not built
    rc = lib.mappo_mlp_forward(None, ctypes.byref(d), None, None, 4, None, None)
    assert rc == -1 and b"hidden_size" in lib.mappo_last_error()
