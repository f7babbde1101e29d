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


def test_wide_entry_points_reject_bad_arguments_and_layout_is_a_pure_function():
    """mappo_wide_layout / mappo_wide_l1_backward (include/mappo_hip.h): argument validation happens before any launch, so the
    error paths run without a GPU.  The layout a producer leaves is a function of (descriptor, producer) only — no host state."""
    from mappo_amd import _lib, ops
    lib = _lib.load()
    wide_mlp = _lib.NetDesc(512, 64, 5, 1, 1, 1, 0)
    wide_critic = _lib.NetDesc(512, 64, 1, 1, 1, 1, 0)
    wide_l2 = _lib.NetDesc(176, 64, 18, 2, 1, 1, 0)       # layer_N = 2: K-chunked kernels
    wide_rec = _lib.NetDesc(322, 64, 1, 1, 1, 1, 1)
    narrow = _lib.NetDesc(54, 64, 1, 1, 1, 1, 0)
    assert ops.wide_layout(wide_mlp, ops.PRODUCER_ACTOR_UPDATE) == 1 and ops.wide_layout(wide_mlp, ops.PRODUCER_MLP_BACKWARD) == 0
    assert ops.wide_layout(wide_l2, ops.PRODUCER_ACTOR_UPDATE) == 0 and ops.wide_layout(wide_l2, ops.PRODUCER_TRUNK_BACKWARD) == 0
    assert ops.wide_layout(wide_rec, ops.PRODUCER_TRUNK_BACKWARD) == 1 and ops.wide_layout(wide_rec, ops.PRODUCER_CRITIC_UPDATE) == 0
    for _ in range(3):                                    # same answer every time, whatever was asked in between
        assert ops.wide_layout(wide_critic, ops.PRODUCER_CRITIC_UPDATE) == ops.wide_layout(wide_mlp, ops.PRODUCER_ACTOR_UPDATE) == 1
        assert ops.wide_layout(wide_mlp, ops.PRODUCER_CRITIC_UPDATE) == 0          # out_dim 5 is not a critic: K-chunked kernels
    assert lib.mappo_wide_layout(None, 1) == -1 and lib.mappo_wide_layout(ctypes.byref(wide_mlp), 7) == -1
    assert b"wide_layout" in lib.mappo_last_error()
    fake = ctypes.c_void_p(4096)                          # non-NULL, never dereferenced: validation fails first
    bw = lambda d, B, layout, ws=fake, stride=1 << 20: lib.mappo_wide_l1_backward(fake, ctypes.byref(d), fake, None, B, ws, fake, stride, 0,
                                                                                  layout, None)
    assert bw(narrow, 64, 1) == -1 and b"in_dim 54" in lib.mappo_last_error()
    assert bw(wide_mlp, 0, 1) == -1 and b"bad arguments" in lib.mappo_last_error()
    assert bw(wide_mlp, 64, 1, ws=None) == -1 and b"bad arguments" in lib.mappo_last_error()
    assert bw(wide_mlp, 64, 2) == -1 and b"layout 2" in lib.mappo_last_error()
    assert bw(wide_mlp, 64, -1) == -1 and b"layout -1" in lib.mappo_last_error()
    assert bw(wide_mlp, 64, 1, stride=100) == -1 and b"slab column range" in lib.mappo_last_error()
    assert lib.mappo_wide_workspace_floats(1000) >= 66 * 1000 and lib.mappo_wide_l1_slabs(76800) == 256
    # producers without their workspace
    rc = lib.mappo_trunk_backward(fake, ctypes.byref(wide_rec), fake, None, 64, fake, fake, 1 << 20, 0, None, None)
    assert rc == -1 and b"wide workspace" in lib.mappo_last_error()
