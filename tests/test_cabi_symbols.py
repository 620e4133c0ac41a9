"""CPU: the C-ABI library loads (no GPU needed) and exports every function include/falnet_hip.h declares,
and the ctypes binding covers exactly that set.  No compute calls."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
LIB = os.path.join(ROOT, "fal_net_amd", "libfalnet_hip.so")


def declared():
    src = open(os.path.join(ROOT, "include", "falnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(falnet_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_hot_path():
    names = declared()
    for must in ("falnet_conv2d", "falnet_wgrad", "falnet_med_head_fwd", "falnet_med_head_bwd", "falnet_med_masks_fwd",
                 "falnet_l1_fwd", "falnet_mse_fwd", "falnet_smooth_fwd", "falnet_adam_step", "falnet_hflip"):
        assert must in names


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built: run python __graft_entry__.py")
def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(LIB)
    for name in declared():
        assert hasattr(lib, name), f"{name} declared in falnet_hip.h but not exported"
    lib.falnet_version.restype = ctypes.c_int
    assert lib.falnet_version() >= 100


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_binding_matches_header():
    from fal_net_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared()
    _lib.lib()  # resolves every symbol with its argtypes
    assert _lib.lib().falnet_channel_pad(0) == _lib.CPAD == 32


def test_struct_layout_matches_header_field_order():
    """falnet_conv_t / falnet_wgrad_t are mirrored by ctypes.Structure: same field names in the same order."""
    from fal_net_amd import _lib
    src = open(os.path.join(ROOT, "include", "falnet_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for cname, cls in (("falnet_src_t", _lib.Src), ("falnet_conv_t", _lib.Conv), ("falnet_wgrad_t", _lib.Wgrad)):
        end = src.index("} " + cname)
        body = src[src.rindex("typedef struct {", 0, end) + len("typedef struct {"):end]
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):
                fields.append(re.sub(r"\[.*?\]", "", part.strip().split()[-1]).lstrip("*"))
        assert fields == [f[0] for f in cls._fields_], cname
