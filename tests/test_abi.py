"""CPU-side checks of the C-ABI boundary: the shared library loads and exports every symbol
include/owl_hip.h declares (no compute calls without a GPU)."""
import ctypes
import os

import pytest

from owl_vit_object_detection_amd import _lib


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_parses():
    protos = _lib.parse_header()
    assert "owl_gemm_nt_bf16" in protos and "owl_last_error" in protos
    ret, args = protos["owl_gemm_nt_bf16"]
    assert ret == "int" and args[0] == ("void*", "stream") and len(args) == 21 and args[-1] == ("int", "tile")
    for name, (ret, args) in protos.items():
        for ty, _ in args:
            assert ty in _lib._CTYPES, (name, ty)


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _lib.parse_header():
        assert hasattr(lib, name), f"{name} declared in include/owl_hip.h but not exported"
    assert built.owl_abi_version() == _lib.header_abi_version() >= 3


def test_no_process_global_setters_in_the_product_abi(built):
    """re-entrant, no global mutable state.  The tuning switches (kernel-choice override, store skipping,
    attention debug flags) must be neither declared in the product header nor exported by the default build."""
    protos = _lib.parse_header()
    banned = ("owl_gemm_set_tile", "owl_gemm_set_persistent", "owl_gemm_debug_nostore", "owl_gemm_debug_slots", "owl_attention_debug")
    for name in banned:
        assert name not in protos, name
    assert not any("_set_" in n or "_debug" in n for n in protos), [n for n in protos if "_set_" in n or "_debug" in n]
    if os.environ.get("OWL_TUNING", "0") != "1":
        lib = ctypes.CDLL(_lib.LIB_PATH)
        for name in banned:
            assert not hasattr(lib, name), f"{name} exported by the default build"
    # every op that needs device scratch has a size query
    for q in ("owl_postprocess_workspace", "owl_box_final_bwd_blocks", "owl_gemm_effective_splits", "owl_gemm_tn_slab_workspace_bytes",
              "owl_attention_bwd_workspace_bytes", "owl_gemm_slab_workspace_bytes"):
        assert q in protos, q


def test_argument_validation_sets_error_without_gpu(built):
    # argument checks run before any HIP call: safe without a device
    with pytest.raises(_lib.OwlLibError, match="null pointer"):
        _lib.call("owl_gemm_nt_bf16", None, 0, None, 0, 0, None, 0, 0, None, None, 0, None, None, 0, 1, 4, 64, 1.0, 1, 0, 0)
    assert "null pointer" in _lib.last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """No CPU / eager fallback: without libowlhip.so every op raises (the product path never routes through the oracle)."""
    from owl_vit_object_detection_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "libowlhip.so"))
    with pytest.raises(L.OwlLibError, match="no CPU fallback|not found"):
        L.load()
    with pytest.raises(L.OwlLibError):
        L.call("owl_abi_version")


def test_abi_version_mismatch_is_refused(built, monkeypatch, tmp_path):
    """A binding generated from a header of another ABI version must not drive the library (changed argument lists / scratch layouts)."""
    from owl_vit_object_detection_amd import _lib as L
    hdr = tmp_path / "owl_hip.h"
    hdr.write_text(open(L.HEADER).read().replace(f"#define OWL_ABI_VERSION {L.header_abi_version()}", "#define OWL_ABI_VERSION 9999"))
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "HEADER", str(hdr))
    with pytest.raises(L.OwlLibError, match="OWL_ABI_VERSION"):
        L.load()


def test_product_package_never_imports_the_oracle():
    """Only tests/, selftest.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    import os
    import re
    pkg = os.path.dirname(os.path.abspath(__import__("owl_vit_object_detection_amd").__file__))
    offenders = []
    for fn in sorted(os.listdir(pkg)):
        if fn.endswith(".py") and fn != "selftest.py":
            src = open(os.path.join(pkg, fn)).read()
            if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                offenders.append(fn)
    assert offenders == []
