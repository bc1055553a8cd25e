"""ctypes binding of libowlhip.so -- the C ABI declared in include/owl_hip.h.

The argtypes are parsed from the header itself so the binding cannot drift from it.  There is NO
fallback: if the shared library is missing or a symbol is absent this module raises, and every op
in the package fails loudly (the product path never routes through the oracle or any CPU path).
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
HEADER = os.path.join(_ROOT, "include", "owl_hip.h")
TUNING_HEADER = os.path.join(_ROOT, "include", "owl_hip_tuning.h")      # bound only when OWL_TUNING=1 (tools/, tuning builds)
LIB_PATH = os.path.join(_PKG, "libowlhip.so")

_CTYPES = {
    "void*": ctypes.c_void_p, "const void*": ctypes.c_void_p,
    "float*": ctypes.c_void_p, "const float*": ctypes.c_void_p,
    "double*": ctypes.c_void_p, "const double*": ctypes.c_void_p,
    "unsigned char*": ctypes.c_void_p, "const unsigned char*": ctypes.c_void_p,
    "int*": ctypes.c_void_p, "const int*": ctypes.c_void_p,
    "int64_t*": ctypes.c_void_p, "const int64_t*": ctypes.c_void_p,
    "int64_t": ctypes.c_int64, "int": ctypes.c_int, "float": ctypes.c_float, "double": ctypes.c_double,
}
_RET = {"int": ctypes.c_int, "const char*": ctypes.c_char_p}


def header_abi_version(path: str = None) -> int:
    """`#define OWL_ABI_VERSION n` of the header the signatures below are generated from."""
    path = path or HEADER
    m = re.search(r"^\s*#\s*define\s+OWL_ABI_VERSION\s+(\d+)", open(path).read(), flags=re.M)
    if not m:
        raise OwlLibError(f"{path} does not define OWL_ABI_VERSION")
    return int(m.group(1))


def parse_header(path: str = None):
    """-> {name: (restype, [(ctype_name, arg_name), ...])} for every prototype in the header."""
    src = open(path or HEADER).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(owl_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                ty = mm.group(1).strip().replace(" *", "*")
                parsed.append((ty, mm.group(2)))
        protos[name] = (ret, parsed)
    return protos


class OwlLibError(RuntimeError):
    pass


_lib = None
_protos = None
_TUNING_PROBE = "owl_gemm_set_persistent"          # exported by OWL_TUNING builds only (csrc/gemm.hip)


def is_tuning_build() -> bool:
    """True if the loaded libowlhip.so is an OWL_TUNING build (tools/ experiments); the shipped library is not."""
    return hasattr(load(), _TUNING_PROBE)


def load():
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OwlLibError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or owl-vit-object-detection_amd/csrc/build.sh). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    # The tuning entries (include/owl_hip_tuning.h) are bound when the LOADED library is a tuning build -- decided from what it exports, not from the
    # environment (OWL_TUNING=1 beside a shipped build used to die on a missing symbol, the reverse refused calls the library has)
    tuning = hasattr(lib, _TUNING_PROBE)
    if os.environ.get("OWL_TUNING", "0") == "1" and not tuning:
        raise OwlLibError(f"OWL_TUNING=1 but {LIB_PATH} is the shipped build (no `{_TUNING_PROBE}`): rebuild it with OWL_TUNING=1 bash csrc/build.sh, or unset OWL_TUNING")
    if tuning:
        _protos.update(parse_header(TUNING_HEADER))
    for name, (ret, args) in _protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise OwlLibError(f"libowlhip.so does not export `{name}` declared in include/owl_hip.h") from e
        fn.restype = _RET[ret]
        fn.argtypes = [_CTYPES[t] for t, _ in args]
    # a stale libowlhip.so behind a newer header (or the reverse) would take garbage trailing arguments or an undersized scratch buffer
    # without any diagnostic: refuse it here
    built, want = int(lib.owl_abi_version()), header_abi_version()
    if built != want:
        raise OwlLibError(f"{LIB_PATH} was built with OWL_ABI_VERSION {built} but include/owl_hip.h declares {want}: rebuild it "
                          "(owl-vit-object-detection_amd/csrc/build.sh)")
    _lib = lib
    return lib


def protos():
    load()
    return _protos


def last_error() -> str:
    return load().owl_last_error().decode()


def call(name: str, *args):
    """Invoke a C-ABI entry; torch tensors are passed as device pointers; raises on rc != 0."""
    lib = load()
    fn = getattr(lib, name)
    conv = []
    for a in args:
        if a is None:
            conv.append(None)
        elif hasattr(a, "data_ptr"):
            conv.append(a.data_ptr())
        else:
            conv.append(a)
    rc = fn(*conv)
    if rc != 0:
        raise OwlLibError(f"{name} failed (rc={rc}): {last_error()}")
    return rc
