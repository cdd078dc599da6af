"""ctypes loader for ``libinstageo_hip.so`` (the C-ABI HIP library).

The signatures are parsed from ``include/instageo_hip.h`` so that the header is the single source of
truth for the boundary.  There is NO fallback: if the shared library is missing or a symbol cannot be
resolved this module raises, and every op in :mod:`instageo_amd.ops` fails loudly.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_PKG_ROOT = os.path.dirname(_HERE)
_REPO_ROOT = os.path.dirname(_PKG_ROOT)
# IG_HIP_LIB: another build of the same library (same header revision) -- same-box A/B runs of two builds (tools/); the default
# is the in-tree library next to this file
LIB_PATH = os.environ.get("IG_HIP_LIB") or os.path.join(_HERE, "libinstageo_hip.so")
HEADER_PATH = os.path.join(_REPO_ROOT, "include", "instageo_hip.h")

_SCALARS = {
    "int": ctypes.c_int,
    "long": ctypes.c_long,
    "unsigned": ctypes.c_uint,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
}


class HipLibraryError(RuntimeError):
    """Raised when the HIP library is missing or a call into it fails."""


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[str, List[str]]]:
    """Return {function: (return_type, [arg C types])} for every prototype in the header."""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos: Dict[str, Tuple[str, List[str]]] = {}
    for m in re.finditer(r"(const char\*|int)\s+(ig_\w+)\s*\(([^)]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        types: List[str] = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                # drop the parameter name (last identifier)
                t = re.sub(r"\s*\w+$", "", a) if not a.endswith("*") else a
                types.append(t.strip())
        protos[name] = (ret, types)
    return protos


def _ctype(t: str):
    if "*" in t:
        return ctypes.c_char_p if t.replace("const ", "") == "char*" else ctypes.c_void_p
    return _SCALARS[t]


_lib = None
_protos = None


def load():
    """Load the shared library (once) and attach argtypes/restype to every declared symbol."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {os.path.join(_PKG_ROOT, 'csrc')}`.  instageo_amd has no CPU fallback."
        )
    # PyTorch-ROCm bundles its own libamdhip64: it must be the HIP runtime already in the process when this library's
    # dependency is resolved, otherwise two runtimes coexist and launches here fail with "no ROCm-capable device" on
    # memory that torch allocated.  (torch is the plumbing for device memory and streams anyway.)
    import torch  # noqa: F401

    lib = ctypes.CDLL(LIB_PATH)
    protos = parse_header()
    for name, (ret, types) in protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise HipLibraryError(f"symbol {name} declared in instageo_hip.h is missing from {LIB_PATH}") from e
        fn.restype = ctypes.c_char_p if ret.startswith("const char") else ctypes.c_int
        fn.argtypes = [_ctype(t) for t in types]
    import hashlib

    want = int(hashlib.md5(open(HEADER_PATH, "rb").read()).hexdigest()[:7], 16)
    got = lib.ig_header_stamp()
    if got != want:
        raise HipLibraryError(
            f"{LIB_PATH} was built against another revision of {HEADER_PATH} (stamp {got:#x}, header {want:#x}): rebuild it "
            f"(`make -C {os.path.join(_PKG_ROOT, 'csrc')}`) -- calling it with this header's signatures would pass shifted arguments."
        )
    _lib, _protos = lib, protos
    return lib


def declared_symbols() -> List[str]:
    return sorted(parse_header().keys())


def last_error() -> str:
    return (load().ig_last_error() or b"").decode()


def call(name: str, *args) -> None:
    """Call an ``int ig_*`` entry point and raise on a non-zero status."""
    rc = getattr(load(), name)(*args)
    if rc != 0:
        raise HipLibraryError(f"{name} failed (rc={rc}): {last_error()}")
