"""ctypes binding of libsgs_hip.so.  Prototypes are parsed from include/sgs_hip.h so the
binding cannot drift from the declared C ABI.  There is NO fallback: if the library is
missing or a symbol is absent, importing / calling fails loudly."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "sgs_hip.h")
LIB_PATH = os.environ.get("SGS_LIB_PATH") or os.path.join(_HERE, "libsgs_hip.so")

_SCALARS = {
    "int": ctypes.c_int, "int64_t": ctypes.c_int64, "uint64_t": ctypes.c_uint64, "uint32_t": ctypes.c_uint32,
    "int32_t": ctypes.c_int32, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
    "sgs_stream_t": ctypes.c_void_p,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every `sgs_*` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int|size_t|void|const\s+char\s*\*)\s+(sgs_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        restype = {"int": ctypes.c_int, "int64_t": ctypes.c_int64, "size_t": ctypes.c_size_t, "void": None}.get(ret.strip(), ctypes.c_char_p)
        argtypes, argnames = [], []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                a = re.sub(r"/\*.*?\*/", "", a).strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    argnames.append(a.split("*")[-1].strip())
                else:
                    toks = a.replace("const ", "").split()
                    argtypes.append(_SCALARS[toks[0]])
                    argnames.append(toks[-1])
        protos[name] = (restype, argtypes, argnames)
    return protos


class SgsLibraryMissing(RuntimeError):
    pass


_lib = None
_protos = None


def lib():
    """The loaded library with typed entry points; raises if it has not been built."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SgsLibraryMissing(
            f"{LIB_PATH} not found: build it with `python sgs-gnn_amd/build.py` "
            "(there is no CPU or PyTorch fallback for the SGS hot path)")
    L = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, (restype, argtypes, _) in _protos.items():
        fn = getattr(L, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if L.sgs_abi_version() != 1:
        raise SgsLibraryMissing(f"{LIB_PATH}: ABI version {L.sgs_abi_version()} != 1")
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        msg = lib().sgs_last_error()
        raise RuntimeError(f"libsgs_hip: {what} failed ({rc}): {msg.decode() if msg else ''}")
