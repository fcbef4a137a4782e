"""ctypes binding of libkalle_hip.so.

The prototypes are parsed from include/kalle_hip.h so the header stays the single source of truth for
the C-ABI.  There is deliberately NO fallback: if the shared library is missing or a symbol cannot be
resolved the import of the product path fails loudly (run `python -m kalle_audio_amd.build`).
"""
import ctypes
import os
import re

# torch must be imported BEFORE libkalle_hip.so is dlopen'ed: the library's DT_NEEDED libamdhip64.so then resolves to
# the HIP runtime torch already loaded, so both share one runtime (and one view of the device, streams and memory).
# Loaded the other way round the process ends up with two HIP runtimes and ours reports hipErrorNoDevice.
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(HERE, "..", "include", "kalle_hip.h")
LIB_PATH = os.environ.get("KALLE_LIB_PATH") or os.path.join(HERE, "libkalle_hip.so")   # (override: A/B two builds on one box)

KALLE_BF16 = 0
KALLE_F32 = 1


class GemmEpilogue(ctypes.Structure):
    """Mirror of `kalle_gemm_epilogue` (include/kalle_hip.h)."""
    _fields_ = [
        ("bias", ctypes.c_void_p),
        ("gate", ctypes.c_void_p),
        ("ldg", ctypes.c_int64),
        ("rows_per_batch", ctypes.c_int32),
        ("residual", ctypes.c_void_p),
        ("ldr", ctypes.c_int64),
        ("accumulate", ctypes.c_int32),
        ("alpha", ctypes.c_float),
        ("c_rows_per_batch", ctypes.c_int32),
        ("c_batch_rows", ctypes.c_int32),
        ("c_row_offset", ctypes.c_int32),
        ("row_mask", ctypes.c_void_p),
        ("glu_mode", ctypes.c_int32),
        ("glu_inner", ctypes.c_int32),
        ("glu_aux", ctypes.c_void_p),
        ("glu_dbias", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p),
        ("workspace_bytes", ctypes.c_int64),
    ]


class Act(ctypes.Structure):
    """Mirror of `kalle_act`."""
    _fields_ = [("code", ctypes.c_int32), ("logscale", ctypes.c_int32), ("alpha", ctypes.c_void_p),
                ("beta", ctypes.c_void_p), ("param", ctypes.c_float)]


class ConvEpilogue(ctypes.Structure):
    """Mirror of `kalle_conv_epilogue`."""
    _fields_ = [("residual", ctypes.c_void_p), ("out_scale", ctypes.c_float), ("accumulate", ctypes.c_int32),
                ("tanh", ctypes.c_int32), ("post_act", Act), ("y_raw", ctypes.c_void_p)]


class LlamaLayer(ctypes.Structure):
    """Mirror of `kalle_llama_layer`."""
    _fields_ = [("input_norm", ctypes.c_void_p), ("wqkv", ctypes.c_void_p), ("wo", ctypes.c_void_p),
                ("post_norm", ctypes.c_void_p), ("wug", ctypes.c_void_p), ("wdown", ctypes.c_void_p),
                ("kv_cache", ctypes.c_void_p)]


class WgradProblem(ctypes.Structure):
    """Mirror of `kalle_wgrad_problem`."""
    _fields_ = [("dy", ctypes.c_void_p), ("lddy", ctypes.c_int64), ("x", ctypes.c_void_p), ("ldx", ctypes.c_int64),
                ("dw", ctypes.c_void_p), ("lddw", ctypes.c_int64), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("tokens", ctypes.c_int32)]


_CTYPE = {
    "int": ctypes.c_int,
    "int32_t": ctypes.c_int32,
    "int64_t": ctypes.c_int64,
    "float": ctypes.c_float,
}


def parse_header(path=HEADER):
    """Return {name: (restype, [argtypes])} for every `kalle_*` function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(kalle_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        argtypes = []
        args = " ".join(args.split())
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    argtypes.append(_CTYPE[a.split()[-2] if len(a.split()) > 1 else a])
        protos[name] = (ctypes.c_char_p if ret.startswith("const char") else ctypes.c_int, argtypes)
    return protos


class KalleError(RuntimeError):
    pass


_ERR = {-1: "KALLE_ERR_ARG (bad shape/alignment/null pointer)", -2: "KALLE_ERR_LAUNCH (HIP launch failed)",
        -3: "KALLE_ERR_UNSUPPORTED"}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KalleError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU/torch fallback exists). "
            "Build it with `python -m kalle_audio_amd.build`.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in parse_header().items():
        fn = getattr(lib, name)  # AttributeError -> loud failure on a missing symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


_TRACE = os.environ.get("KALLE_TRACE")      # debugging aid: name every C-ABI call on stderr and wait for it to finish


def check(code, what):
    if _TRACE:
        import sys
        print("[kalle] " + what, file=sys.stderr, flush=True)
        torch.cuda.synchronize()
    if code != 0:
        detail = ""
        if code == -2 and _lib is not None:
            detail = " [" + (_lib.kalle_last_error() or b"").decode() + "]"
        raise KalleError(f"{what} failed: {_ERR.get(code, code)}{detail}")
