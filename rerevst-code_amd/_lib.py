"""ctypes binding of librerevst_hip.so (C ABI: include/rerevst_hip.h).

There is NO CPU fallback: if the shared library is missing or does not load, importing
the symbols raises.  (The numpy oracle under oracle/ is test infrastructure and is never
used from here.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "librerevst_hip.so")
STATE_FLOATS = 17536
MAX_STYLES = 8

_lib = None

# name -> (restype, argtypes); must list every symbol declared in include/rerevst_hip.h
SYMBOLS = {
    "rrv_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rrv_destroy": (C.c_int, [C.c_void_p]),
    "rrv_last_error": (C.c_char_p, [C.c_void_p]),
    "rrv_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "rrv_finalize_weights": (C.c_int, [C.c_void_p]),
    "rrv_prepare_style": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "rrv_clean": (C.c_int, [C.c_void_p]),
    "rrv_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rrv_compute": (C.c_int, [C.c_void_p]),
    "rrv_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rrv_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rrv_transfer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rrv_transfer_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rrv_transfer_blend_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "rrv_get_preclamp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rrv_sync": (C.c_int, [C.c_void_p]),
    "rrv_profile_begin": (C.c_int, [C.c_void_p]),
    "rrv_profile_end": (C.c_int, [C.c_void_p]),
    "rrv_profile_count": (C.c_int, [C.c_void_p]),
    "rrv_profile_entry": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}


def load():
    """Load the HIP library (once).  Raises OSError with a build hint when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not found: build it with `python __graft_entry__.py build` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib
