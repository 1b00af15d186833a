"""ctypes binding of librerevst_hip.so (C ABI: include/rerevst_hip.h).

There is NO CPU fallback: if the shared library is missing or does not load, importing
the symbols raises.  (The numpy oracle under oracle/ is test infrastructure and is never
used from here.)
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# RRV_LIB_PATH: another build of the same library (A/B runs of kernel variants: `python -m rerevst-code_amd.build` with extra -D flags)
LIB_PATH = os.environ.get("RRV_LIB_PATH") or os.path.join(HERE, "librerevst_hip.so")
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
    "rrv_set_workspace_cap": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rrv_last_compute_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "rrv_get_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rrv_set_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rrv_comm_unique_id": (C.c_int, [C.c_char_p]),
    "rrv_comm_init_rank": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rrv_comm_destroy": (C.c_int, [C.c_void_p]),
    "rrv_broadcast_state": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "rrv_transfer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rrv_transfer_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_long)]),
    "rrv_transfer_wait": (C.c_int, [C.c_void_p, C.c_long]),
    "rrv_transfer_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rrv_transfer_batch_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rrv_transfer_blend_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "rrv_transfer_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rrv_transfer_blend": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "rrv_transfer_frames_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rrv_transfer_frames": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rrv_generate_content_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rrv_generate_content_features_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "rrv_add_patch": (C.c_int, [C.c_void_p, C.c_int]),
    "rrv_transfer_features": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_void_p]),
    "rrv_transfer_features_batch": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p]),
    "rrv_release_features": (C.c_int, [C.c_void_p]),
    "rrv_set_multistyle_group": (C.c_int, [C.c_void_p, C.c_int]),
    "rrv_set_f43": (C.c_int, [C.c_void_p, C.c_int]),
    "rrv_set_feature_cache_cap": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rrv_feature_cache_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "rrv_transfer_frame_mode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "rrv_get_preclamp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "rrv_get_preclamp_image": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "rrv_sync": (C.c_int, [C.c_void_p]),
    "rrv_set_pipeline": (C.c_int, [C.c_void_p, C.c_int]),
    "rrv_set_host_io": (C.c_int, [C.c_void_p, C.c_int]),
    "rrv_debug_copy_tensor": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rrv_set_grid_share": (C.c_int, [C.c_void_p, C.c_int]),
    "rrv_set_caller_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "rrv_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "rrv_host_free": (C.c_int, [C.c_void_p]),
    "rrv_host_register": (C.c_int, [C.c_void_p, C.c_size_t]),
    "rrv_host_unregister": (C.c_int, [C.c_void_p]),
    "rrv_set_debug": (C.c_int, [C.c_void_p, C.c_int]),
    "rrv_debug_selftest": (C.c_int, [C.c_void_p]),
    "rrv_debug_fail_alloc": (C.c_int, [C.c_void_p, C.c_int]),
    "rrv_profile_begin": (C.c_int, [C.c_void_p]),
    "rrv_profile_end": (C.c_int, [C.c_void_p]),
    "rrv_profile_count": (C.c_int, [C.c_void_p]),
    "rrv_profile_entry": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_float),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}


def _share_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so
    (SONAME libamdhip64.so.7, NEEDED by torch as "libamdhip64.so"); if our library pulled
    /opt/rocm's copy in first, a later `import torch` would load a SECOND runtime that sees
    no GPU.  Pre-loading torch's copy (when torch is installed) makes both bind to it,
    whatever the import order.  RRV_SYSTEM_HIP=1 skips this."""
    if os.environ.get("RRV_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
            cand = os.path.join(libdir, "libamdhip64.so")
            if os.path.exists(cand):
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            # rrv_broadcast_state dlopens RCCL lazily: point it at the copy built against the same HIP runtime
            rccl = os.path.join(libdir, "librccl.so")
            if os.path.exists(rccl):
                os.environ.setdefault("RRV_RCCL_PATH", rccl)
    except Exception:
        pass


def load():
    """Load the HIP library (once).  Raises OSError with a build hint when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not found: build it with `python __graft_entry__.py build` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    _share_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)     # AttributeError if the .so lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib
