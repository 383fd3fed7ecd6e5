"""ctypes binding of liboei_hip.so (the C ABI declared in include/oei_hip.h).

The library is the product path: there is no CPU fallback.  If the shared object is missing or
does not export the declared ABI, importing this module raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('OEI_LIB_OVERRIDE') or os.path.join(_HERE, 'liboei_hip.so')   # override: diagnostic builds only
ABI_VERSION = 1

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int_p = ctypes.POINTER(ctypes.c_int)
vp = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/oei_hip.h declares
SIGNATURES = {
    'oei_last_error': (ctypes.c_char_p, []),
    'oei_abi_version': (ctypes.c_int, []),
    'oei_model_create': (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, vp, vp,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_double, vp,
                                        ctypes.c_int, ctypes.c_double]),
    'oei_model_destroy': (ctypes.c_int, [vp]),
    'oei_model_refactor': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, vp, ctypes.c_int, ctypes.c_double]),
    'oei_model_info': (ctypes.c_int, [vp, c_double_p, c_int_p, c_int_p]),
    'oei_model_get_factor': (ctypes.c_int, [vp, vp]),
    'oei_predict_f': (ctypes.c_int, [vp, ctypes.c_int, vp, vp, vp, vp]),
    'oei_omega_batched': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    'oei_sdp_batched': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    'oei_solution_derivative_batched': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                       vp, vp, vp, vp, vp, vp, vp, vp]),
    'oei_value_grad_batched': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp]),
    'oei_value_grad_warm_batched': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'oei_value_grad_batched_noalloc': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp]),
    'oei_hessian_batched': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]),
    'oei_project_flags': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    'oei_domega_batched': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    'oei_set_profiling': (ctypes.c_int, [vp, ctypes.c_int]),
    'oei_get_stage_times': (ctypes.c_int, [vp, vp]),
    'oei_get_solver_kernel_time': (ctypes.c_int, [vp, vp]),
    'oei_set_pipelining': (ctypes.c_int, [vp, ctypes.c_int]),
    'oei_set_z_free': (ctypes.c_int, [vp, ctypes.c_int]),
    'oei_argmin': (ctypes.c_int, [ctypes.c_int, vp, ctypes.c_longlong, vp, vp, vp]),
    'oei_best_record': (ctypes.c_int, [ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_longlong, vp, vp]),
    'oei_select_record': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, vp, vp, vp]),
    'oei_fit_create': (ctypes.c_int, [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    'oei_fit_destroy': (ctypes.c_int, [vp]),
    'oei_fit_nlml_batched': (ctypes.c_int, [vp, ctypes.c_int, vp, ctypes.c_int, ctypes.c_double, ctypes.c_int, vp, vp, vp]),
    'oei_lbfgs_layout': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                        ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t),
                                        ctypes.POINTER(ctypes.c_size_t)]),
    'oei_lbfgs_init': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp]),
    'oei_lbfgs_advance': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, vp,
                                         ctypes.POINTER(ctypes.c_int), vp]),
    'oei_newton_layout': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t),
                                         ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t),
                                         ctypes.POINTER(ctypes.c_size_t)]),
    'oei_newton_init': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, vp, vp]),
    'oei_newton_advance': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_double,
                                          ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, vp,
                                          ctypes.POINTER(ctypes.c_int), vp]),
    'oei_multistart_workspace': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_size_t)]),
    'oei_lbfgs_minimize_rounds': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                 ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int,
                                                 ctypes.c_int, ctypes.POINTER(ctypes.c_int), vp]),
    'oei_newton_minimize_rounds': (ctypes.c_int, [vp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double,
                                                  ctypes.c_double, ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_int,
                                                  ctypes.c_int, ctypes.POINTER(ctypes.c_int), vp]),
    'oei_model_nlml': (ctypes.c_int, [vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_int]),
}


class OeiError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'liboei_hip.so not found at %s: build it with `make -C %s` (hipcc, gfx950). '
            'There is no CPU fallback.' % (LIB_PATH, os.path.join(_HERE, 'csrc')))
    # torch (which owns device memory and streams for this package) must have brought up its HIP
    # runtime before this library's code objects register: loaded in the other order, the
    # library's first runtime call reports that no device is present.
    import torch
    if torch.cuda.is_available():
        torch.cuda.init()
        torch.empty(1, device='cuda')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.oei_abi_version() != ABI_VERSION:
        raise ImportError('liboei_hip.so ABI %d != expected %d' % (lib.oei_abi_version(), ABI_VERSION))
    return lib


lib = _load()


def check(rc, what):
    if rc != 0:
        raise OeiError('%s failed (%d): %s' % (what, rc, lib.oei_last_error().decode()))
