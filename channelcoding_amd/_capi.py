"""ctypes binding of libchannelcoding_amd.so (include/channelcoding_amd.h).

Plumbing only.  The library is built in-tree by ``__graft_entry__.build()`` /
``make -C channelcoding_amd/csrc``; if it is missing this module raises -- there
is no Python or CPU fallback for any decode path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHANNELCODING_AMD_LIB: another build of the same library (kernel experiments under profiles/); never a fallback
LIB_PATH = os.environ.get("CHANNELCODING_AMD_LIB") or os.path.join(_HERE, "libchannelcoding_amd.so")

# enums of channelcoding_amd.h
OK, ERR_INVALID_ARGUMENT, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_LENGTH, ERR_NOT_IN_FIELD = range(8)
FAMILY_BCH, FAMILY_RS = 0, 1
ALG_PGZ, ALG_BM, ALG_EUKLID = 0, 1, 2
ALG_MS, ALG_NMS, ALG_OMS, ALG_SCMS1, ALG_SCMS2, ALG_2DNMS = 16, 17, 18, 19, 20, 21
CODING_DIVISION, CODING_MULTIPLICATION = 0, 1
STOP_AS_SHIPPED, STOP_PUBLISHED, STOP_PARITY = 0, 1, 2
DEVICE_CURRENT, DEVICE_NONE = -1, -2
FRAME_OK, FRAME_NOT_CONVERGED, FRAME_LOCATOR, FRAME_RECHECK, FRAME_ERASURES = range(5)
MC_FRAMES, MC_WORD_ERRORS, MC_BIT_ERRORS, MC_FAILURES, MC_UNDETECTED, MC_ITER_SUM, MC_CHANNEL_BIT_ERRORS = range(7)
MC_ITER_HIST, MC_NCOUNTERS = 8, 64

SOFT_ALGS = (ALG_MS, ALG_NMS, ALG_OMS, ALG_SCMS1, ALG_SCMS2, ALG_2DNMS)
HARD_ALGS = (ALG_PGZ, ALG_BM, ALG_EUKLID)


class Desc(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("family", C.c_int32), ("q", C.c_uint32), ("t", C.c_uint32), ("n", C.c_uint32),
        ("mu", C.c_uint32), ("step", C.c_uint32), ("coding", C.c_int32), ("algorithm", C.c_int32),
        ("iterations", C.c_uint32), ("alpha", C.c_double), ("beta", C.c_double), ("stop_rule", C.c_int32),
        ("device", C.c_int32), ("modular_polynomial", C.c_uint32), ("reserved", C.c_uint32),
    ]


class CcError(RuntimeError):
    def __init__(self, status, where):
        self.status = status
        msg = lib().cc_status_string(status).decode()
        detail = lib().cc_last_error().decode()
        super().__init__("%s: %s%s" % (where, msg, (" (" + detail + ")") if detail else ""))


_lib = None
_VP = C.c_void_p
_SIGNATURES = {
    "cc_version": (C.c_char_p, []),
    "cc_status_string": (C.c_char_p, [C.c_int]),
    "cc_last_error": (C.c_char_p, []),
    "cc_code_create": (C.c_int, [C.POINTER(Desc), C.POINTER(_VP)]),
    "cc_code_destroy": (None, [_VP]),
    "cc_desc_init": (None, [C.POINTER(Desc)]),
    "cc_n": (C.c_uint32, [_VP]), "cc_k": (C.c_uint32, [_VP]), "cc_l": (C.c_uint32, [_VP]),
    "cc_t": (C.c_uint32, [_VP]), "cc_dmin": (C.c_uint32, [_VP]), "cc_rate": (C.c_double, [_VP]),
    "cc_to_string": (C.c_int, [_VP, C.c_char_p, C.c_size_t]),
    "cc_get_poly": (C.c_int, [_VP, C.c_int, _VP, C.c_size_t]),
    "cc_get_H": (C.c_int, [_VP, _VP]),
    "cc_get_H_alt": (C.c_int, [_VP, _VP, C.POINTER(C.c_uint32)]),
    "cc_code_create_with_H": (C.c_int, [C.POINTER(Desc), _VP, C.c_uint32, C.POINTER(_VP)]),
    "cc_minsum_create": (C.c_int, [C.POINTER(Desc), _VP, C.c_uint32, C.c_uint32, C.POINTER(_VP)]),
    "cc_encode_batch": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "cc_encode_batch_dev": (C.c_int, [_VP, _VP, _VP, C.c_size_t, _VP]),
    "cc_correct_hard_batch": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t]),
    "cc_correct_hard_batch_dev": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP]),
    "cc_correct_hard_f32_batch": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t]),
    "cc_correct_hard_f32_batch_dev": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP]),
    "cc_correct_soft_batch": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t]),
    "cc_correct_soft_batch_dev": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP]),
    "cc_extract_batch": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "cc_extract_batch_dev": (C.c_int, [_VP, _VP, _VP, C.c_size_t, _VP]),
    "cc_decode_hard_batch": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t]),
    "cc_decode_soft_batch": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t]),
    "cc_mc_run_dev": (C.c_int, [_VP, C.c_double, C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, _VP, _VP]),
    "cc_awgn_llr_dev": (C.c_int, [_VP, C.c_double, C.c_uint64, C.c_uint64, C.c_size_t, C.c_int, _VP, _VP, _VP]),
    "cc_sigma": (C.c_double, [_VP, C.c_double]),
    "cc_encode_batch_u16": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "cc_encode_batch_u16_dev": (C.c_int, [_VP, _VP, _VP, C.c_size_t, _VP]),
    "cc_correct_hard_batch_u16": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t]),
    "cc_correct_hard_batch_u16_dev": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP]),
    "cc_extract_batch_u16": (C.c_int, [_VP, _VP, _VP, C.c_size_t]),
    "cc_extract_batch_u16_dev": (C.c_int, [_VP, _VP, _VP, C.c_size_t, _VP]),
    "cc_get_poly_u16": (C.c_int, [_VP, C.c_int, _VP, C.c_size_t]),
    "cc_q": (C.c_uint32, [_VP]),
    "cc_diag_table": (C.c_int, [_VP, _VP, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                C.POINTER(C.c_uint32)]),
    "cc_kernel_info": (C.c_int, [_VP, C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                 C.POINTER(C.c_uint32)]),
}


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    """Loads the HIP library; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "channelcoding_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C channelcoding_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        # torch wheels bundle their own libamdhip64.so; two HIP runtimes cannot share one process
        # ("No HIP GPUs are available" for whichever initialises second).  Loading torch first makes our
        # library bind to the runtime torch already mapped, so both see the same devices and streams.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(status, where):
    if status != OK:
        raise CcError(status, where)
