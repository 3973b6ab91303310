"""ctypes binding of libasr_hip.so (the C ABI declared in include/asr_hip.h).

There is no CPU fallback: importing this module without the built library raises, and every
wrapper refuses non-CUDA tensors.  PyTorch is used for device memory and streams only.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'lib', 'libasr_hip.so')

F32, BF16 = 0, 1
ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2

_vp, _i, _l, _f, _u64 = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    'asr_gemm': [_vp, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _i, _i, _i, _i, _i, _l, _l, _l, _i, _i, _i, _vp],
}
_RESTYPES = {
    'asr_last_error': ctypes.c_char_p,
    'asr_device_arch': ctypes.c_char_p,
    'asr_version': ctypes.c_int,
}


class HipLibraryMissing(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            'libasr_hip.so not found at %s — build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(there is no CPU fallback for the HIP path)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name, rt in _RESTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = []
        fn.restype = rt
    return lib


_lib = _load()


def lib():
    return _lib


def exported_symbols():
    return list(SIGNATURES.keys()) + list(_RESTYPES.keys())


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('HIP path got a non-CUDA tensor; there is no CPU fallback')
    return ctypes.c_void_p(t.data_ptr())


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what, rc, _lib.asr_last_error().decode()))


def call(name, *args):
    check(getattr(_lib, name)(*args), name)


def _f32c(t):
    assert t.dtype == torch.float32, t.dtype
    return t


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_kc=1, b_kc=1, bias=None, act=ACT_NONE, accum=0, splits=1,
         batch=1, sA=0, sB=0, sC=0, seqT=0, bshift=0, prec=BF16):
    """Raw contraction on (views of) fp32 CUDA tensors; see include/asr_hip.h::asr_gemm."""
    call('asr_gemm', ptr(_f32c(A)), ptr(_f32c(B)), ptr(_f32c(C)), ptr(bias), M, N, K, lda, ldb, ldc,
         a_kc, b_kc, act, accum, splits, batch, sA, sB, sC, seqT, bshift, prec, stream_ptr())
