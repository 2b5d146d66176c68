"""DLPack producer for DeviceArray in plain ctypes (dlpack.h v0.8 structs; no torch / cupy import).

`make_capsule(arr)` -> PyCapsule "dltensor" around a DLManagedTensor whose `data` is the HIP device pointer, device =
(kDLROCM, device id), C-contiguous (strides NULL).  The managed tensor, its shape array and a reference to the DeviceArray
live in `_live` until the consumer calls the deleter (or, if nobody consumed the capsule, until the capsule is collected)."""
import ctypes as C

import numpy as np

kDLCPU, kDLCUDA, kDLROCM = 1, 2, 10
kDLInt, kDLUInt, kDLFloat = 0, 1, 2


class DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", DLDevice), ("ndim", C.c_int32), ("dtype", DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(DLManagedTensor))
DLManagedTensor._fields_ = [("dl_tensor", DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DELETER)]

_live = {}          # address of the DLManagedTensor -> (managed tensor, shape array, DeviceArray)
_NAME, _USED = b"dltensor", b"used_dltensor"


@_DELETER
def _deleter(mt_ptr):
    _live.pop(C.addressof(mt_ptr.contents), None)


_CAPSULE_DTOR = C.CFUNCTYPE(None, C.c_void_p)


@_CAPSULE_DTOR
def _capsule_destructor(capsule):
    # a capsule nobody consumed still owns its tensor (a consumed one was renamed "used_dltensor" and its consumer calls the deleter)
    api = C.pythonapi
    if api.PyCapsule_IsValid(C.c_void_p(capsule), _NAME):
        addr = api.PyCapsule_GetPointer(C.c_void_p(capsule), _NAME)
        _live.pop(addr, None)


def _api():
    api = C.pythonapi
    api.PyCapsule_New.restype = C.py_object
    api.PyCapsule_New.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
    api.PyCapsule_IsValid.restype = C.c_int
    api.PyCapsule_IsValid.argtypes = [C.c_void_p, C.c_char_p]
    api.PyCapsule_GetPointer.restype = C.c_void_p
    api.PyCapsule_GetPointer.argtypes = [C.c_void_p, C.c_char_p]
    return api


def dl_dtype(dtype):
    dtype = np.dtype(dtype)
    code = {"f": kDLFloat, "i": kDLInt, "u": kDLUInt}.get(dtype.kind)
    if code is None:
        raise TypeError("no DLPack type for %s" % dtype)
    return DLDataType(code, dtype.itemsize * 8, 1)


def make_capsule(arr):
    mt = DLManagedTensor()
    shape = (C.c_int64 * len(arr.shape))(*arr.shape)
    t = mt.dl_tensor
    t.data = C.c_void_p(int(arr.ptr))
    t.device = DLDevice(kDLROCM, int(getattr(arr.sim, "device_id", 0)))
    t.ndim = len(arr.shape)
    t.dtype = dl_dtype(arr.dtype)
    t.shape = C.cast(shape, C.POINTER(C.c_int64))
    t.strides = None
    t.byte_offset = 0
    mt.manager_ctx = None
    mt.deleter = _deleter
    addr = C.addressof(mt)
    _live[addr] = (mt, shape, arr)
    return _api().PyCapsule_New(addr, _NAME, C.cast(_capsule_destructor, C.c_void_p))


def exports_of(sim):
    """how many DLPack exports of `sim`'s buffers are still held by a consumer (their deleter has not run)"""
    return sum(1 for (_, _, arr) in list(_live.values()) if getattr(arr, "sim", None) is sim)


def read_capsule(capsule):
    """(tests) the fields of a not-yet-consumed "dltensor" capsule"""
    api = _api()
    api.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
    addr = api.PyCapsule_GetPointer(capsule, _NAME)
    api.PyCapsule_GetPointer.argtypes = [C.c_void_p, C.c_char_p]
    mt = DLManagedTensor.from_address(addr)
    t = mt.dl_tensor
    return {"data": t.data, "device": (t.device.device_type, t.device.device_id), "ndim": t.ndim,
            "dtype": (t.dtype.code, t.dtype.bits, t.dtype.lanes), "shape": tuple(t.shape[i] for i in range(t.ndim)),
            "strides": None if not t.strides else tuple(t.strides[i] for i in range(t.ndim)), "byte_offset": t.byte_offset,
            "address": addr}
