"""Raw device buffers for callers of the C ABI that hand over DEVICE pointers (bench.py's trait-loop leg, tests): hipMalloc / hipFree /
hipMemcpy of the HIP runtime itself through ctypes -- what a Rust caller would get from its own hip-sys binding.  No torch, no kernels."""
import ctypes as C

import numpy as np

_hip = None


def hip():
    global _hip
    if _hip is None:
        for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _hip = C.CDLL(name)
                break
            except OSError:
                continue
        if _hip is None:
            raise RuntimeError("libamdhip64.so not found")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        _hip.hipSetDevice.argtypes = [C.c_int]
    return _hip


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: hipError {rc}")


class DeviceBuffer:
    """`count` elements of `dtype` in device memory of `device`; .ptr is the raw address (an int, what Context methods accept)"""

    def __init__(self, count, dtype, device=0, zero=True):
        self.dtype, self.count = np.dtype(dtype), int(count)
        self.nbytes = self.count * self.dtype.itemsize
        _ok(hip().hipSetDevice(int(device)), "hipSetDevice")
        p = C.c_void_p()
        _ok(hip().hipMalloc(C.byref(p), max(self.nbytes, 1)), "hipMalloc")
        self.ptr = int(p.value)
        if zero and self.nbytes:
            _ok(hip().hipMemset(C.c_void_p(self.ptr), 0, self.nbytes), "hipMemset")

    def at(self, element_offset):
        """address of element `element_offset`"""
        return self.ptr + int(element_offset) * self.dtype.itemsize

    def to_host(self, shape=None):
        out = np.empty(self.count, dtype=self.dtype)
        if self.nbytes:
            _ok(hip().hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), self.nbytes, 2), "hipMemcpy D2H")
        return out.reshape(shape) if shape is not None else out

    def from_host(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype).reshape(-1)
        if a.size != self.count:
            raise ValueError(f"expected {self.count} elements, got {a.size}")
        if self.nbytes:
            _ok(hip().hipMemcpy(C.c_void_p(self.ptr), a.ctypes.data_as(C.c_void_p), self.nbytes, 1), "hipMemcpy H2D")

    def free(self):
        if getattr(self, "ptr", 0):
            hip().hipFree(C.c_void_p(self.ptr))
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:      # noqa: BLE001
            pass
