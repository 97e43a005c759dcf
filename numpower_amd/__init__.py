"""numpower_amd — MI355X-native back end for NumPower's fp32 NDArray hot path.

Layers (bottom up):
  csrc/ + include/np_hip.h   hand-written HIP kernels for gfx950 behind a C ABI (libnp_hip.so)
  host/ + include/numpower_host.h
                             C++ host mirror of the reference's NDArray L2/L3 entry points
                             (NDArray_Add_Float, reduce, NDArray_Matmul, NDArray_ToGPU ...)
  ndarray.py                 Python stand-in for the PHP `NDArray` class surface (tests, bench)
  device.py / _lib.py        ctypes access to the C ABI
"""
__version__ = "0.1.0"
