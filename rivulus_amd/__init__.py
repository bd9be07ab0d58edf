"""rivulus_amd -- MI355X (gfx950) execution backend for the Rivulus filter/project/scan hot path.

The product is `csrc/librivulus_gpu.so` (C ABI: include/rivulus_gpu.h) and the C++ host
layer built on it; `rivulus_amd.capi` is the ctypes driver used by tests and bench.py.
"""
__all__ = ["capi"]
