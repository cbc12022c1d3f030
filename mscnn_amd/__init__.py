"""mscnn_amd -- MI355X-native (gfx950) implementation of the MS-CNN detection hot path.

  csrc/            hand-written HIP kernels + the C ABI (include/mscnn_hip.h) -> libmscnn_hip.so
  host/            C++ mirror of the reference's caffe::Layer / Net interface on top of the C ABI
  hipapi.py        ctypes binding of the C ABI for Python callers (tests, bench)

The package never imports anything from oracle/ and has no CPU fallback.
"""
__version__ = "0.1.0"
