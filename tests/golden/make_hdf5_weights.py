"""Writes Caffe-style HDF5 weight snapshots with the REAL HDF5 library (libhdf5 + libhdf5_hl through ctypes: H5Fcreate,
H5Gcreate2, H5LTmake_dataset_{float,double,int} -- the calls Net::ToHDF5 / hdf5_save_nd_dataset make, net.cpp:868-918,
util/hdf5.cpp:95-142), so that the product's own HDF5 subset reader (mscnn_amd/host/src/hdf5_lite.cpp) is checked against
files it did not write.  The library only exists in the build container (/opt/conda/lib); the small fixture
tests/golden/weights_small.h5 is committed, tests regenerate larger ones when the library is present.

  python tests/golden/make_hdf5_weights.py        # rewrites tests/golden/weights_small.h5
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

_CANDIDATES = [("/opt/conda/lib/libhdf5.so.103", "/opt/conda/lib/libhdf5_hl.so.100"),
               ("/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5_hl.so")]
_libs = None


def libs():
    global _libs
    if _libs is None:
        pairs = list(_CANDIDATES)
        a, b = ctypes.util.find_library("hdf5"), ctypes.util.find_library("hdf5_hl")
        if a and b:
            pairs.append((a, b))
        for core, hl in pairs:
            try:
                h5 = C.CDLL(core, mode=C.RTLD_GLOBAL)
                h5l = C.CDLL(hl, mode=C.RTLD_GLOBAL)
            except OSError:
                continue
            hid = C.c_int64                     # hid_t is int64_t since HDF5 1.10
            maj, mnr, rel = C.c_uint(), C.c_uint(), C.c_uint()
            h5.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
            if (maj.value, mnr.value) < (1, 10):
                hid = C.c_int
            h5.H5open()
            h5.H5Fcreate.restype = hid; h5.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, hid, hid]
            h5.H5Gcreate2.restype = hid; h5.H5Gcreate2.argtypes = [hid, C.c_char_p, hid, hid, hid]
            h5.H5Gclose.argtypes = [hid]; h5.H5Fclose.argtypes = [hid]
            for fn in ("H5LTmake_dataset_float", "H5LTmake_dataset_double", "H5LTmake_dataset_int"):
                getattr(h5l, fn).argtypes = [hid, C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.c_void_p]
            _libs = (h5, h5l, (maj.value, mnr.value, rel.value))
            break
        else:
            _libs = False
    return _libs


def available():
    return bool(libs())


def write(path, layers):
    """layers: [(layer_name, [array, ...])]; dtype float32 / float64 / int32 picks the H5LTmake_dataset_* flavour."""
    h5, h5l, _ = libs()
    H5F_ACC_TRUNC = 2
    f = h5.H5Fcreate(os.fsencode(str(path)), H5F_ACC_TRUNC, 0, 0)
    assert f >= 0, "H5Fcreate"
    g = h5.H5Gcreate2(f, b"data", 0, 0, 0)          # net.cpp:874
    assert g >= 0
    for name, blobs in layers:
        lg = h5.H5Gcreate2(g, name.encode(), 0, 0, 0)
        assert lg >= 0, name
        for j, a in enumerate(blobs):
            a = np.ascontiguousarray(a)
            dims = (C.c_uint64 * max(a.ndim, 1))(*a.shape)
            fn = {np.dtype(np.float32): h5l.H5LTmake_dataset_float, np.dtype(np.float64): h5l.H5LTmake_dataset_double,
                  np.dtype(np.int32): h5l.H5LTmake_dataset_int}[a.dtype]
            assert fn(lg, str(j).encode(), a.ndim, dims, a.ctypes.data_as(C.c_void_p)) >= 0, (name, j)
        h5.H5Gclose(lg)
    h5.H5Gclose(g)
    h5.H5Fclose(f)


def small_fixture():
    """Layers of kitti_car/mscnn-7s-576 small enough to commit: float, double and int datasets, and 24 extra groups so that the
    "data" group's B-tree has several symbol-table leaves (and names the reader must ignore)."""
    rng = np.random.default_rng(20260925)
    layers = [("conv1_1", [rng.standard_normal((64, 3, 3, 3)).astype(np.float32), rng.standard_normal(64).astype(np.float32)]),
              ("cls_pred", [rng.standard_normal((5, 4096)).astype(np.float64), rng.standard_normal(5).astype(np.float64)]),
              ("conv1_2", [rng.standard_normal((64, 64, 3, 3)).astype(np.float32), rng.integers(-5, 6, 64).astype(np.int32)])]
    for k in range(24):
        layers.append((f"not_in_the_net_{k:02d}", [rng.standard_normal((2, 3)).astype(np.float32)]))
    return layers


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    assert available(), "libhdf5 not found"
    write(os.path.join(here, "weights_small.h5"), small_fixture())
    print("wrote weights_small.h5 with HDF5", ".".join(map(str, libs()[2])), os.path.getsize(os.path.join(here, "weights_small.h5")), "bytes")
