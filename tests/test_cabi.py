"""The drop-in boundary without a GPU: every function the public headers declare is exported by the shared library that
implements it (and nothing with the product prefix is exported that a header does not declare), and reference-style C++
(a layer binding the C ABI, a REGISTER_LAYER_CLASS user layer, the Net driver) compiles against the mirror headers."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = [("include/mscnn_hip.h", "mscnn_amd/libmscnn_hip.so", "MSCNN_API"),
         ("include/mscnn_net.h", "mscnn_amd/libmscnn_caffe.so", "MSCNN_NET_API"),
         ("include/mscnn_dist.h", "mscnn_amd/libmscnn_dist.so", "MSCNN_DIST_API")]


def declared(header, macro):
    text = open(os.path.join(ROOT, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(macro + r"\s+[\w\s\*]+?\b(mscnn_\w+)\s*\(", text))


@pytest.mark.parametrize("header,lib,macro", PAIRS)
def test_exports_match_header(header, lib, macro):
    want = declared(header, macro)
    assert len(want) >= 5, want
    path = os.path.join(ROOT, lib)
    assert os.path.exists(path), f"{lib} not built (python -c 'import __graft_entry__ as g; g.build()')"
    if lib.endswith("libmscnn_caffe.so"):
        ctypes.CDLL(os.path.join(ROOT, "mscnn_amd/libmscnn_hip.so"), mode=ctypes.RTLD_GLOBAL)
    L = ctypes.CDLL(path)                                   # loads without a GPU
    for name in sorted(want):
        assert hasattr(L, name), f"{lib} does not export {name} declared in {header}"
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-1].startswith("mscnn_")}
    assert exported == want, (sorted(exported - want), sorted(want - exported))


def test_product_libraries_do_not_link_vendor_math_or_the_oracle():
    for lib in ("mscnn_amd/libmscnn_hip.so", "mscnn_amd/libmscnn_caffe.so", "mscnn_amd/libmscnn_dist.so"):
        out = subprocess.run(["ldd", os.path.join(ROOT, lib)], capture_output=True, text=True).stdout
        for bad in ("rocblas", "hipblas", "MIOpen", "mkl", "oracle", "mscnn_ref", "torch"):
            assert bad not in out, (lib, bad)


def test_reference_style_cpp_compiles_against_the_mirror_headers():
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    cmd = [hipcc, "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-x", "c++",
           "-I" + os.path.join(ROOT, "mscnn_amd/host/include"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests/boundary/user_layers.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def build_boundary_binary(out_dir, source="run_boundary.cpp"):
    """tests/boundary/run_boundary.cpp (the user layers of user_layers.cpp in a live Net + the reference's SyncedMemory / Blob
    test cases) -- or another program of tests/boundary -- compiled as plain C++ against the mirror headers and linked with the
    product libraries."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = os.path.join(str(out_dir), os.path.splitext(source)[0])
    libdir = os.path.join(ROOT, "mscnn_amd")
    cmd = [hipcc, "-std=c++17", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-x", "c++", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "mscnn_amd/host/include"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests/boundary", source), "-L" + libdir, "-lmscnn_caffe", "-lmscnn_hip",
           "-Wl,-rpath," + libdir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_default_flow_program_compiles_against_the_mirror_headers(tmp_path):
    """tests/boundary/run_default_flow.cpp = the INTEGRATION.md section 2 flow as a program (no calibration call); it runs in the GPU
    suite (test_default_flow_is_safe_without_a_calibration_call), here it must compile and link with -Wall -Werror."""
    exe = build_boundary_binary(tmp_path, "run_default_flow.cpp")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr


def test_user_registered_layers_build_a_net_without_a_device(tmp_path):
    """REGISTER_LAYER_CLASS from user code reaches the product's registry, the prototxt naming the user types parses, the
    automatic Split is inserted, LayerSetUp / Reshape of the user layers run -- everything short of Forward (GPU test)."""
    exe = build_boundary_binary(tmp_path)
    r = subprocess.run([exe, str(tmp_path / "boundary.prototxt"), "construct-only"], capture_output=True, text=True)
    assert r.returncode == 0 and "CONSTRUCTION OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])


def test_wgemm_is_refused_above_2gib_planes():
    """wgemm.hip's out-of-range lane-offset sentinel is 0x80000000, so V / M / Up must stay below 2^31 bytes;
    the F(4x4,3x3) form of a conv2_2-shaped layer crosses that at batch 7 (M = 36 x 128 x 120,960 x 4 B = 2.23 GB) and must plan its
    plane GEMMs on the per-plane igemm kernel there (weight-layout id < 200), on wgemm (200 + variant) below."""
    import ctypes as C
    from mscnn_amd import hipapi as h
    L = h.lib()

    def gemm_id(N):
        d = h.ConvDesc(N, 128, 288, 960, 128, 3, 3, 1, 1, 1, 1, 1, 1, 6, 0, 0, 0)
        p = C.c_void_p()
        assert L.mscnn_conv2d_plan_create(C.byref(d), C.byref(p)) == 0
        try:
            assert L.mscnn_conv2d_plan_kernel(p).decode() == "winograd_f4x4_3x3"
            return (L.mscnn_conv2d_plan_weight_layout(p) >> 8) & 0xFFFF
        finally:
            L.mscnn_conv2d_plan_destroy(p)
    assert gemm_id(1) >= 200 and gemm_id(6) >= 200          # 36 x 128 x 103,680 x 4 B = 1.91 GB: still below the sentinel
    assert gemm_id(7) < 200 and gemm_id(8) < 200
