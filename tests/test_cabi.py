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


def build_boundary_binary(out_dir):
    """tests/boundary/run_boundary.cpp (the user layers of user_layers.cpp in a live Net + the reference's SyncedMemory / Blob
    test cases) compiled as plain C++ against the mirror headers and linked with the product libraries."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    exe = os.path.join(str(out_dir), "run_boundary")
    libdir = os.path.join(ROOT, "mscnn_amd")
    cmd = [hipcc, "-std=c++17", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-x", "c++", "-I/opt/rocm/include",
           "-I" + os.path.join(ROOT, "mscnn_amd/host/include"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests/boundary/run_boundary.cpp"), "-L" + libdir, "-lmscnn_caffe", "-lmscnn_hip",
           "-Wl,-rpath," + libdir, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_user_registered_layers_build_a_net_without_a_device(tmp_path):
    """REGISTER_LAYER_CLASS from user code reaches the product's registry, the prototxt naming the user types parses, the
    automatic Split is inserted, LayerSetUp / Reshape of the user layers run -- everything short of Forward (GPU test)."""
    exe = build_boundary_binary(tmp_path)
    r = subprocess.run([exe, str(tmp_path / "boundary.prototxt"), "construct-only"], capture_output=True, text=True)
    assert r.returncode == 0 and "CONSTRUCTION OK" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])
