"""The C++ host-side mirror (include/mgard_hip.hpp) driven by a C++ program written like a
MGARD-X low-level API consumer; built with hipcc against libmgard_hip.so and run on the GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["lowlevel_roundtrip", "highlevel_roundtrip", "highlevel_api_example"])
def test_cpp_roundtrip(tmp_path, prog):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / prog)
    lib = os.path.join(ROOT, "mgard_amd", "libmgard_hip.so")
    assert os.path.exists(lib), "libmgard_hip.so is not built"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17",
                           "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", prog + ".cpp"),
                           "-L", os.path.dirname(lib), "-lmgard_hip",
                           "-Wl,-rpath," + os.path.dirname(lib), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and "OK" in out.stdout


def test_cpp_header_compiles_on_host():
    """No GPU needed: the header-only mirror must compile as plain C++17 against the C ABI."""
    src = ('#include "mgard_hip.hpp"\n#include "compress_hip.hpp"\n'
           'int main() { mgard_hip::HighLevelConfig c; return c.dev_id; }\n')
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        "-x", "c++", "-"], input=src, text=True, capture_output=True)
    assert p.returncode == 0, p.stderr


def test_mgard_x_namespace_header_compiles_on_host():
    """include/compress_x_hip.hpp: the reference's names in namespace mgard_x, plain C++17."""
    src = ('#include "compress_x_hip.hpp"\n'
           'int main() { mgard_x::Config c; c.lossless = mgard_x::lossless_type::Huffman_Zstd;\n'
           '  c.dev_type = mgard_x::device_type::HIP; c.domain_decomposition_sizes = {4, 4};\n'
           '  std::vector<mgard_x::SIZE> shape{8, 8, 8}; return (int)shape.size() + c.dev_id - 3; }\n')
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        "-x", "c++", "-"], input=src, text=True, capture_output=True)
    assert p.returncode == 0, p.stderr
