"""The C header is valid C99, the C++ adapter compiles and links against the library,
and its host-only getters work (no compute without a GPU)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_is_plain_c(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include "orbslamm_hip.h"\nint main(void){OrbxKeyPoint k; return sizeof(k)==28?0:1;}\n')
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-fsyntax-only", str(src)])


def test_cpp_adapter_builds_and_runs(tmp_path):
    from orbslamm_amd import _lib
    _lib.lib()
    exe = str(tmp_path / "adapter")
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "adapter_compile.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "orbslamm_amd"), "-lorbslamm_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "orbslamm_amd"), "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "adapter ok" in out.stdout and "no CPU fallback" in out.stderr
