"""The C ABI driven by a caller that is not Python: tests/abi/abi_caller.cpp is compiled against include/pcc_hip.h,
linked with libpcc_hip.so and run -- coordinate canonicalisation (pack, radix sort, unique) and an MFMA-path convolution,
each checked against host loops inside the program; the capacity guard of the pack entry point is exercised too."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_caller_builds_and_runs(tmp_path):
    pkg = os.path.join(ROOT, "unified_point_cloud_compression_amd")
    exe = str(tmp_path / "abi_caller")
    cmd = ["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "abi", "abi_caller.cpp"),
           "-I", os.path.join(ROOT, "include"), "-L", pkg, "-lpcc_hip", f"-Wl,-rpath,{pkg}", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ABI OK" in out.stdout, (out.returncode, out.stdout, out.stderr)
