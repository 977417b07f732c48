"""The boundary is a real C ABI: a plain C program (no Python/torch in the process) links libmi355scf.so,
drives it through include/mi355scf.h and reproduces the Szabo-Ostlund H2/STO-3G RHF energy."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_plain_c_program_through_the_abi(tmp_path):
    csrc = os.path.join(ROOT, "computational-chemistry-ai_amd", "csrc")
    exe = str(tmp_path / "c_abi_smoke")
    cmd = ["gcc", "-O1", "-D__HIP_PLATFORM_AMD__", os.path.join(ROOT, "tests", "c_abi_smoke.c"), "-I", os.path.join(ROOT, "include"),
           "-I", "/opt/rocm/include", "-L", csrc, "-lmi355scf", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "E(RHF) = -1.11671" in out.stdout
