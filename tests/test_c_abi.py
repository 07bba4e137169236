"""The C ABI used from C: tests/c/abi_smoke.c includes include/l2o_abi.h, links libl2o_hip.so and runs without
Python or torch in the process -- what the binding of INTEGRATION.md relies on.  CPU: builds and runs the
host-only half; GPU: the whole program (one l2o_unroll checked against the definition and the step path)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
EXE = os.path.join(ROOT, "build", "abi_smoke")
LIBDIR = os.path.join(ROOT, "open_l2o_amd")


def _build():
    os.makedirs(os.path.dirname(EXE), exist_ok=True)
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), SRC,
           "-o", EXE, "-L" + LIBDIR, "-l:libl2o_hip.so", "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_c_client_builds_and_host_entry_points_work():
    _build()
    out = subprocess.run([EXE, "--host"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert "host-only checks passed" in out.stdout


@pytest.mark.gpu
def test_c_client_runs_one_unroll_on_the_gpu():
    _build()
    out = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    print(out.stdout.strip())
    assert out.returncode == 0, out.stderr
    assert "f(x_0) matches" in out.stdout
