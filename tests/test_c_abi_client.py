"""include/mcd.h from plain C: tests/c_abi/abi_client.c is compiled as strict C99 (-Wall -Wextra -Werror -pedantic)
against the header and linked with libmcd_hip.so -- the boundary a cgo / JNI / ctypes binding of the reference would sit
on.  Without a GPU it checks linkage and the error behaviour; with one, a closed-form evaluation."""
import os
import subprocess

import pytest

from conftest import ensure_library_built

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    lib_dir = os.path.dirname(ensure_library_built())
    exe = str(tmp_path / "abi_client")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c_abi", "abi_client.c"), "-o", exe, "-L", lib_dir, "-lmcd_hip", "-lm",
                    "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_header_is_c99_and_every_entry_point_links(tmp_path):
    res = subprocess.run([_build(tmp_path)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "entry points link from C" in res.stdout


@pytest.mark.gpu
def test_c_client_evaluates_closed_form_on_the_device(tmp_path):
    res = subprocess.run([_build(tmp_path), "--gpu"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "gpu closed form ok" in res.stdout
