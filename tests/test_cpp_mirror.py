"""The C++ mirror of the reference surface (include/microflow.hpp): it must compile against
the C ABI with plain g++ (CPU check), and on a GPU the reference's unit tests written against
it must pass (tests/cpp/reference_kats.cpp)."""
import os
import subprocess

import pytest

from tests.conftest import MODELS, ROOT

SRC = os.path.join(ROOT, "tests", "cpp", "reference_kats.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "reference_kats")


SHIM_SRC = os.path.join(ROOT, "tests", "cpp", "shim_flow.cpp")
SHIM_EXE = os.path.join(ROOT, "tests", "cpp", "shim_flow")


def build(src=SRC, exe=EXE):
    from microflow_rs_amd import _lib
    _lib.lib()  # make sure libmicroflow_amd.so exists
    libdir = os.path.dirname(_lib.lib_path())
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
           "-L", libdir, "-lmicroflow_amd", "-L", "/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return exe


def test_cpp_mirror_compiles_and_links():
    assert os.path.exists(build())


@pytest.mark.gpu
def test_reference_kats_through_cpp_mirror():
    exe = build()
    out = subprocess.run([exe, MODELS], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ok" in out.stdout


def test_shim_flow_compiles_and_links():
    """tests/cpp/shim_flow.cpp: the Rust shim's call sequence against the plain C header"""
    assert os.path.exists(build(SHIM_SRC, SHIM_EXE))


@pytest.mark.gpu
def test_shim_call_sequence_reproduces_the_reference_vectors():
    """rust/ cannot be compiled here; its exact ABI call sequence (ModelSet::new, predict,
    predict_quantized, predict_batch incl. the column-major Buffer flatten/unflatten) in C++ gives the
    reference's whole-model outputs (tests/{sine,speech,person_detect}.rs)."""
    exe = build(SHIM_SRC, SHIM_EXE)
    out = subprocess.run([exe, MODELS], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "shim flow ok" in out.stdout
