"""The header-only C++ facade (include/channelcoding_amd/cyclic.hpp): compiles as plain C++14 with g++
against the C ABI (CPU check), refuses to run without a GPU, and passes the exercises.c++-style program
on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "facade_exercises")


def build():
    lib = os.path.join(ROOT, "channelcoding_amd")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "facade_exercises.cpp"), "-o", BIN, "-L" + lib, "-lchannelcoding_amd",
           "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True, capture_output=True, text=True)


def test_facade_compiles_and_fails_loudly_without_gpu():
    build()
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([BIN], capture_output=True, text=True)
        assert out.returncode == 1 and "no usable HIP device" in out.stderr


@pytest.mark.gpu
def test_facade_exercises_on_gpu():
    build()
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout and out.stdout.count("ok ") >= 15
