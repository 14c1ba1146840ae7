"""The header-only C++ facade (include/channelcoding_amd/cyclic.hpp): compiles as plain C++14 with g++
against the C ABI (CPU check), refuses to run without a GPU, and passes the exercises.c++-style program
on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "facade_exercises")
ADAPTER = os.path.join(ROOT, "tests", "cpp", "decoder_adapter")


def build(name="facade_exercises"):
    lib = os.path.join(ROOT, "channelcoding_amd")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", os.path.join(ROOT, "tests", "cpp", name),
           "-L" + lib, "-lchannelcoding_amd", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]


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


def test_decoder_adapter_compiles_and_fails_loudly_without_gpu():
    """include/channelcoding_amd/simulation.hpp: the reference's type-erased `decoder` (simulation.h:23-69)."""
    build("decoder_adapter")
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([ADAPTER], capture_output=True, text=True)
        assert out.returncode == 1 and "no usable HIP device" in out.stderr


@pytest.mark.gpu
def test_decoder_adapter_on_gpu():
    build("decoder_adapter")
    out = subprocess.run([ADAPTER], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout and out.stdout.count("ok ") >= 4
