"""The C++ counterparts of the reference's example drivers (examples/train_mnist.rs,
examples/train_mnist_cnn.rs) run end to end on the host library
(SURVEY 8d's synthetic rows carry random labels, so the check is a finite loss near ln 10; learning
curves against the oracle are tests/test_gpu_step.py's job)."""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "examples" / "_build"


def _run(name, *flags):
    exe = BIN / name
    subprocess.check_call(["make", "-s", "-C", str(ROOT / "examples")])   # no-op when current; never run a stale binary
    out = subprocess.run([str(exe), "--data-dir", "/nonexistent", *flags], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "Training Complete!" in out.stdout
    return [float(x) for x in re.findall(r"Train Loss: ([0-9.]+)", out.stdout)]


def test_examples_build_without_a_gpu():
    subprocess.check_call(["make", "-C", str(ROOT / "examples")])
    assert (BIN / "train_mnist").exists() and (BIN / "train_mnist_cnn").exists() and (BIN / "cabi_step").exists()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(), ("--eager",)], ids=["graph", "eager"])
def test_train_mnist_example_runs(mode):
    losses = _run("train_mnist", "--epochs", "3", "--train-n", "4096", "--test-n", "1024", "--batch-size", "256", *mode)
    assert len(losses) == 3 and all(l == l and 1.0 < l < 3.0 for l in losses)      # finite, near ln(10): the synthetic labels are noise


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [(), ("--full-backward",)], ids=["faithful", "full_backward"])
def test_train_mnist_cnn_example_runs(mode):
    losses = _run("train_mnist_cnn", "--epochs", "2", "--train-n", "1024", "--test-n", "512", "--batch-size", "128", *mode)
    assert len(losses) == 2 and all(l == l and l < 5.0 for l in losses)


def test_boundary_headers_are_plain_c11():
    """include/*.h is the drop-in boundary: it must compile as C (extern "C", plain pointers and sizes)"""
    for h in ("taper_hip.h", "taper_host.h"):
        subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-x", "c", "-fsyntax-only", "-"],
                       input=f'#include "{ROOT}/include/{h}"\n', text=True, check=True)


@pytest.mark.gpu
def test_plain_c_client_trains_through_the_abi():
    """examples/cabi_step.c: the fused 2-launch MLP step driven from C through th_* alone"""
    subprocess.check_call(["make", "-s", "-C", str(ROOT / "examples")])
    out = subprocess.run([str(BIN / "cabi_step")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "loss decreased" in out.stdout and "adam t = 50" in out.stdout
