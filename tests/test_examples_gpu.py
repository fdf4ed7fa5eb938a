"""The example scripts (the build's own counterparts of the reference's code/test_clip.py and
code/search_image.py main(), SURVEY.md section 8b "who calls it") run end to end on the GPU."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "examples", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_zero_shot_example():
    probs = _load("zero_shot_synthetic").main(["--model", "tiny-test"])
    assert probs.shape == (1, 3) and np.isfinite(probs).all()
    assert abs(float(probs.sum()) - 1.0) < 1e-5


def test_gallery_search_example():
    report = _load("gallery_search_synthetic").main(["--model", "tiny-test", "--classes", "4", "--per-class", "40",
                                                     "--image-size", "96"])
    assert len(report) == 8
    for c, name, pos, neg, hits in report:
        assert np.isfinite([pos, neg]).all()
        if name == "image mean":
            # images of a class share a prototype: their scores against the class query separate from the rest
            assert pos > neg, (c, pos, neg)
            assert hits >= 8, (c, hits)


def test_c_program_uses_the_abi_without_python(tmp_path):
    """examples/c_abi_search.c: a plain C host (HIP runtime + include/mmr.h, no torch) gets the same top-k as a
    double-precision host loop."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs gcc and the ROCm headers")
    from mmr_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    exe = tmp_path / "c_abi_search"
    subprocess.check_call([gcc, "-std=c99", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                           "-I", "/opt/rocm/include", os.path.join(ROOT, "examples", "c_abi_search.c"), "-o", str(exe),
                           "-L", libdir, "-l:libmmr_hip.so", "-L", "/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "matches the host loop" in out.stdout
