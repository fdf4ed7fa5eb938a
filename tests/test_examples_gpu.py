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
