import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_small():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "golden_small.npz"))


@pytest.fixture(scope="session")
def golden_vitb16():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "golden_vitb16.npz"))


@pytest.fixture(scope="session")
def golden_vitb32():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "golden_vitb32.npz"))


@pytest.fixture(scope="session")
def golden_vitl14():
    import numpy as np
    return np.load(os.path.join(REPO, "tests", "golden", "golden_vitl14_336.npz"))


def oracle_clip():
    """The CPU oracle's `clip` stand-in (tests only)."""
    import importlib
    return importlib.import_module("oracle.clip")


_ORACLE_LOGITS = {}


def oracle_logits(name, images, template, classnames):
    """logits_per_image of the reference loop on the CPU oracle -- clip_model(image[None], text) image by image, as utils/clip_pseudolabels.py:31-37
    runs it -- for a pool of images.  They do not depend on k, and several GPU tests ask for the same pool: computed once per (model, prompts, pool
    content) and process (the live oracle was 200 of the GPU suite's 620 seconds in r05)."""
    import hashlib
    import torch
    from oracle import wrappers as W
    key = (name, template, tuple(classnames), tuple(images.shape), hashlib.sha256(images.contiguous().numpy().tobytes()).hexdigest())
    if key not in _ORACLE_LOGITS:
        oc = oracle_clip()
        om, _ = oc.load(name)
        text = oc.tokenize(W.zero_shot_prompt_strings(template, classnames))
        with torch.no_grad():
            _ORACLE_LOGITS[key] = torch.cat([om(images[i:i + 1], text)[0] for i in range(images.shape[0])])
    return _ORACLE_LOGITS[key]


def write_report(name, payload):
    """Per-case evidence of a GPU test as a JSON file: tests/_out/<name> (what the round-end driver pulls from the GPU box) and, where the box has a
    gpurun_out/ to merge back, gpurun_out/tests_out/<name>.  Assertions stay in the tests; this keeps the numbers behind them."""
    import json
    for d in (os.path.join(REPO, "tests", "_out"), os.path.join(REPO, "gpurun_out", "tests_out")):
        try:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, name), "w") as f:
                json.dump(payload, f, indent=1, default=str)
        except OSError:
            pass


def structured_pool(seed, n, res, device="cuda", block=64):
    """[n, 3, res, res] images of grip_amd.data.synthetic.structured_images on `device`, generated block by block on a thread pool (the counter RNG is
    numpy on the host: 6 000 ViT-B/16 images take 20 s on one thread -- the GPU suite spent more time drawing images than encoding them)."""
    from concurrent.futures import ThreadPoolExecutor

    import torch
    from grip_amd.data.synthetic import structured_images
    pool = torch.empty(n, 3, res, res, device=device)
    with ThreadPoolExecutor(max_workers=8) as ex:
        for lo, x in ex.map(lambda lo: (lo, structured_images(seed, lo, min(lo + block, n), res)), range(0, n, block)):
            pool[lo:lo + x.shape[0]] = x.to(device)
    return pool
