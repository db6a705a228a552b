"""pytest wiring: the `gpu` marker, repo root on sys.path, golden-file loader."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests must fail loudly (not skip) when there is no GPU; plain runs skip them."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        have = False
    if have:
        return
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return  # explicitly requested: let them run and fail
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


# ---- parity margin record (profiles/parity_r6.json): the GPU parity tests report how far inside the bar they are -------------
_PARITY = {}


@pytest.fixture(scope="session")
def record_parity():
    """record_parity(key, dict): kept for the session and written to gpurun_out/parity_r6.json (merged into an existing
    file) when the session ends -- max scaled errors per config, worlds masked by the guard band, drift percentiles."""
    def rec(key, value):
        _PARITY[key] = value
    return rec


def _jsonable(x):
    if isinstance(x, dict):
        return {str(k): _jsonable(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_jsonable(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    if isinstance(x, np.ndarray):
        return x.tolist()
    return x


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "parity_r6.json")
    data = {}
    if os.path.exists(path):
        try:
            data = json.load(open(path))
        except Exception:
            data = {}
    data.update(_jsonable(_PARITY))
    data["_bar"] = "|gpu - fp64 oracle| <= 1e-5 * max(1, |oracle|) per element, teacher-forced; integer outputs exact"
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)
