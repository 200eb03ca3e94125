import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "survey_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def gpu_ops():
    """The HIP path.  Fails (never skips, never falls back) when run without the extension."""
    import torch
    from mlmcpathintegral_amd import abi, ops
    abi.load()
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    torch.cuda.set_device(0)
    return ops


# ---- statistical comparisons: every one is recorded with its z-score --------------------------------------------------------
ZSCORES = []
GATE_SIGMA = 3.0      # |z| gate of every recorded comparison (about 40 per GPU session: false alarm probability ~ 10 % for
                      # fresh random numbers; the streams are counter based and seeded, so a session's z-scores are
                      # reproducible -- what is committed under profiles/ is what the gate saw)
HEADLINE_SIGMA = 2.0  # the north-star pair: GPU chain vs CPU chain and vs the closed form at the headline shape


def zcheck(name, value, error, reference, reference_error=0.0, gate=None):
    """Assert |value - reference| < gate * combined error; record (name, z) for the session summary."""
    import math
    combined = math.hypot(error, reference_error)
    z = (value - reference) / combined if combined > 0 else float("inf")
    g = GATE_SIGMA if gate is None else gate
    ZSCORES.append({"name": name, "value": value, "error": error, "reference": reference,
                    "reference_error": reference_error, "z": z, "gate": g})
    print(f"[z] {name}: {value:.7g} +- {error:.2g} vs {reference:.7g} +- {reference_error:.2g}  z = {z:+.2f} (gate {g:g})")
    assert abs(z) < g, f"{name}: z = {z:+.2f} exceeds {g:g} sigma"
    return z


def pytest_sessionfinish(session, exitstatus):
    if not ZSCORES:
        return
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "zscores.json"), "w") as f:
            json.dump({"gate_sigma": GATE_SIGMA, "headline_sigma": HEADLINE_SIGMA, "count": len(ZSCORES),
                       "max_abs_z": max(abs(z["z"]) for z in ZSCORES), "comparisons": ZSCORES}, f, indent=1)
    except OSError:
        pass
    worst = max(ZSCORES, key=lambda z: abs(z["z"]))
    print(f"\n[z] {len(ZSCORES)} statistical comparisons, max |z| = {abs(worst['z']):.2f} ({worst['name']})")
