"""CPU test of the drop-in claim against the reference's own classes (skipped where /root/reference is absent,
e.g. on the GPU box).  Runs tests/dropin_check.py in a fresh interpreter: the reference has to be importable
before ptgnn_amd.layers is first imported."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import shims  # noqa: E402


@pytest.mark.skipif(not shims.reference_available(), reason="reference checkout not mounted")
def test_reference_model_accepts_our_layers_and_their_state_dicts():
    env = dict(os.environ, PYTHONHASHSEED="0")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_check.py")], env=env,
                          capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    assert "DROPIN_OK" in proc.stdout
