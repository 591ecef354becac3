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
def test_reference_container_and_batcher_run_our_layers_on_cpu_and_match_the_reference_layers(tmp_path):
    """Two fresh interpreters: (1) the reference imports the oracle's torch_scatter restatement; the reference's own
    container + batcher + pooling module run around our layers on CPU tensors and match its own layers to 1e-6
    (eval, training with dropout, every parameter gradient, save / restore-on-cpu / predict); (2) the same with
    `ptgnn_amd.scatter.install()` AS `torch_scatter` -- no oracle scatter in the process -- against run (1)'s numbers."""
    env = dict(os.environ, PYTHONHASHSEED="0")
    saved = str(tmp_path / "oracle_scatter_run.npz")
    for extra, tag in (["--scatter", "oracle", "--save", saved], "scatter=oracle"), \
                      (["--scatter", "facade", "--expect", saved], "scatter=facade"):
        proc = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_check.py")] + extra, env=env,
                              capture_output=True, text=True, timeout=600)
        assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
        assert "DROPIN_OK" in proc.stdout and tag in proc.stdout
