"""Short, seeded runs of the two soaks (-m gpu): tools/soak_all.py (every kernel family and alternative path against the oracle,
single blocks and groups) and tools/soak_cli.py (random CLI sessions against the reference binary built from its own sources).
The long runs are recorded under profiles/ (r6_soak_*.txt); these keep the tools themselves and a slice of their input space
inside the test suite."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, seconds, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", script), str(seconds), str(seed)], capture_output=True,
                       timeout=900, cwd=ROOT)
    out = r.stdout.decode("utf-8", "replace") + r.stderr.decode("utf-8", "replace")
    assert r.returncode == 0, out[-3000:]
    return out


def test_soak_of_every_kernel_family_against_the_oracle():
    out = _run("soak_all.py", 20, 20261004)
    assert "every stage equal to the oracle's" in out, out[-2000:]


def test_soak_of_cli_sessions_against_the_reference_binary():
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "shafa")):
        pytest.skip("oracle/_ref/shafa is not built (it is made where /root/reference exists and travels with the tree)")
    out = _run("soak_cli.py", 20, 20261004)
    assert "equal to the reference binary's" in out, out[-2000:]
