"""The 8-queries-per-gather filter scan (lance_amd/csrc/search_q8.hip, round 3) is selected by LANCE_HIP_Q8=1, read once per
process: the partition-major parity cases run again in a child process with the switch on (M = 16 shapes take the 8-bit
table, the two-phase cut of the merge kernel and the per-segment rescan; the others are unaffected).  Bit-equal to the oracle
like the default path -- the variant is a different FILTER, the survivors are re-evaluated in the reference's arithmetic."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_pm_scan_cases_with_the_8_query_filter():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LANCE_HIP_Q8="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_pm_scan.py"), "-m", "gpu", "-q", "-x",
                        "-k", "f32_every_instantiation or loose_bounds or two_class or random_shapes or overflow", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
