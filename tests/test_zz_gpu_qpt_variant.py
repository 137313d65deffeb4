"""The per-query-table filter (lance_amd/csrc/search_qt.hip, LANCE_HIP_QPT=1) was written at the end of round 3
against its CPU specification (scripts/sim/pqt_filter_spec.py) after the round's GPU budget was spent: it compiles, it has
never run.  This test is the first thing to run on hardware next round; until then it only runs when asked to
(LANCE_TEST_UNVALIDATED=1), so that an unvalidated experimental path cannot turn the suite red.

What it does: the tiled-table parity cases (M = 48 / 64 / 96) again in a child process with the switch on -- bit-equal to the
oracle like the default path, since the variant only changes the FILTER."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.skipif(os.environ.get("LANCE_TEST_UNVALIDATED") != "1", reason="experimental path, not yet run on hardware (set LANCE_TEST_UNVALIDATED=1)")
def test_tiled_cases_with_per_query_tables():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LANCE_HIP_QPT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_pm_scan.py"), os.path.join(root, "tests", "test_zz_gpu_fullconfig.py"),
                        "-m", "gpu", "-q", "-x", "-k", "tiled or loose_bounds or c3", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
