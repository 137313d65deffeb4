"""The per-query-table filter for the tiled PQ shapes (lance_amd/csrc/search_qt.hip): with r = q - cen_p the table entry
splits into a per-QUERY table, a constant per stored ROW and a scalar per (query, partition) pair, so the integer table is built once
per query and a work item loads its four queries' tables instead of computing them.  LANCE_HIP_QPT (read once per process):
  1 -- tables after the bound pass, scale from the bound T (first run on hardware: round 4, gpurun r04a: parity test + 176 fuzz cases);
  2 -- tables BEFORE the bound pass, scale from the distance of an average code, shared by the bound pass and the main pass
       (the DEFAULT since gpurun r04g, so the main suite covers it);
  0 -- off: a table build per (query, partition) item as in round 3.
The variant only changes the FILTER; survivors are re-evaluated in the reference's arithmetic, so the tiled-table parity cases
(M = 48 / 64 / 96) and the C3 full-configuration case must stay bit-equal to the oracle.  They run again in a child process per non-default mode."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["0", "1"])
def test_tiled_cases_with_per_query_tables(mode):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LANCE_HIP_QPT=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_gpu_pm_scan.py"), os.path.join(root, "tests", "test_zz_gpu_fullconfig.py"),
                        "-m", "gpu", "-q", "-x", "-k", "tiled or loose_bounds or c3", "-p", "no:cacheprovider"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
