"""Unpacks tests/golden/ref_index.npz -- index directories written by real Lance releases, archived from the reference's
backward-compatibility test data by tests/golden/make_ref_index_fixtures.py -- into a temporary directory, once per
process.  Layout under the returned directory: see the docstring of the make script."""
import atexit
import os
import shutil
import tempfile

import numpy as np

_ARCHIVE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_index.npz")
_dir = None


def ref_index_dir():
    global _dir
    if _dir is None:
        d = tempfile.mkdtemp(prefix="lance_ref_index_")
        with np.load(_ARCHIVE) as z:
            for key in z.files:
                path = os.path.join(d, key)
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "wb") as f:
                    f.write(z[key].tobytes())
        atexit.register(shutil.rmtree, d, ignore_errors=True)
        _dir = d
    return _dir
