"""lance_amd -- MI355X-native engine for Lance's IVF-PQ hot path.

k-means training, PQ codebook learning/encoding and the flat + IVF-PQ distance scans run
as hand-written gfx950 HIP kernels in liblance_hip.so (C ABI: include/lance_hip.h); this
package is the Python host side mirroring the reference's interface for that path.
There is no CPU fallback: importing works anywhere, running needs the built library and
a GPU.
"""
from . import _lib
from ._lib import LanceHipError

__all__ = ["_lib", "LanceHipError", "Engine", "DeviceIndex", "KMeans", "IvfPqParams", "IvfPqIndex", "create_index",
           "flat_knn", "train_ivf_centroids", "train_pq_codebook", "default_engine", "load_index", "validate_vector_index", "IndicesBuilder", "IvfModel", "PqModel"]


def __getattr__(name):
    # torch-dependent modules are imported lazily so that `import lance_amd` stays cheap
    if name in ("Engine", "DeviceIndex"):
        from . import engine
        return getattr(engine, name)
    if name in ("KMeans", "IvfPqParams", "IvfPqIndex", "create_index", "flat_knn", "train_ivf_centroids",
                "train_pq_codebook", "default_engine", "load_index", "validate_vector_index"):
        from . import vector
        return getattr(vector, name)
    if name in ("IndicesBuilder", "IvfModel", "PqModel"):
        from . import indices
        return getattr(indices, name)
    raise AttributeError(name)
