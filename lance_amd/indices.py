"""Staged index building on the MI355X engine, mirroring `lance.indices.IndicesBuilder`
(python/python/lance/indices/builder.py:26-520; models: indices/ivf.py, indices/pq.py).

The reference class works on a Lance dataset + column name; this one works on the vector matrix itself (numpy or a
torch tensor already in HBM) -- the part of the reference class that is on the hot path.  Stage names, arguments,
defaults and validation errors follow the reference so that a pipeline written against `IndicesBuilder` reads the same:

    builder = IndicesBuilder(vectors)
    ivf = builder.train_ivf(num_partitions=256, distance_type="l2")       # builder.py:60-147
    pq = builder.train_pq(ivf, num_subvectors=16)                         # builder.py:149-204
    parts = builder.assign_ivf_partitions(ivf)                            # builder.py:206-259 -> (row_id, partition)
    for batch in builder.transform_vectors(ivf, pq): ...                  # builder.py:261-315 -> shuffle-buffer batches

Models are Arrow arrays exactly as the reference's IvfModel / PqModel hold them (centroids: FixedSizeList[d] with
nlist rows; codebook: FixedSizeList[d] with 256 rows = [M][256][d/M] flattened); save/load use Arrow IPC files here
(the reference writes Lance files, which need pylance).
"""
import math
import warnings

import numpy as np
import pyarrow as pa

PARTITION_COLUMN = "__ivf_part_id"
PQ_COLUMN = "__pq_code"


class IvfModel:
    def __init__(self, centroids: pa.FixedSizeListArray, distance_type: str):
        self.centroids = centroids
        self.distance_type = distance_type

    @property
    def num_partitions(self) -> int:
        return len(self.centroids)

    def to_numpy(self):
        d = self.centroids.type.list_size
        return np.asarray(self.centroids.values.to_numpy(zero_copy_only=False)).reshape(-1, d)

    def save(self, uri: str):
        schema = pa.schema([pa.field("centroids", self.centroids.type)], metadata={b"distance_type": self.distance_type.encode()})
        with pa.OSFile(uri, "wb") as sink, pa.ipc.new_file(sink, schema) as w:
            w.write_batch(pa.record_batch([self.centroids], schema=schema))

    @classmethod
    def load(cls, uri: str):
        t = pa.ipc.open_file(uri).read_all()
        return cls(t["centroids"].combine_chunks(), t.schema.metadata[b"distance_type"].decode())


class PqModel:
    def __init__(self, num_subvectors: int, codebook: pa.FixedSizeListArray):
        self.num_subvectors = num_subvectors
        self.codebook = codebook

    @property
    def dimension(self):
        return self.codebook.type.list_size

    def to_numpy(self):
        """(M, 2^nbits, d/M) float array"""
        flat = np.asarray(self.codebook.values.to_numpy(zero_copy_only=False))
        return flat.reshape(self.num_subvectors, -1, self.dimension // self.num_subvectors)

    def save(self, uri: str):
        schema = pa.schema([pa.field("codebook", self.codebook.type)], metadata={b"num_subvectors": str(self.num_subvectors).encode()})
        with pa.OSFile(uri, "wb") as sink, pa.ipc.new_file(sink, schema) as w:
            w.write_batch(pa.record_batch([self.codebook], schema=schema))

    @classmethod
    def load(cls, uri: str):
        t = pa.ipc.open_file(uri).read_all()
        return cls(int(t.schema.metadata[b"num_subvectors"].decode()), t["codebook"].combine_chunks())


class IndicesBuilder:
    def __init__(self, vectors, engine=None):
        shape = tuple(vectors.shape)
        if len(shape) != 2:
            raise TypeError(f"Vector column must be a 2-D (rows, dimension) array of floats or int8, got shape {shape}")
        self.vectors = vectors
        self.num_rows, self.dimension = int(shape[0]), int(shape[1])
        self._engine = engine

    # ---- stages (device work) -------------------------------------------------------------------------------
    def train_ivf(self, num_partitions=None, *, distance_type="l2", accelerator=None, sample_rate: int = 256, max_iters: int = 50,
                  seed: int = 42) -> IvfModel:
        num_partitions = self._determine_num_partitions(num_partitions, self.num_rows)
        self._verify_ivf_sample_rate(sample_rate, num_partitions, self.num_rows)
        distance_type = self._normalize_distance_type(distance_type)
        self._verify_ivf_params(num_partitions)
        from . import vector as lv
        params = lv.IvfPqParams(int(num_partitions), 1, 8, lv._normalize_metric_type(distance_type), max_iters, sample_rate, seed)
        cent, _, _ = lv.train_ivf_centroids(self.vectors, params, self._engine)
        c = cent.cpu().numpy()
        arr = pa.FixedSizeListArray.from_arrays(pa.array(c.reshape(-1)), c.shape[1])
        return IvfModel(arr, distance_type)

    def train_pq(self, ivf_model: IvfModel, num_subvectors=None, *, sample_rate: int = 256, max_iters: int = 50, seed: int = 42) -> PqModel:
        num_subvectors = self._normalize_pq_params(num_subvectors, self.dimension)
        self._verify_pq_sample_rate(self.num_rows, sample_rate)
        from . import vector as lv
        params = lv.IvfPqParams(ivf_model.num_partitions, num_subvectors, 8, lv._normalize_metric_type(ivf_model.distance_type),
                                max_iters, sample_rate, seed)
        cb, _ = lv.train_pq_codebook(self.vectors, ivf_model.to_numpy(), params, self._engine)
        flat = cb.cpu().numpy().reshape(-1)
        return PqModel(num_subvectors, pa.FixedSizeListArray.from_arrays(pa.array(flat), self.dimension))

    def assign_ivf_partitions(self, ivf_model: IvfModel, accelerator=None) -> pa.Table:
        """-> table (row_id: uint64, partition: uint32), rows without a partition dropped (builder.py:206-259)"""
        from . import vector as lv
        eng = self._engine or lv.default_engine()
        metric = lv._normalize_metric_type(ivf_model.distance_type)
        x = eng.normalize(self.vectors) if metric == "cosine" else self.vectors
        ids, _ = eng.assign(x, ivf_model.to_numpy(), "l2" if metric == "cosine" else metric)
        part = ids.cpu().numpy().view(np.uint32)
        keep = part != 0xFFFFFFFF
        return pa.table({"row_id": pa.array(np.arange(self.num_rows, dtype=np.uint64)[keep]), "partition": pa.array(part[keep])})

    def transform_vectors(self, ivf: IvfModel, pq: PqModel, batch_size: int = 10240):
        """-> iterator of RecordBatches (row_id, __ivf_part_id, __pq_code): the unsorted storage of builder.py:261-315"""
        from . import arrow_io, vector as lv
        eng = self._engine or lv.default_engine()
        part, codes, _ = eng.ivfpq_encode(self.vectors, ivf.to_numpy(), pq.to_numpy(), lv._normalize_metric_type(ivf.distance_type))
        return arrow_io.shuffle_buffer_batches(np.arange(self.num_rows, dtype=np.uint64), part.cpu().numpy(), codes.cpu().numpy(), batch_size)

    # ---- argument checks: same rules and messages as the reference (builder.py:409-487) -------------------
    def _determine_num_partitions(self, num_partitions, num_rows):
        return round(math.sqrt(num_rows)) if num_partitions is None else num_partitions

    def _normalize_pq_params(self, num_subvectors, dimension):
        if num_subvectors is None:
            if dimension % 16 == 0:
                return dimension // 16
            if dimension % 8 == 0:
                return dimension // 8
            raise ValueError(f"vector dimension {dimension} is not divisible by 16 or 8. PQ performance will be poor."
                             "  Cowardly refusing to create PQ model.  Please specify num_subvectors manually.")
        if not isinstance(num_subvectors, int):
            raise ValueError("num_subvectors must be an int")
        if num_subvectors < 1:
            raise ValueError("num_subvectors must be greater than 0")
        if num_subvectors > dimension:
            raise ValueError("num_subvectors must be less than or equal to the dimension of the vectors")
        if dimension % num_subvectors != 0:
            raise ValueError(f"dimension ({dimension}) must be divisible by num_subvectors ({num_subvectors}) without remainder")
        return num_subvectors

    def _verify_base_sample_rate(self, sample_rate):
        if not isinstance(sample_rate, int) or sample_rate < 2:
            raise ValueError(f"The sample_rate must be an int greater than 1, got {sample_rate}")

    def _verify_pq_sample_rate(self, num_rows, sample_rate):
        self._verify_base_sample_rate(sample_rate)
        if 256 * sample_rate > num_rows:
            raise ValueError("There are not enough rows in the dataset to create PQ codebook with a sample rate of "
                             f"{sample_rate}.  {sample_rate * 256} rows needed and there are {num_rows}")

    def _verify_ivf_sample_rate(self, sample_rate, num_partitions, num_rows):
        self._verify_base_sample_rate(sample_rate)
        if num_partitions * sample_rate > num_rows:
            raise ValueError(f"There are not enough rows in the dataset to create IVF centroids with {num_partitions} partitions and "
                             f"a sample rate of {sample_rate}. {sample_rate * num_partitions} rows needed and there are {num_rows}")

    def _verify_ivf_params(self, num_partitions):
        if num_partitions is None:
            raise ValueError("num_partitions and num_sub_vectors are required for IVF_PQ")
        if isinstance(num_partitions, float):
            warnings.warn("num_partitions is float, converting to int")
        elif not isinstance(num_partitions, int):
            raise TypeError(f"num_partitions must be int, got {type(num_partitions)}")

    def _normalize_distance_type(self, distance_type):
        if not isinstance(distance_type, str) or distance_type.lower() not in ("l2", "cosine", "euclidean", "dot"):
            raise ValueError(f"Distance type {distance_type} not supported.")
        return distance_type.lower()
