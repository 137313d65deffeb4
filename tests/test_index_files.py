"""Index files: the native reader/writer of `index.idx` + `auxiliary.idx` against index
files written by real Lance releases (tests/golden/ref_index.npz, archived from the reference's own backward-compatibility
fixtures by tests/golden/make_ref_index_fixtures.py), and -- the strongest parity pin in this repo -- the oracle's
assign / residual / PQ-encode / loss arithmetic against what the reference itself stored in those files.

No GPU: parsing, writing and the oracle run on the host.  The files -> HBM half is in test_gpu_parity.py.
"""
import ctypes as C
import json
import os
import shutil
import struct

import numpy as np
import pytest

import oracle
from lance_amd import _lib
from lance_amd import index_file as IF
from lance_file_probe import Probe, fields
from ref_fixtures import ref_index_dir

HERE = os.path.dirname(os.path.abspath(__file__))
REF27 = os.path.join(ref_index_dir(), "v0.27.1_pq_in_schema")
REF29 = os.path.join(ref_index_dir(), "v0.29.0_index")


# ---- reading what the reference wrote --------------------------------------------------------------------------------
def test_read_reference_index_v0_27():
    """datagen.py of the fixture: 512 x 32 random f32, IVF_PQ, num_partitions=1, num_sub_vectors=4 (Lance 0.27.1,
    PQ codebook still inline in the schema metadata)."""
    c = IF.read_index_files(REF27)
    assert (c.index_type, c.metric, c.dtype) == ("IVF_PQ", "l2", "float32")
    assert c.centroids.shape == (1, 32) and c.codebook.shape == (4, 256, 8)
    assert (c.num_sub_vectors, c.nbits, c.transposed) == (4, 8, True)
    assert c.part_offsets.tolist() == [0, 512]
    assert np.array_equal(c.row_ids, np.arange(512, dtype=np.uint64))
    assert c.codes.shape == (2048,) and c.loss is not None and 1000 < c.loss < 2000
    # independent walk of the same bytes
    p = Probe(os.path.join(REF27, "auxiliary.idx"))
    assert (p.major, p.minor, p.magic) == (0, 3, b"LANC")
    assert p.page_bytes(1) == c.codes.tobytes() and p.page_bytes(0) == c.row_ids.tobytes()
    meta = json.loads(json.loads(p.metadata["storage_metadata"])[0])
    tensor = {a: v for a, _, v in fields(bytes(meta["codebook_tensor"]))}
    assert tensor[3] == c.codebook.tobytes()          # [m][256][d/m] is the flattened FSL(d) x 256 tensor as stored


def test_read_reference_index_v0_29():
    c = IF.read_index_files(REF29)     # 256 x 16, one partition, 4 sub-vectors
    assert (c.index_type, c.metric) == ("IVF_PQ", "l2")
    assert c.centroids.shape == (1, 16) and c.codebook.shape == (4, 256, 4) and c.part_offsets.tolist() == [0, 256]
    assert sorted(c.row_ids.tolist()) == list(range(256))
    # every stored code must be its own nearest codeword's index for the decoded... no raw data in the tree for this
    # fixture; structural checks only
    assert c.codes_row_major().shape == (256, 4)


def test_read_data_file_column():
    vec = IF.read_column(os.path.join(REF27, "data.lance"), "vec", np.float32, 32)
    ids = IF.read_column(os.path.join(REF27, "data.lance"), "id", np.int64)
    assert vec.shape == (512, 32) and np.array_equal(ids, np.arange(512))
    assert 0.0 <= vec.min() and vec.max() < 1.0     # pc.random
    with pytest.raises(_lib.LanceHipError):
        IF.read_column(os.path.join(REF27, "data.lance"), "nope", np.float32)
    with pytest.raises(ValueError):
        IF.read_column(os.path.join(REF27, "data.lance"), "vec", np.float32, 16)


# ---- the pin: oracle arithmetic == what the reference stored -----------------------------------------------------------
def test_oracle_reproduces_reference_pq_codes_and_loss():
    """Inputs: the fixture's raw vectors + the centroids and codebook the reference trained.  The oracle's
    residual (residual.rs:58-102) + PQ encode (pq.rs:116-191, L2 argmin over 256 codewords per sub-vector) must give
    the reference's stored `__pq_code` bytes exactly, and the f64 sum of the oracle's f32 assignment distances
    (l2.rs:57-91 lane order; ivf/transform.rs loss) must equal the k-means loss the reference recorded, to the bit."""
    c = IF.read_index_files(REF27)
    vec = IF.read_column(os.path.join(REF27, "data.lance"), "vec", np.float32, 32)
    x = vec[c.row_ids.astype(np.int64)]
    part, dist = oracle.assign(x, c.centroids, "l2")
    assert np.array_equal(part, c.part_ids())
    res = oracle.residual(x, c.centroids, part)
    codes = oracle.pq_encode(res, c.codebook, "l2")
    assert np.array_equal(codes, c.codes_row_major())
    assert float(dist.astype(np.float64).sum()) == c.loss
    # and the stored layout is the per-partition transpose (pq/storage.rs:430-449)
    assert np.array_equal(oracle.transpose(codes).reshape(-1), c.codes)


def test_oracle_search_on_reference_index_is_self_consistent():
    """ADC distances from the oracle on the reference's own index: a stored vector queried against it finds itself
    among the nearest (the codes are its own quantisation) -- mirrors python/lance/util.py:171-220."""
    c = IF.read_index_files(REF27)
    vec = IF.read_column(os.path.join(REF27, "data.lance"), "vec", np.float32, 32)
    q = vec[:32]
    lut_hits = 0
    codes_t = c.codes.reshape(4, 512)
    for i in range(32):
        lut = oracle.build_lut(q[i] - c.centroids[0], c.codebook, "l2")
        d = oracle.pq_scan(lut, codes_t, "l2")
        ids, _ = oracle.heap_topk(d, c.row_ids, 10)
        lut_hits += int(i in ids.tolist())
    assert lut_hits >= 30


def _legacy_index(path):
    """Legacy (v1, Lance <= 0.21) index.idx: [per partition: PQ codes n*m u8, row ids n*8] ... pb Index (length-prefixed)
    at the position held in the 16-byte footer.  Test-side parse only -- the engine does not read this format.
    -> (centroids [nlist,d], codebook [m,256,d/m], lengths, raw bytes)"""
    import struct
    b = open(path, "rb").read()
    assert b[-4:] == b"LANC"
    pos = struct.unpack("<Q", b[-16:-8])[0]
    ln = struct.unpack_from("<I", b, pos)[0]
    vi = [v for fn, _, v in fields(b[pos + 4:pos + 4 + ln]) if fn == 5][0]          # Index.vector_index
    stages = [v for fn, _, v in fields(vi) if fn == 3]
    ivf = [v for fn, _, v in fields(stages[0]) if fn == 2][0]
    pq = [v for fn, _, v in fields(stages[1]) if fn == 3][0]
    ivf_f = {fn: v for fn, _, v in fields(ivf)}
    pq_f = {fn: v for fn, _, v in fields(pq)}
    from lance_file_probe import packed
    lengths = packed(ivf_f[3])
    d, m = pq_f[3], pq_f[2]
    cent = np.frombuffer({fn: v for fn, _, v in fields(ivf_f[4])}[3], np.float32).reshape(len(lengths), d)
    cb = np.frombuffer({fn: v for fn, _, v in fields(pq_f[5])}[3], np.float32).reshape(m, 256, d // m)
    return cent, cb, lengths, b


def test_oracle_kmeans_reproduces_reference_trained_centroid():
    """v0.21.0_legacy (in tests/golden/ref_index.npz): Lance 0.21.0 built IVF1/PQ1 over 256 x 16 random vectors, so the IVF
    k-means trained on ALL rows (256 <= sample_rate * 1) and its one centroid is the M-step mean of every row -- free of
    the unseeded initialisation.  The oracle's k-means (f32 running sum per cluster in row order, then the division:
    kmeans.rs:371-446) lands on the stored centroid bit for bit; a float64 mean does not."""
    base = os.path.join(ref_index_dir(), "v0.21.0_legacy")
    cent, cb, lengths, raw = _legacy_index(os.path.join(base, "index_256.idx"))
    x = IF.read_column(os.path.join(base, "data_256.lance"), "vector", np.float32, 16)
    assert lengths == [256] and cent.shape == (1, 16) and x.shape == (256, 16)
    c, _, _, _ = oracle.kmeans_train(x, 1, max_iters=50, seed=123)
    assert np.array_equal(c.view(np.uint32), cent.view(np.uint32))
    assert not np.array_equal(x.mean(0, dtype=np.float64).astype(np.float32).view(np.uint32), cent[0].view(np.uint32))
    # and the stored codes of both the 256-row index and the 32-row delta are the oracle's residual + encode
    for idx_name, data_name, n in (("index_256.idx", "data_256.lance", 256), ("index_delta32.idx", "data_32.lance", 32)):
        cent2, cb2, lengths2, raw2 = _legacy_index(os.path.join(base, idx_name))
        assert lengths2 == [n] and np.array_equal(cent2, cent) and np.array_equal(cb2, cb)
        xs = IF.read_column(os.path.join(base, data_name), "vector", np.float32, 16)
        codes = np.frombuffer(raw2[:n], np.uint8)
        rid = np.frombuffer(raw2[n:n + 8 * n], np.uint64)
        rows = (rid & np.uint64(0xFFFFFFFF)).astype(np.int64)          # row address = fragment << 32 | offset
        res = oracle.residual(xs[rows], cent, np.zeros(n, np.uint32))
        assert np.array_equal(oracle.pq_encode(res, cb, "l2")[:, 0], codes)


def test_oracle_reproduces_reference_ivf4_pq16_assignment_and_codes():
    """v0.8.14_ivf4_pq16.npz (in tests/golden/ref_index.npz): two IVF_PQ indices Lance 0.8.14 built over 1000 / 2000 rows of
    128-d vectors with 4 partitions and 16 sub-vectors (the SIFT / BASELINE C2 shape: sub-dimension 8).  With the
    reference's own centroids and codebook the oracle must put every row in the partition the reference stored it
    under (argmin over several centroids, kmeans.rs:1350-1369) and produce its 16 code bytes (pq.rs:116-191) --
    3000 rows, 48,000 bytes, all equal; encoding WITHOUT the residual step reproduces only ~90%, so the check bites."""
    z = np.load(os.path.join(ref_index_dir(), "v0.8.14_ivf4_pq16.npz"))
    for k in (0, 1):
        x = z["x"][z[f"rows{k}"]]
        cent, cb = z[f"centroids{k}"], z[f"codebook{k}"]
        part, _ = oracle.assign(x, cent, "l2")
        assert np.array_equal(part, z[f"part{k}"])
        codes = oracle.pq_encode(oracle.residual(x, cent, part), cb, "l2")
        assert np.array_equal(codes, z[f"codes{k}"])
        assert (oracle.pq_encode(x, cb, "l2") == z[f"codes{k}"]).mean() < 0.97
        # the whole transform chain + per-partition storage the oracle builds == the reference's grouping
        oidx = oracle.build_index(x, cent, cb)
        assert np.array_equal(oidx.part_offsets, np.concatenate([[0], np.cumsum(np.bincount(z[f"part{k}"], minlength=4))]))


def _as_dir(tmp_path, src, name):
    d = tmp_path / name
    d.mkdir()
    shutil.copyfile(src, d / "index.idx")
    return d


def test_native_reader_opens_legacy_v1_indices(tmp_path):
    """Index directories of Lance <= 0.21 hold one legacy-format index.idx (pb Index behind a 16-byte footer, per
    partition [row-major codes][row ids]).  The native reader returns the same arrays as the independent Python
    extraction behind the npz / the test-side parser, and `write` upgrades them to the current two-file layout."""
    base = ref_index_dir()
    z = np.load(os.path.join(base, "v0.8.14_ivf4_pq16.npz"))
    for k, name in enumerate(("index_1000.idx", "index_2000.idx")):          # repeated-float codebook, 4 partitions
        c = IF.read_index_files(_as_dir(tmp_path, os.path.join(base, "v0.8.14_legacy", name), f"a{k}"))
        assert (c.index_type, c.metric, c.dtype, c.num_sub_vectors, c.nbits, c.transposed) == ("IVF_PQ", "l2", "float32", 16, 8, False)
        assert np.array_equal(c.centroids, z[f"centroids{k}"]) and np.array_equal(c.codebook, z[f"codebook{k}"])
        assert np.array_equal(c.codes_row_major(), z[f"codes{k}"]) and np.array_equal(c.part_ids(), z[f"part{k}"])
    for name, n in (("index_256.idx", 256), ("index_delta32.idx", 32)):      # tensor codebook, 1 partition
        src = os.path.join(base, "v0.21.0_legacy", name)
        c = IF.read_index_files(_as_dir(tmp_path, src, name[:-4]))
        cent, cb, lengths, raw = _legacy_index(src)
        assert np.array_equal(c.centroids, cent) and np.array_equal(c.codebook, cb) and c.part_offsets.tolist() == [0, n]
        assert c.codes.tobytes() == raw[:n] and c.row_ids.tobytes() == raw[n:9 * n]
        IF.write_index_files(tmp_path / (name[:-4] + "_v3"), c)              # upgrade: transposed storage, two files
        up = IF.read_index_files(tmp_path / (name[:-4] + "_v3"))
        assert up.transposed and np.array_equal(up.codes_row_major(), c.codes_row_major()) and np.array_equal(up.row_ids, c.row_ids)
        assert np.array_equal(up.codebook, c.codebook) and np.array_equal(up.centroids, c.centroids)
    # truncations of a legacy file are errors, not crashes
    raw = open(os.path.join(base, "v0.21.0_legacy", "index_256.idx"), "rb").read()
    for cut in (2304 + 20, 2000, 30):
        d = tmp_path / f"cut{cut}"
        d.mkdir()
        (d / "index.idx").write_bytes(raw[:cut] + raw[-16:])
        with pytest.raises(_lib.LanceHipError):
            IF.read_index_files(d)


# ---- writing -----------------------------------------------------------------------------------------------------------
def _same(a: IF.IndexFileContents, b: IF.IndexFileContents):
    assert (a.index_type, a.metric, a.dtype, a.num_sub_vectors, a.nbits, a.loss) == (b.index_type, b.metric, b.dtype, b.num_sub_vectors, b.nbits, b.loss)
    for k in ("centroids", "part_offsets", "row_ids", "codebook", "codes", "vectors"):
        x, y = getattr(a, k), getattr(b, k)
        assert (x is None) == (y is None), k
        if x is not None:
            assert x.dtype == y.dtype and np.array_equal(x.view(np.uint8), y.view(np.uint8)), k


def test_rewrite_of_reference_index_matches_its_bytes(tmp_path):
    """Re-writing the parsed reference index must reproduce the reference's own bytes wherever the format pins them:
    data pages, page/column metadata, the IVF messages, the centroid tensor, the footer.  (Differences that remain:
    the codebook moves from schema metadata to a global buffer, as current Lance writes it; the schema of 0.27 carried
    a storage-class attribute current Lance dropped; metadata map order is a HashMap's.)"""
    c = IF.read_index_files(REF27)
    out = tmp_path / "idx"
    IF.write_index_files(out, c)
    _same(IF.read_index_files(out), c)
    ref_aux, new_aux = Probe(os.path.join(REF27, "auxiliary.idx")), Probe(out / "auxiliary.idx")
    ref_idx, new_idx = Probe(os.path.join(REF27, "index.idx")), Probe(out / "index.idx")
    assert (new_aux.major, new_aux.minor, new_aux.ncol) == (ref_aux.major, ref_aux.minor, ref_aux.ncol)
    for col in (0, 1):
        assert new_aux.page_bytes(col) == ref_aux.page_bytes(col)
        rp, np_ = ref_aux.pages[col][0], new_aux.pages[col][0]
        assert (np_["length"], np_["sizes"], np_["encoding"], np_["priority"]) == (rp["length"], rp["sizes"], rp["encoding"], rp["priority"])
        assert np_["offsets"][0] % 64 == 0
    assert new_aux.global_buffer(1) == ref_aux.global_buffer(1)             # pb IVF{offsets, lengths}
    assert new_aux.length == ref_aux.length == 512
    ref_meta = json.loads(json.loads(ref_aux.metadata["storage_metadata"])[0])
    new_meta = json.loads(json.loads(new_aux.metadata["storage_metadata"])[0])
    assert new_aux.global_buffer(new_meta["codebook_position"]) == bytes(ref_meta["codebook_tensor"])   # same pb Tensor
    assert {k: new_meta[k] for k in ("nbits", "num_sub_vectors", "dimension", "transposed")} == {k: ref_meta[k] for k in ("nbits", "num_sub_vectors", "dimension", "transposed")}
    assert new_aux.metadata["distance_type"] == ref_aux.metadata["distance_type"] and new_aux.metadata["lance:ivf"] == ref_aux.metadata["lance:ivf"]
    # index.idx: identical IVF message (offsets, lengths, centroid tensor, loss), column metadata and schema metadata
    assert new_idx.global_buffer(1) == ref_idx.global_buffer(1)
    assert new_idx.column_meta == ref_idx.column_meta
    assert new_idx.metadata == ref_idx.metadata
    assert [f[2] for f in new_idx.schema_fields] == [f[2] for f in ref_idx.schema_fields] == [b"__flat_marker"]
    assert [(f[2], f[5]) for f in new_aux.schema_fields] == [(f[2], f[5]) for f in ref_aux.schema_fields]
    # padding byte and alignment of the reference writer (writer.rs:42-44)
    raw = new_aux.b
    assert raw[7:64] == bytes([72]) * 57          # the 7-byte IVF buffer is padded to 64 with 'H'


def _random_pq(rng, n, d, nlist, m, nbits, dtype="float32"):
    part = np.sort(rng.integers(0, nlist, n)).astype(np.uint32)
    if nlist > 2:
        part[part == 1] = 0       # an empty partition
    offs = np.zeros(nlist + 1, np.uint32)
    np.cumsum(np.bincount(part, minlength=nlist), out=offs[1:])
    cb = m // 2 if nbits == 4 else m
    codes_rm = rng.integers(0, 256, (n, cb), dtype=np.uint8)
    codes = np.concatenate([codes_rm[offs[p]:offs[p + 1]].T.reshape(-1) for p in range(nlist)]) if n else np.empty(0, np.uint8)
    cast = (lambda a: a.astype(np.float16).astype(np.float32)) if dtype == "float16" else (lambda a: a)
    return IF.IndexFileContents(
        index_type="IVF_PQ", metric="cosine", dtype=dtype, centroids=cast(rng.standard_normal((nlist, d)).astype(np.float32)),
        part_offsets=offs, row_ids=rng.permutation(n).astype(np.uint64) + 7, codebook=cast(rng.standard_normal((m, 1 << nbits, d // m)).astype(np.float32)),
        codes=codes, num_sub_vectors=m, nbits=nbits, transposed=True, loss=float(rng.random())), codes_rm


@pytest.mark.parametrize("dtype,nbits", [("float32", 8), ("float16", 8), ("float32", 4)])
def test_pq_roundtrip(tmp_path, dtype, nbits):
    rng = np.random.default_rng(3)
    c, codes_rm = _random_pq(rng, 1000, 32, 7, 8, nbits, dtype)
    IF.write_index_files(tmp_path / "i", c)
    back = IF.read_index_files(tmp_path / "i")
    _same(back, c)
    assert np.array_equal(back.codes_row_major(), codes_rm)
    assert np.array_equal(back.part_ids(), np.repeat(np.arange(7, dtype=np.uint32), np.diff(c.part_offsets.astype(np.int64))))
    t = {a: v for a, _, v in fields(Probe(tmp_path / "i" / "index.idx").global_buffer(1))}
    tensor = {a: v for a, _, v in fields(t[4])}
    assert tensor[1] == (1 if dtype == "float16" else 2) and len(tensor[3]) == 7 * 32 * (2 if dtype == "float16" else 4)


def test_multi_page_columns(tmp_path, monkeypatch):
    """Columns cut into several pages (the reference's encoder flushes a page every few MiB) read back identically."""
    monkeypatch.setenv("LANCE_HIP_MAX_PAGE_BYTES", "1000")
    rng = np.random.default_rng(4)
    c, _ = _random_pq(rng, 900, 16, 5, 4, 8)
    IF.write_index_files(tmp_path / "i", c)
    p = Probe(tmp_path / "i" / "auxiliary.idx")
    assert len(p.pages[0]) == 8 and len(p.pages[1]) == 4          # 125 row ids / 250 codes per page
    assert [pg["priority"] for pg in p.pages[1]] == [0, 250, 500, 750]
    assert all(pg["offsets"][0] % 64 == 0 for col in p.pages for pg in col)
    _same(IF.read_index_files(tmp_path / "i"), c)


def test_empty_index_roundtrip(tmp_path):
    rng = np.random.default_rng(5)
    c, _ = _random_pq(rng, 0, 16, 3, 4, 8)
    IF.write_index_files(tmp_path / "i", c)
    back = IF.read_index_files(tmp_path / "i")
    assert back.part_offsets.tolist() == [0, 0, 0, 0] and back.row_ids.size == 0 and back.codes.size == 0
    assert np.array_equal(back.codebook, c.codebook)


@pytest.mark.parametrize("dtype", ["float32", "float16"])
def test_ivf_flat_roundtrip(tmp_path, dtype):
    rng = np.random.default_rng(6)
    n, d, nlist = 300, 24, 4
    offs = np.array([0, 100, 100, 250, 300], np.uint32)
    vec = rng.standard_normal((n, d)).astype(dtype)
    c = IF.IndexFileContents(index_type="IVF_FLAT", metric="l2", dtype=dtype,
                             centroids=rng.standard_normal((nlist, d)).astype(dtype).astype(np.float32), part_offsets=offs,
                             row_ids=np.arange(n, dtype=np.uint64)[::-1].copy(), vectors=vec, loss=None)
    IF.write_index_files(tmp_path / "f", c)
    back = IF.read_index_files(tmp_path / "f")
    assert back.index_type == "IVF_FLAT" and back.loss is None and back.num_sub_vectors == 0
    _same(back, c)
    p = Probe(tmp_path / "f" / "auxiliary.idx")
    assert json.loads(json.loads(p.metadata["storage_metadata"])[0]) == {"dim": d}
    assert p.schema_fields[1][5] == (b"fixed_size_list:halffloat:24" if dtype == "float16" else b"fixed_size_list:float:24")
    assert json.loads(Probe(tmp_path / "f" / "index.idx").metadata["lance:index"]) == {"type": "IVF_FLAT", "distance_type": "l2"}


@pytest.mark.parametrize("dtype,code,width", [("float16", 1, 2), ("float32", 2, 4)])
def test_fsl_to_tensor_like_reference(tmp_path, dtype, code, width):
    """lance-index/src/vector/utils.rs:298-320 `test_fsl_to_tensor`: a 4 x 5 zero matrix becomes pb Tensor{data_type,
    shape [4, 5], data 20 * width bytes} -- here through the writer's centroid tensor (f64 tensors are not written:
    the engine's models are f32 / f16)."""
    c = IF.IndexFileContents(index_type="IVF_FLAT", metric="l2", dtype=dtype, centroids=np.zeros((4, 5), np.float32),
                             part_offsets=np.zeros(5, np.uint32), row_ids=np.empty(0, np.uint64), vectors=np.empty((0, 5), dtype))
    IF.write_index_files(tmp_path / "t", c)
    ivf = {fn: v for fn, _, v in fields(Probe(tmp_path / "t" / "index.idx").global_buffer(1))}
    tensor = {fn: v for fn, _, v in fields(ivf[4])}
    from lance_file_probe import packed
    assert tensor[1] == code and packed(tensor[2]) == [4, 5] and len(tensor[3]) == 20 * width


# ---- refusing what is not understood ---------------------------------------------------------------------------------
def _open_err(path):
    with pytest.raises(_lib.LanceHipError) as e:
        IF.read_index_files(path)
    return e.value


def test_errors_are_reported_not_crashes(tmp_path):
    assert _open_err(tmp_path / "missing").code == _lib.EIO
    d = tmp_path / "bad"
    shutil.copytree(REF27, d)
    raw = (d / "auxiliary.idx").read_bytes()
    (d / "auxiliary.idx").write_bytes(raw[:-4] + b"XXXX")
    assert "magic" in str(_open_err(d))
    (d / "auxiliary.idx").write_bytes(raw[:-8] + b"\x02\x00\x01\x00LANC")      # version 2.1
    assert "2.1" in str(_open_err(d))
    (d / "auxiliary.idx").write_bytes(raw[:100])
    assert _open_err(d).code == _lib.EIO
    (d / "auxiliary.idx").write_bytes(raw[:20])
    assert "too small" in str(_open_err(d))
    # index type the engine does not implement
    (d / "auxiliary.idx").write_bytes(raw)
    idx = (d / "index.idx").read_bytes()
    (d / "index.idx").write_bytes(idx.replace(b'"IVF_PQ"', b'"IVF_SQ"'))
    assert _open_err(d).code == _lib.ENOTSUP
    # the two files must belong together
    d2 = tmp_path / "mixed"
    shutil.copytree(REF27, d2)
    shutil.copyfile(os.path.join(REF29, "index.idx"), d2 / "index.idx")
    assert "dimension" in str(_open_err(d2)) or "disagree" in str(_open_err(d2))


def test_reader_survives_corrupted_metadata(tmp_path):
    """Byte flips anywhere behind the data pages (descriptor, column metadata, tables, footer): every outcome is either
    a parsed index or an error code -- never a crash or an out-of-bounds read (the reader bounds-checks every offset)."""
    rng = np.random.default_rng(11)
    raw = bytearray(open(os.path.join(REF29, "auxiliary.idx"), "rb").read())
    meta_start = 3136          # first byte behind the page buffers of this fixture (global buffer 0)
    d = tmp_path / "fz"
    shutil.copytree(REF29, d)
    lib = _lib.load()
    outcomes = {0: 0}
    for _ in range(300):
        b = bytearray(raw)
        for pos in rng.integers(meta_start, len(b), rng.integers(1, 4)):
            b[pos] = rng.integers(0, 256)
        (d / "auxiliary.idx").write_bytes(bytes(b))
        h = C.c_void_p()
        rc = lib.lance_hip_index_file_open(os.fspath(d).encode(), C.byref(h))
        outcomes[rc] = outcomes.get(rc, 0) + 1
        if rc == 0:
            v = _lib.IndexFileView()
            assert lib.lance_hip_index_file_get(h, C.byref(v)) == 0 and v.n_rows == 256
            lib.lance_hip_index_file_close(h)
        else:
            assert rc in (_lib.EINVAL, _lib.EIO, _lib.ENOTSUP) and lib.lance_hip_last_error()
    assert sum(outcomes.values()) == 300 and len(outcomes) > 1
    # the same for the legacy (v1) reader: flips inside the pb Index message and the footer
    raw = bytearray(open(os.path.join(ref_index_dir(), "v0.21.0_legacy", "index_256.idx"), "rb").read())
    d2 = tmp_path / "fz_legacy"
    d2.mkdir()
    seen = set()
    for _ in range(300):
        b = bytearray(raw)
        for pos in rng.integers(2304, len(b), rng.integers(1, 4)):
            b[pos] = rng.integers(0, 256)
        (d2 / "index.idx").write_bytes(bytes(b))
        h = C.c_void_p()
        rc = lib.lance_hip_index_file_open(os.fspath(d2).encode(), C.byref(h))
        seen.add(rc)
        if rc == 0:
            lib.lance_hip_index_file_close(h)
        else:
            assert rc in (_lib.EINVAL, _lib.EIO, _lib.ENOTSUP)
    assert 0 in seen and len(seen) > 1


def test_reader_survives_corrupted_multi_page_metadata(tmp_path, monkeypatch):
    """The same on columns cut into several pages: flips inside the column metadata (page lengths, buffer offsets and sizes,
    priorities), the offset tables and the footer of a multi-page auxiliary.idx.  A wrapped `length * row_bytes` or row count
    must be caught by the overflow-checked arithmetic of lance_file.cpp, not lead to a read past the mapping; values that stay
    consistent simply decode to different rows."""
    monkeypatch.setenv("LANCE_HIP_MAX_PAGE_BYTES", "1000")
    rng = np.random.default_rng(21)
    c, _ = _random_pq(rng, 900, 16, 5, 4, 8)
    IF.write_index_files(tmp_path / "i", c)
    aux = tmp_path / "i" / "auxiliary.idx"
    raw = bytearray(aux.read_bytes())
    p = Probe(aux)
    data_end = max(pg["offsets"][i] + pg["sizes"][i] for col in p.pages for pg in col for i in range(len(pg["offsets"])))
    assert len(p.pages[0]) > 1 and data_end < len(raw) - 40
    lib = _lib.load()
    outcomes = {}
    for it in range(400):
        b = bytearray(raw)
        if it % 4 == 0:
            # targeted: overwrite a varint inside a column's metadata with a huge value (page length / buffer size wrap-around)
            ccol = int(rng.integers(0, p.ncol))
            pos, sz = struct.unpack_from("<QQ", raw, p.cmo + 16 * ccol)
            at = int(pos + rng.integers(0, max(sz - 10, 1)))
            b[at:at + 10] = bytes([0xFF] * 9 + [0x01])              # 2^64 - 1 as a varint
        else:
            for q in rng.integers(data_end, len(b), rng.integers(1, 4)):
                b[q] = rng.integers(0, 256)
        aux.write_bytes(bytes(b))
        h = C.c_void_p()
        rc = lib.lance_hip_index_file_open(os.fspath(tmp_path / "i").encode(), C.byref(h))
        outcomes[rc] = outcomes.get(rc, 0) + 1
        if rc == 0:
            v = _lib.IndexFileView()
            assert lib.lance_hip_index_file_get(h, C.byref(v)) == 0
            if v.n_rows:      # every byte the view exposes must be readable
                np.ctypeslib.as_array(C.cast(v.row_ids, C.POINTER(C.c_uint64)), shape=(v.n_rows,)).sum()
            lib.lance_hip_index_file_close(h)
        else:
            assert rc in (_lib.EINVAL, _lib.EIO, _lib.ENOTSUP) and lib.lance_hip_last_error()
    assert sum(outcomes.values()) == 400 and len(outcomes) > 1 and any(rc != 0 for rc in outcomes)


def test_write_validates_arguments(tmp_path):
    rng = np.random.default_rng(8)
    c, _ = _random_pq(rng, 50, 16, 3, 4, 8)
    c.part_offsets = c.part_offsets.copy()
    c.part_offsets[-1] -= 1
    with pytest.raises(_lib.LanceHipError, match="offsets"):
        IF.write_index_files(tmp_path / "x", c)


def test_shuffle_buffers_as_a_lance_file_roundtrip(tmp_path):
    """lance_hip_shuffle_buffer_write: (row_id, __ivf_part_id, __pq_code) rows as a Lance v2.0 file (python/lance/vector.py:659-665
    schema), read back column by column with the native reader; rows without a partition are dropped, un-transposed codes."""
    from lance_amd import arrow_io
    rng = np.random.default_rng(3)
    n, m = 70000, 16          # > one 32 MiB page? no: 70000 x 16 B -- several reads of the same page; big enough for two u64 pages? one
    part = rng.integers(0, 256, n).astype(np.uint32)
    part[rng.random(n) < 0.01] = 0xFFFFFFFF
    codes = rng.integers(0, 256, (n, m)).astype(np.uint8)
    rid = rng.permutation(n).astype(np.uint64) + 5
    path = str(tmp_path / "shuffle_0.lance")
    written = arrow_io.write_shuffle_buffers_lance(path, rid, part, codes)
    keep = part != 0xFFFFFFFF
    assert written == int(keep.sum())
    r, p, c = arrow_io.read_shuffle_buffers_lance(path)
    assert (r == rid[keep]).all() and (p == part[keep]).all() and (c == codes[keep]).all()
    # the Arrow route yields the same rows
    got = [b for b in arrow_io.shuffle_buffer_batches(rid, part, codes, batch_size=9000)]
    assert sum(b.num_rows for b in got) == written
    assert (np.concatenate([b.column(0).to_numpy() for b in got]) == r).all()
    # int32 partition ids (what the device returns), implicit row ids
    written2 = arrow_io.write_shuffle_buffers_lance(str(tmp_path / "s2.lance"), None, part.view(np.int32), codes)
    r2, _, _ = arrow_io.read_shuffle_buffers_lance(str(tmp_path / "s2.lance"))
    assert written2 == written and (r2 == np.nonzero(keep)[0].astype(np.uint64)).all()
