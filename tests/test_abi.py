"""CPU-side checks of the drop-in boundary: the C-ABI library loads without a GPU, exports
every symbol include/lance_hip.h declares, and fails loudly (no fallback) without a device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from lance_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        g.build()
    return _lib.load()


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "lance_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(lance_hip_[a-z0-9_]+)\s*\(", hdr)))


def test_header_symbols_are_exported(lib):
    from lance_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/lance_hip.h but not exported"
    assert sorted(_lib.SYMBOLS) == syms, "python binding list out of sync with the header"


def test_no_device_fails_loudly(lib):
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.lance_hip_ctx_create(0, None, C.byref(h))
    assert rc < 0
    assert b"HIP" in lib.lance_hip_last_error() or b"device" in lib.lance_hip_last_error()
    import lance_amd
    with pytest.raises(RuntimeError):
        lance_amd.create_index(__import__("numpy").zeros((10, 8), "float32"), num_partitions=2, num_sub_vectors=2)


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under lance_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lance_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".cuh")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in txt and "lance_oracle" not in txt and "oracle/" not in txt, os.path.join(dirpath, f)


def test_bench_and_smoke_refuse_to_run_without_a_gpu():
    """bench.py and __graft_entry__.smoke() measure / check the HIP path only: without a device they must stop with a
    clear message instead of timing or checking anything on the CPU."""
    import subprocess
    import sys
    from conftest import has_gpu
    if has_gpu():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "MI355X" in (r.stderr + r.stdout) and "{" not in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "__graft_entry__.py"), "smoke"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_header_is_plain_c_and_usable_from_c(tmp_path):
    """include/lance_hip.h compiles as strict C99 and a C program linked against liblance_hip.so drives the index-file
    entry points (open / view / write / re-open / column read / error channel) -- the shape a cgo or Rust `extern "C"`
    binding has.  Host-only calls: no GPU needed."""
    import subprocess
    from lance_amd import _lib
    src = os.path.join(ROOT, "tests", "c", "index_file_roundtrip.c")
    exe = str(tmp_path / "ifr")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), src, "-o", exe,
           "-L", libdir, "-llance_hip", f"-Wl,-rpath,{libdir}", "-Wl,-rpath-link,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    from ref_fixtures import ref_index_dir
    ref = os.path.join(ref_index_dir(), "v0.27.1_pq_in_schema")
    r = subprocess.run([exe, ref, str(tmp_path / "out"), os.path.join(ref, "data.lance")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "d=32 nlist=1 m=4 nbits=8 rows=512" in r.stdout and "loss=1394.7242410182953" in r.stdout
    assert "rows=512 row_bytes=128" in r.stdout and r.stdout.strip().endswith("ok")


def test_native_index_io_glue_under_sanitizers(tmp_path):
    """lance_hip_index_load / _load_lists / _save are the one piece of new native code that needs a device to run.  Their
    host logic (chunked uploads through the pinned staging buffer, list-shard packing, f16 narrowing, export -> files) is
    compiled here for the CPU against a malloc-backed HIP stand-in (tests/c/hip_shim) with host stand-ins for the three
    engine calls it makes, and driven under AddressSanitizer + UBSan on a reference-written index, a legacy one, a
    synthetic f16 / 4-bit one and an IVF_FLAT one -- whole and as 2- and 3-way list shards, with 1000-byte staging chunks."""
    import shutil
    import subprocess
    from ref_fixtures import ref_index_dir
    exe = str(tmp_path / "io_harness")
    csrc = os.path.join(ROOT, "lance_amd", "csrc")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I", os.path.join(ROOT, "tests", "c", "hip_shim"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "index_io_host_harness.cpp"), os.path.join(csrc, "index_file.cpp"),
           os.path.join(csrc, "lance_file.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    if r.returncode != 0 and "sanitize" in r.stderr:
        pytest.skip("g++ without sanitizer runtimes")
    assert r.returncode == 0, r.stderr[-3000:]
    legacy = tmp_path / "legacy"
    legacy.mkdir()
    shutil.copyfile(os.path.join(ref_index_dir(), "v0.8.14_legacy", "index_2000.idx"), legacy / "index.idx")
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    env = dict(os.environ, LANCE_HIP_STAGE_CHUNK="1000", ASAN_OPTIONS="detect_leaks=1")
    r = subprocess.run([exe, os.path.join(ref_index_dir(), "v0.27.1_pq_in_schema"), str(legacy), str(scratch)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]
    assert "largest chunk 1000 bytes" in r.stdout


@pytest.mark.gpu
def test_c99_program_drives_the_device_entry_points(tmp_path, oracle):
    """tests/c/device_pipeline.c: a plain-C99 program (no torch, no HIP headers) trains, encodes, builds the storage layout on
    the host, creates the index from it and searches -- through lance_hip_malloc / memcpy_h2d / the entry points in the call
    order of integration/rust/lance-linalg/src/hip.rs.  Everything it writes out is compared with the oracle bit for bit."""
    import subprocess
    import numpy as np
    from lance_amd.testing import sift_like
    f32 = np.float32
    n, d, nlist, m, nq, k, nprobes, refine, ivf_iters, pq_iters, seed = 30000, 64, 32, 16, 500, 10, 9, 5, 8, 6, 17
    x = sift_like(n, d, seed=5)
    q = sift_like(nq, d, seed=6)
    allow = (np.random.default_rng(3).random(n) < 0.3).astype(np.uint8)
    inp, outp, exe = tmp_path / "in.bin", tmp_path / "out.bin", tmp_path / "device_pipeline"
    with open(inp, "wb") as fh:
        np.array([n, d, nlist, m, nq, k, nprobes, refine, ivf_iters, pq_iters, seed, 0], np.uint32).tofile(fh)
        x.tofile(fh); q.tofile(fh); allow.tofile(fh)
    libdir = os.path.join(ROOT, "lance_amd")
    r = subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "device_pipeline.c"),
                        "-o", str(exe), "-L", libdir, "-l:liblance_hip.so", "-Wl,-rpath," + libdir], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([str(exe), str(inp), str(outp)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]
    raw = np.fromfile(outp, np.uint8)
    pos = 0

    def take(count, dt):
        nonlocal pos
        a = raw[pos:pos + count * np.dtype(dt).itemsize].view(dt)
        pos += count * np.dtype(dt).itemsize
        return a
    cent = take(nlist * d, f32).reshape(nlist, d)
    cb = take(256 * d, f32).reshape(m, 256, d // m)
    part = take(n, np.uint32)
    codes = take(n * m, np.uint8).reshape(n, m)
    probes = take(nq * nprobes, np.uint32).reshape(nq, nprobes)
    res = [(take(nq * k, np.uint64).reshape(nq, k), take(nq * k, f32).reshape(nq, k)) for _ in range(3)]
    assert pos == raw.size
    samp = x[: nlist * 256]
    oc, _, _, _ = oracle.kmeans_train(samp, nlist, max_iters=ivf_iters, balance_factor=f32(1.0) / f32(samp.shape[0]), seed=seed)
    assert (cent.view(np.uint32) == oc.view(np.uint32)).all(), "IVF centroids"
    opart, _ = oracle.assign(x, oc)
    ores = oracle.residual(x[:65536], oc, opart[:65536])
    ocb, _ = oracle.pq_train(ores, m, max_iters=pq_iters, seed=seed + 1)
    assert (cb.view(np.uint32) == ocb.view(np.uint32)).all(), "PQ codebook"
    oidx = oracle.build_index(x, oc, ocb, "l2")
    assert (part == oidx.part_ids).all() and (codes == oidx.codes_rowmajor).all()
    pi, _ = oracle.find_partitions(q, oc, nprobes)
    assert (probes == pi).all()
    for (gi, gd), kw in zip(res, (dict(), dict(refine=refine, raw=x), dict(prefilter=allow.astype(bool)))):
        oi, od = oidx.search(q, k, nprobes, **kw)
        assert (gi == oi).all() and (gd.view(np.uint32) == od.view(np.uint32)).all(), kw.keys()
