"""The Rust adapters under integration/rust cannot be compiled here (no rustc / cargo in the image).  What CAN be checked without a
compiler is checked: every `extern "C"` declaration of integration/rust/lance-linalg/src/hip.rs names a function include/lance_hip.h
declares, with the same number of parameters; every `hip::<item>` the lance-index adapters use is defined in that module; and the
adapters call no helper of the reference's k-means types that the reference does not have (`kmeans_hip.rs` called
`KMeansParams::init_centroids_f32` and `KMeans::from_f32_centroids`, which exist nowhere) -- reference items are looked up in
/root/reference when it is present (this container), and in a recorded list of the items used otherwise (the GPU box)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GLUE = os.path.join(ROOT, "integration", "rust")
REF_KMEANS = "/root/reference/rust/lance-index/src/vector/kmeans.rs"

# items of rust/lance-index/src/vector/kmeans.rs the adapters rely on (checked against the file when it is there)
REFERENCE_ITEMS = {
    "KMeans::with_centroids": r"pub fn with_centroids\(",
    "KMeanInit::Incremental": r"Incremental\(Arc<FixedSizeListArray>\)",
    "KMeanInit::Random": r"\bRandom,",
    "KMeansParams.init": r"pub init: KMeanInit",
    "KMeansParams.max_iters": r"pub max_iters: u32",
    "KMeansParams.tolerance": r"pub tolerance: f64",
    "KMeansParams.balance_factor": r"pub balance_factor: f32",
    "KMeansParams.hierarchical_k": r"pub hierarchical_k: usize",
    "KMeansParams.distance_type": r"pub distance_type: DistanceType",
    "KMeansAlgo::compute_membership_and_dist": r"fn compute_membership_and_dist\(",
    "KMeansAlgo::to_kmeans": r"fn to_kmeans\(",
}


def _read(*parts):
    with open(os.path.join(*parts)) as fh:
        return fh.read()


def _header_functions():
    """name -> number of parameters, from include/lance_hip.h"""
    h = re.sub(r"/\*.*?\*/", " ", _read(ROOT, "include", "lance_hip.h"), flags=re.S)
    h = re.sub(r"//[^\n]*", " ", h)
    out = {}
    for m in re.finditer(r"\b(lance_hip_\w+)\s*\(([^;{]*?)\)\s*;", h, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else args.count(",") + 1
    return out


def _split_args(s):
    depth, cur, out = 0, "", []
    for ch in s:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur); cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return out


def test_extern_block_matches_the_header():
    hip_rs = _read(GLUE, "lance-linalg", "src", "hip.rs")
    header = _header_functions()
    decls = re.findall(r"pub fn (lance_hip_\w+)\s*\((.*?)\)\s*(?:->\s*[^;]+)?;", hip_rs, flags=re.S)
    assert len(decls) >= 25
    for name, args in decls:
        assert name in header, f"hip.rs declares {name}, which include/lance_hip.h does not"
        assert len(_split_args(args)) == header[name], f"{name}: {len(_split_args(args))} parameters in hip.rs, {header[name]} in the header"


def test_adapters_only_use_what_exists():
    hip_rs = _read(GLUE, "lance-linalg", "src", "hip.rs")
    defined = set(re.findall(r"pub (?:fn|struct|enum|static|type|const)\s+(\w+)", hip_rs))
    vec = os.path.join(GLUE, "lance-index", "src", "vector")
    for fn in sorted(os.listdir(vec)):
        src = re.sub(r"//[^\n]*", "", _read(vec, fn))      # code only: the doc comments cite reference items by name
        for item in set(re.findall(r"\bhip::(\w+)", src)):
            if item == "self":
                continue
            assert item in defined, f"{fn} uses hip::{item}, which hip.rs does not define"
        for grp in re.findall(r"use lance_linalg::hip::\{([^}]*)\}", src):
            for item in (x.strip() for x in grp.split(",")):
                if item and item != "self":
                    assert item in defined, f"{fn} imports hip::{item}, which hip.rs does not define"
        # method-style calls on the reference's k-means types must be items the reference has, or helpers defined in this file
        local_fns = set(re.findall(r"\bfn (\w+)", src))
        for ty, meth in re.findall(r"\b(KMeans|KMeansParams)::(\w+)\(", src):
            assert f"{ty}::{meth}" in REFERENCE_ITEMS or meth in local_fns, f"{fn} calls {ty}::{meth}: not in the reference"
        for meth in re.findall(r"\bparams\.(\w+)\(", src):
            assert meth in local_fns, f"{fn} calls params.{meth}(): KMeansParams has only public fields"
        for field in re.findall(r"\bparams\.(\w+)\b(?!\()", src):
            assert f"KMeansParams.{field}" in REFERENCE_ITEMS, f"{fn} reads params.{field}: not a field of the reference's KMeansParams"


def test_recorded_reference_items_exist_in_the_reference():
    if not os.path.exists(REF_KMEANS):
        import pytest
        pytest.skip("/root/reference is not present on this machine")
    ref = _read(REF_KMEANS)
    for item, pat in REFERENCE_ITEMS.items():
        assert re.search(pat, ref), f"{item}: pattern {pat!r} not found in {REF_KMEANS}"
