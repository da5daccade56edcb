"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/hnsw_mi355x.h declares, and fails loudly (no CPU fallback) when no device is present."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def H():
    import __graft_entry__ as ge
    ge._load_build_module().build()
    import ocaml_hnsw_amd as H
    H.load()
    return H


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "hnsw_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(hnsw_[a-z_0-9]+)\s*\(", hdr)))


def test_exports_every_declared_symbol(H):
    L = H.load()
    declared = _declared_symbols()
    assert len(declared) >= 12
    assert sorted(H.ABI_SYMBOLS) == declared
    for s in declared:
        assert hasattr(L, s), s
    assert L.hnsw_abi_version() == H.ABI_VERSION == 3


def test_ctypes_mirror_declares_every_argument_list(H):
    """Every entry point that takes arguments has its ctypes argtypes set with the header's arity: without them ctypes
    passes a Python int as a 32-bit C int and a 64-bit pointer arrives truncated (found the hard way: a device fault)."""
    L = H.load()
    hdr = open(os.path.join(ROOT, "include", "hnsw_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    for m in re.finditer(r"\b(hnsw_[a-z_0-9]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        n_args = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        fn = getattr(L, name)
        if n_args == 0:
            continue
        assert fn.argtypes is not None, "%s: no argtypes in ocaml-hnsw_amd/__init__.py" % name
        assert len(fn.argtypes) == n_args, "%s: %d argtypes, the header declares %d arguments" % (name, len(fn.argtypes), n_args)


def test_no_torch_types_in_abi():
    hdr = open(os.path.join(ROOT, "include", "hnsw_mi355x.h")).read()
    assert "torch" not in hdr and "at::" not in hdr and "std::" not in hdr


def test_product_does_not_touch_oracle():
    """The product path must never route through the oracle or any CPU fallback."""
    pkg = os.path.join(ROOT, "ocaml-hnsw_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", ".ml")):
                txt = open(os.path.join(dp, f)).read()
                assert "liboracle" not in txt and "hnsw_oracle" not in txt and "from oracle" not in txt, f
    # ... and outside tests/ only the places the rules allow use it: bench.py's cpu_baseline leg and
    # __graft_entry__ (builds the checker; smoke() checks against it).  tools/ never does.
    import re
    pat = re.compile(r"(from\s+oracle\b|import\s+oracle\b|liboracle|hnsw_oracle)")
    users = []
    for dp, dn, fs in os.walk(ROOT):
        dn[:] = [x for x in dn if x not in (".git", "gpurun_out", "__pycache__", "tests", "oracle", "build")]
        for f in fs:
            if f.endswith((".py", ".sh", ".hip", ".h", ".hpp", ".cpp", ".c")) and pat.search(open(os.path.join(dp, f), errors="ignore").read()):
                users.append(os.path.relpath(os.path.join(dp, f), ROOT))
    assert sorted(users) == ["__graft_entry__.py", "bench.py"], users


def test_fails_loudly_without_device(H):
    if H.device_count() > 0:
        pytest.skip("a device is present")
    X = np.zeros((4, 8), np.float32)
    hg = H.Hgraph(X, np.zeros(4, np.int32), np.full((4, 4), -1, np.int32), entry_point=0)
    with pytest.raises(H.Failure, match="no HIP device"):
        H.Ohnsw.knn_batch_bigarray(hg, 2, X)


def test_argument_validation_is_host_side(H):
    # shapes are checked before anything touches a device
    with pytest.raises(H.InvalidArgument):
        H.Hgraph(np.zeros((4, 8), np.float32), np.zeros(3, np.int32), np.zeros((4, 4), np.int32))


def test_ocaml_binding_binds_every_declared_symbol():
    """ocaml/hnsw_mi355x.ml (source only: the image has no OCaml toolchain) must at least NAME every entry point the
    header declares in a `foreign` binding, and wrap the single-query forms the reference exposes (Ohnsw.knn,
    lib/ohnsw.ml:859-875; Hnsw.Ba.knn, lib/hnsw.ml:763-767)."""
    ml = open(os.path.join(ROOT, "ocaml-hnsw_amd", "ocaml", "hnsw_mi355x.ml")).read()
    missing = [s for s in _declared_symbols() if not re.search(r'foreign[^"]*"%s"' % s, ml)]
    assert not missing, missing
    for wrapper in ("let knn ", "let ba_knn ", "let knn_batch_bigarray ", "let knn_batch ", "let distance_batch ",
                    "let select_neighbours ", "let build ", "let save ", "let load ", "let stats ", "let isolated ", "let stats_compute ",
                    "let pin ", "let unpin ", "let alloc_mat ", "let alloc_ids ", "let scratch_for ", "let export "):
        assert wrapper in ml, wrapper
    # struct field orders match the header (ctypes structures are positional)
    hdr = open(os.path.join(ROOT, "include", "hnsw_mi355x.h")).read()
    for struct, prefix in (("hnsw_search_params", "p_"), ("hnsw_layer_stats", "ls_"), ("hnsw_build_params", "b_")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        c_fields = [f.strip().lstrip("*") for decl in body.split(";") if decl.strip()
                    for f in decl.strip().split(None, 1)[1].split(",")]
        ml_fields = re.findall(r'field %s "([a-z_0-9]+)"' % struct.replace("hnsw_", ""), ml)
        assert ml_fields == c_fields, (struct, ml_fields, c_fields)
