"""The C-ABI library loads on a CPU-only box and exports every symbol include/egopack_hip.h declares
(no compute calls without a GPU)."""
import ctypes

from egopack_amd import _lib


def test_library_is_built_and_loads():
    lib = _lib.load()
    assert lib.egk_version() >= 100
    assert isinstance(_lib.last_error(), str)


def test_every_header_symbol_is_exported_and_bound():
    lib = _lib.load()
    declared = _lib.header_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/egopack_hip.h but not exported"
    assert set(declared) == set(_lib.SIGNATURES), set(declared) ^ set(_lib.SIGNATURES)


def test_gemm_descriptor_layout_matches_header():
    # field order / count of struct egk_gemm_desc as declared in the header
    import re
    text = _lib.HEADER.read_text()
    body = re.search(r"typedef struct egk_gemm_desc \{(.*?)\} egk_gemm_desc;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            names.append(re.findall(r"[A-Za-z_][A-Za-z0-9_]*", part)[-1])
    assert names == [f[0] for f in _lib.GemmDesc._fields_]


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.load()
    assert lib.egk_gemm(None, None) == -1  # EGK_EINVAL: null descriptor
    assert "null descriptor" in _lib.last_error()
    assert lib.egk_prof_count() > 30
    assert lib.egk_gemm_splitk(128, 256, 8192, 1) > 1 and lib.egk_gemm_splitk(6144, 1024, 1024, 1) == 1
    assert lib.egk_rowln_bwd_ws_rows(6144) >= 1 and lib.egk_graphln_ws_bytes(6144, 1024, 3) > 0
    name = ctypes.create_string_buffer(64)
    n, ms, fl, by = ctypes.c_int64(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    assert lib.egk_prof_get(0, name, 64, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) == 0
    assert name.value.decode().startswith("gemm")
    # the grouped max aggregation: null pointers and a group count outside 1 .. 4
    buf = (ctypes.c_void_p * 4)(1, 1, 1, 1)
    assert lib.egk_gather_max_group_fwd(None, None, buf, buf, 2, None, None, 8, 256, 4, 1) == -1 and "null pointer" in _lib.last_error()
    one = ctypes.c_void_p(1)
    assert lib.egk_gather_max_group_fwd(None, one, buf, buf, 5, one, one, 8, 256, 4, 1) == -1 and "1 .. 4 groups" in _lib.last_error()
    assert lib.egk_gather_max_group_fwd(None, one, buf, buf, 3, one, one, 0, 256, 4, 1) == 0  # (no rows: nothing launched)
    prev = lib.egk_gather_max_tune(-1)
    assert prev in (0, 1) and lib.egk_gather_max_tune(0) == prev and lib.egk_gather_max_tune(prev) == 0


def test_heavy_row_threshold_is_shared_by_the_csr_builder_and_the_kernel():
    """data.build_csr lists EXACTLY the rows the gather kernel leaves to its split launches."""
    from egopack_amd import _lib, data
    assert _lib.load().egk_csr_heavy_threshold() == data.HEAVY_DEGREE
    import torch
    ei = torch.stack([torch.zeros(100, dtype=torch.long), torch.arange(1, 101)])  # node 0 -> 100 targets
    g = data.build_csr(ei, 101)
    assert g.t_heavy.tolist() == [0] and g.heavy.numel() == 0 and g.t_heavy.dtype == torch.int32
    ei = torch.stack([torch.zeros(data.HEAVY_DEGREE, dtype=torch.long), torch.arange(1, data.HEAVY_DEGREE + 1)])
    assert data.build_csr(ei, data.HEAVY_DEGREE + 1).t_heavy.numel() == 0  # exactly the threshold: not listed
