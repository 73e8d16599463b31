"""CPU tests of the C-ABI boundary: the library loads, exports every symbol include/*.h
declares, and its host-side argument handling works without a GPU (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from rnnt_amd import engine
    if not os.path.exists(engine.LIB_PATH):
        engine.build()
    return engine.lib()


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "rnnt_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rnnt_engine_\w+)\s*\(", text)))


def test_header_symbols_exported(lib):
    names = _declared_symbols()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rnnt_engine.h but not exported"


def test_python_binding_lists_every_entry_point():
    from rnnt_amd import engine
    declared = set(_declared_symbols())
    assert set(engine.EXPORTS) <= declared


def test_version_and_workspace_queries(lib):
    assert lib.rnnt_engine_version() >= 1
    n = ctypes.c_size_t(0)
    assert lib.rnnt_engine_workspace_bytes(32, 1000, 201, 512, 1024, 0, ctypes.byref(n)) == 0
    logits = 32 * 1000 * 201 * 1024 * 4
    assert logits < n.value < 2 * logits  # logits + hidden + slabs, never a second logits copy
    m = ctypes.c_size_t(0)
    assert lib.rnnt_engine_loss_workspace_bytes(2, 5, 3, 8, 0, ctypes.byref(m)) == 0 and m.value > 0
    assert lib.rnnt_engine_joint_fwd_workspace_bytes(2, 5, 3, 16, 8, 0, ctypes.byref(m)) == 0


def test_error_codes_and_messages(lib):
    lib.rnnt_engine_last_error.restype = ctypes.c_char_p
    n = ctypes.c_size_t(0)
    assert lib.rnnt_engine_workspace_bytes(0, 5, 3, 16, 8, 0, ctypes.byref(n)) == -1
    assert b"non-positive" in lib.rnnt_engine_last_error()
    assert lib.rnnt_engine_workspace_bytes(2, 5, 3, 16, 6, 0, ctypes.byref(n)) == -2      # V % 4
    assert lib.rnnt_engine_workspace_bytes(2, 5, 3, 18, 8, 0, ctypes.byref(n)) == -2      # H % 4
    assert lib.rnnt_engine_workspace_bytes(2, 5, 2000, 16, 8, 0, ctypes.byref(n)) == -2   # U1
    assert lib.rnnt_engine_workspace_bytes(2, 5, 3, 16, 8, 7, ctypes.byref(n)) == -2      # dtype
    assert lib.rnnt_engine_workspace_bytes(2, 5, 3, 16, 8, 0, None) == -1
    # null pointers are rejected before anything is launched
    strides = (ctypes.c_int64 * 3)(80, 16, 1)
    rc = lib.rnnt_engine_joint_loss_fwd_bwd(None, strides, None, None, None, None, None, None, 2, 5,
                                            3, 16, 8, 7, ctypes.c_float(-1), ctypes.c_float(0.5), 0,
                                            None, None, None, None, None, None, ctypes.c_size_t(0),
                                            None)
    assert rc == -1 and b"null" in lib.rnnt_engine_last_error()


def test_layout_is_consistent(lib):
    from rnnt_amd import engine
    L = engine.layout(3, 17, 9, 64, 32)
    offs = [L.logits, L.hidden, L.denom_s, L.lpb_s, L.lpe_s, L.alpha_s, L.beta_s, L.coef, L.wpack,
            L.enc_copy, L.slab_enc, L.slab_pred, L.slab_w, L.slab_b, L.counters, L.total]
    assert offs == sorted(offs) and all(o % 256 == 0 for o in offs)
    assert L.D == 17 + 9 - 1 and L.n_ublk == 1 and L.n_ttile == 5
    assert L.rows_pad % 16 == 0 and L.rows_pad > 3 * 17 * 9
    assert L.total == engine.workspace_bytes(3, 17, 9, 64, 32)


def test_bf16_dtype_queries_and_validation(lib):
    """RNNT_DTYPE_BF16 (include/rnnt_engine.h): fused entry only, H % 128, V % 128 (any H: 512-column passes)."""
    from rnnt_amd import engine
    lib.rnnt_engine_last_error.restype = ctypes.c_char_p
    n = ctypes.c_size_t(0)
    assert lib.rnnt_engine_workspace_bytes(32, 1000, 201, 512, 1024, 1, ctypes.byref(n)) == 0
    f32 = engine.workspace_bytes(32, 1000, 201, 512, 1024, "fp32")
    assert n.value == engine.workspace_bytes(32, 1000, 201, 512, 1024, "bf16") < f32  # bf16 hidden
    for H, V in ((64, 128), (600, 1024), (512, 1000)):
        assert lib.rnnt_engine_workspace_bytes(2, 5, 3, H, V, 1, ctypes.byref(n)) == -2
        assert b"RNNT_DTYPE_BF16" in lib.rnnt_engine_last_error()
    assert lib.rnnt_engine_workspace_bytes(2, 5, 3, 1024, 1024, 1, ctypes.byref(n)) == 0  # the reference's joint width
    assert lib.rnnt_engine_workspace_bytes(2, 5, 3, 640, 1024, 1, ctypes.byref(n)) == 0
    # the standalone loss / joint entries have no bf16 variant
    assert lib.rnnt_engine_loss_workspace_bytes(2, 5, 3, 128, 1, ctypes.byref(n)) == -2
    L = engine.layout(3, 17, 9, 128, 256, "bf16")
    assert L.rows_pad % 32 == 0 and L.rows_pad > 3 * 17 * 9
    assert engine.dtype_code("bf16") == engine.DTYPE_BF16 == 1 and engine.dtype_code("fp32") == 0
    import torch
    assert engine.dtype_code(torch.bfloat16) == 1
    with pytest.raises(ValueError):
        engine.dtype_code("fp8")


def test_greedy_scan_entry_validation(lib):
    lib.rnnt_engine_last_error.restype = ctypes.c_char_p
    n = ctypes.c_size_t(0)
    assert lib.rnnt_engine_greedy_scan_workspace_bytes(32, 512, 1024, ctypes.byref(n)) == 0
    assert n.value >= 32 * 1024 * 4 + 32 * 512 * 4
    assert lib.rnnt_engine_greedy_scan_workspace_bytes(129, 512, 1024, ctypes.byref(n)) == -1
    assert lib.rnnt_engine_greedy_scan_workspace_bytes(8, 510, 1024, ctypes.byref(n)) == -2
    assert lib.rnnt_engine_greedy_scan_workspace_bytes(8, 508, 1024, ctypes.byref(n)) == -2  # H % 8
    rc = lib.rnnt_engine_greedy_scan(None, ctypes.c_int64(512), ctypes.c_int64(1), None, None, None, 0, 8, 512,
                                     1024, 1023, None, None, ctypes.c_size_t(0), None)
    assert rc == -1 and b"null" in lib.rnnt_engine_last_error()


def test_allreduce_entry_validation(lib):
    """rnnt_engine_allreduce refuses null arguments before it looks for RCCL (no GPU, no RCCL call)."""
    lib.rnnt_engine_last_error.restype = ctypes.c_char_p
    assert lib.rnnt_engine_allreduce(None, ctypes.c_size_t(8), None, None) == -1
    assert b"null" in lib.rnnt_engine_last_error()


def _header_signatures():
    """{name: kinds} from the prototypes of include/rnnt_engine.h (i int, q int64_t, z size_t, f float,
    d double, p pointer / array)."""
    text = open(os.path.join(ROOT, "include", "rnnt_engine.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    protos = re.findall(r"(?:^|\n)\s*(?:const\s+)?(?:int|void|char)\s*\*?\s*(rnnt_engine_\w+)\s*\(([^;{]*?)\)\s*;",
                        text, flags=re.S)
    scalar = {"int": "i", "int32_t": "i", "int64_t": "q", "size_t": "z", "float": "f", "double": "d"}
    out = {}
    for name, args in protos:
        kinds = ""
        for a in args.split(","):
            a = " ".join(a.split())
            if a == "void":
                continue
            if "*" in a or "[" in a:
                kinds += "p"
            else:
                words = a.replace("const ", "").split()
                kinds += scalar[" ".join(words[:-1]) if len(words) > 1 else words[0]]
        out[name] = kinds
    return out


def test_ctypes_signatures_match_the_header(lib):
    """Every declared entry point has typed ctypes bindings, and the table in rnnt_amd/engine.py is the
    header's prototypes (argument count and kind), so a prototype edit cannot drift silently."""
    from rnnt_amd import engine
    hdr = _header_signatures()
    assert set(hdr) == set(_declared_symbols())
    assert engine.SIGNATURES == hdr
    for name, kinds in hdr.items():
        assert len(getattr(lib, name).argtypes) == len(kinds), name


def test_typed_bindings_reject_bad_scalars(lib):
    """a dimension passed as a non-int, or beyond 2^31, raises instead of being truncated."""
    n = ctypes.c_size_t(0)
    with pytest.raises(ctypes.ArgumentError):
        lib.rnnt_engine_workspace_bytes(2.0, 5, 3, 16, 8, 0, ctypes.byref(n))
    with pytest.raises(ctypes.ArgumentError):
        lib.rnnt_engine_workspace_bytes(2 ** 31 + 2, 5, 3, 16, 8, 0, ctypes.byref(n))
    with pytest.raises(ctypes.ArgumentError):
        lib.rnnt_engine_workspace_bytes(True, 5, 3, 16, 8, 0, ctypes.byref(n))
    with pytest.raises(ctypes.ArgumentError):
        lib.rnnt_engine_workspace_bytes(2, 5, 3, 16, 8, 0, 3.5)  # not a pointer
    with pytest.raises(TypeError):  # too few arguments
        lib.rnnt_engine_workspace_bytes(2, 5, 3, 16, 8, 0)
    assert lib.rnnt_engine_workspace_bytes(2, 5, 3, 16, 8, 0, ctypes.byref(n)) == 0


def test_engine_sources_only_enqueue_kernels():
    """The library's contract (include/rnnt_engine.h: "every call only ENQUEUES work on `stream`", no
    allocation, no synchronisation) and the HIP-graph rule of DESIGN.md §3 (a captured call is a chain of
    KERNEL nodes: fills and copies are kernels, never memset/memcpy nodes) as a source check: none of these
    runtime calls may appear in rnnt_amd/csrc outside comments."""
    banned = re.compile(r"\bhip(MemsetAsync|Memset|MemcpyAsync|Memcpy|Memcpy2D\w*|DeviceSynchronize|StreamSynchronize|"
                        r"EventSynchronize|Malloc\w*|Free\w*|HostMalloc|StreamCreate\w*|GraphLaunch)\s*\(")
    csrc = os.path.join(ROOT, "rnnt_amd", "csrc")
    hits = []
    for fn in sorted(os.listdir(csrc)):
        if not fn.endswith((".hip", ".hpp", ".h", ".cpp")):
            continue
        text = open(os.path.join(csrc, fn)).read()
        text = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), text, flags=re.S)
        for ln, line in enumerate(text.splitlines(), 1):
            code = line.split("//", 1)[0]
            if banned.search(code):
                hits.append(f"{fn}:{ln}: {line.strip()}")
    assert not hits, "runtime calls the engine must not make:\n" + "\n".join(hits)


def test_dw_split_count_follows_the_tile_count(lib):
    """The split-K count of the dW stage is 256 // (workgroup tiles per split), and the f16x2 route's tile count at H % 256 == 128
    (BASELINE config 4: H = 640) is whole-block tiles + one TALL tile per pair of v blocks over the odd 128 columns (k_dw_x2m, round 6:
    10 tiles, 25 splits) — the half-empty third h block of round 5 made it 12 tiles, 21 splits.  Host logic only."""
    from rnnt_amd.engine import WsLayout, dtype_code
    def n_split(H, V, dtype):
        L = WsLayout()
        assert lib.rnnt_engine_workspace_layout(8, 4000, 601, H, V, dtype_code(dtype), ctypes.byref(L)) == 0
        return L.n_split
    assert n_split(640, 1024, "f16x2") == 25          # 4 x 2 whole-block tiles + 2 tall tiles
    assert n_split(640, 1024, "bf16x3") == 21         # (its dW keeps the half-empty block: 12 tiles)
    assert n_split(512, 1024, "f16x2") == 32          # 8 tiles
    assert n_split(896, 1024, "f16x2") == 256 // 14   # 4 x 3 + 2
    assert n_split(640, 768, "f16x2") == 256 // 9     # V % 512 != 0: no tall tiles (3 x 3)
    assert n_split(384, 1024, "f16x2") == 256 // 8    # fewer than two whole h blocks: no tall tiles (4 x 2)
