"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/np_hip.h declares, the ctypes prototypes cover all of them, and the host layer's
argument checking / marshalling works without a GPU (no compute calls here)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared(header: Path, pattern: str):
    text = re.sub(r"/\*.*?\*/", "", header.read_text(), flags=re.S)
    return sorted(set(re.findall(pattern, text)))


def test_np_hip_exports_every_declared_symbol():
    from numpower_amd import _lib
    lib = _lib.load()
    product = _declared(ROOT / "include" / "np_hip.h", r"\b(np_\w+)\s*\(")
    probes = _declared(ROOT / "include" / "np_hip_debug.h", r"\b(np_\w+)\s*\(")
    # the product header holds no kernel-variant switch and no probe: those live in np_hip_debug.h (same library)
    assert not [n for n in product if n.endswith("_set_variant") or "_debug_" in n or n.startswith("np_debug_") or n == "np_select_last_path"]
    assert all(n.endswith("_set_variant") or "_debug_" in n or n.startswith("np_debug_") or n in ("np_select_last_path", "np_comm_sync_mode")
               for n in probes), probes
    assert len(product) >= 35 and len(probes) >= 14
    names = sorted(set(product) | set(probes))
    for n in names:
        assert hasattr(lib, n), "libnp_hip.so does not export %s" % n
        assert n in _lib.PROTOTYPES, "numpower_amd/_lib.py has no prototype for %s" % n
    assert set(_lib.PROTOTYPES) <= set(names)


def test_host_library_exports_every_declared_symbol():
    from numpower_amd import ndarray
    h = ndarray._load_host()
    names = _declared(ROOT / "include" / "numpower_host.h",
                      r"\b(NDArray\w+|reduce|numpower_host_\w+)\s*\(")
    names = [n for n in names if not n.startswith("NDArray_FDATA") and n not in (
        "NDArray_NDIM", "NDArray_SHAPE", "NDArray_NUMELEMENTS", "NDArray_DEVICE", "NDArray_ADDREF")]
    assert len(names) >= 40
    for n in names:
        assert hasattr(h, n), "libnumpower_host.so does not export %s" % n


def test_enum_values_match_header():
    """The Python-side op tables are positional; pin them against the header's enums."""
    from numpower_amd import _lib
    text = (ROOT / "include" / "np_hip.h").read_text()
    m = re.search(r"typedef enum np_unary_op \{(.*?)\} np_unary_op;", text, re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    names = [t.split("=")[0].strip() for t in body.split(",") if t.strip()]
    names = [n for n in names if n != "NP_UNARY_OP_COUNT"]
    assert [n[3:].lower() for n in names] == list(_lib.UNARY_OPS)
    from oracle import oracle
    assert list(oracle.UNARY) == list(_lib.UNARY_OPS)
    assert oracle.BINARY == _lib.BINARY_OPS and oracle.REDUCE == _lib.REDUCE_OPS


def test_avx_body_end_helper():
    from numpower_amd import _lib
    lib = _lib.load()
    for n in range(0, 70):
        i = 0
        while i < n - 7:   # the reference's loop header, arithmetics.c:251
            i += 8
        assert lib.np_avx_body_end(n) == i


def test_no_device_is_a_loud_error():
    """Without a GPU every device entry point fails with the reference's message; nothing falls
    back to the host."""
    from numpower_amd import _lib
    lib = _lib.load()
    n = C.c_int()
    if lib.np_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    p = C.c_void_p()
    assert lib.np_malloc(C.byref(p), 1024) != 0
    assert b"No GPU device available" in lib.np_last_error()
    from numpower_amd.ndarray import Error, NDArray
    with pytest.raises(Error, match="No GPU device available or CUDA not enabled"):
        NDArray.array([[1, 2], [3, 4]]).gpu()
    with pytest.raises(Error, match="only computes on the GPU"):
        NDArray.array([[1, 2], [3, 4]]) + 2


def test_host_marshalling_on_cpu():
    from numpower_amd.ndarray import NDArray
    a = NDArray.array([[1, 2, 3], [4, 5, 6]])
    assert a.shape() == [2, 3] and a.size() == 6 and a.ndim() == 2 and not a.isGPU()
    assert a.toArray() == [[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]
    assert a[1].toArray() == [4.0, 5.0, 6.0]
    assert a[1][2] == 6.0          # 0-d view -> float (RETURN_NDARRAY)
    row = a[0]
    del a                           # the view holds its base (NDArray_ADDREF)
    assert row.toArray() == [1.0, 2.0, 3.0]
    z = NDArray.zeros([2, 2])
    z.fill(7.0)
    assert z.toArray() == [[7.0, 7.0], [7.0, 7.0]]


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under numpower_amd/ may reference it."""
    banned = re.compile(r"import\s+oracle|from\s+oracle|libnp_oracle|np_oracle|oracle_\w+\s*\(|oracle/")
    for path in (ROOT / "numpower_amd").rglob("*"):
        if path.suffix in (".py", ".hip", ".cpp", ".h") and path.name != "build.py":
            assert not banned.search(path.read_text()), "%s references the oracle" % path


def test_views_and_initializers_on_cpu_arrays():
    """Metadata-only host functions work on CPU-resident arrays too (no kernel is needed): reshape /
    flatten / expand_dims / contiguous slices are views or plain copies, full / identity are plain
    stores; anything that would need arithmetic or a gather on the host raises."""
    from numpower_amd.ndarray import Error, NDArray
    x = np.arange(24, dtype=np.float32).reshape(4, 6)
    a = NDArray.array(x)
    r = NDArray.reshape(a, [3, 8])
    assert not r.isGPU() and r.toArray() == x.reshape(3, 8).tolist()
    assert NDArray.flatten(a).toArray() == x.reshape(-1).tolist()
    assert NDArray.expand_dims(a, [0, -1]).shape() == [1, 4, 6, 1]
    assert a.slice([1, 3]).toArray() == x[1:3].tolist()          # contiguous rows: a view
    assert a.slice([2]).toArray() == x[2].tolist()
    assert a.slice([-1]).toArray() == x[3].tolist()
    assert a.slice([1, 3]).slice([0, 1]).toArray() == x[1:2].tolist()
    with pytest.raises(Error, match="only computes on the GPU"):
        a.slice([0, 4, 2])                                        # strided: would need a gather
    with pytest.raises(Error, match="too many indices for array."):
        a.slice([0], [0], [0])
    with pytest.raises(Error, match="slice step cannot be zero"):
        a.slice([0, 2, 0])
    with pytest.raises(Error, match="incompatible shape in reshape call."):
        NDArray.reshape(a, [5, 5])
    with pytest.raises(Error, match="invalid axis or axes provided."):
        NDArray.expand_dims(a, 7)
    assert NDArray.full([2, 3], 1.5).toArray() == [[1.5] * 3] * 2
    assert NDArray.ones([3]).toArray() == [1.0, 1.0, 1.0]
    assert NDArray.identity(2).toArray() == [[1.0, 0.0], [0.0, 1.0]]
    assert NDArray.identity(0).shape() == [0]
    with pytest.raises(Error, match="negative dimensions are not allowed"):
        NDArray.identity(-1)
    with pytest.raises(Error, match="only computes on the GPU"):
        NDArray.arange(10, 0, 1, 0)
    with pytest.raises(Error, match="arange: zero length"):
        NDArray.arange(0, 5, 1, 0)
    # compute entry points refuse CPU arrays with the same message everywhere
    for fn in (lambda: NDArray.maximum(a, a), lambda: NDArray.array_equal(a, a), lambda: NDArray.diagonal(a),
               lambda: NDArray.outer(NDArray.array(x[0]), NDArray.array(x[1])), lambda: a.contiguous()):
        with pytest.raises(Error, match="only computes on the GPU"):
            fn()


def test_manipulation_and_order_stat_argument_checks_on_cpu_arrays():
    """The manipulation wrappers that are pure views work on CPU arrays; the ones that move data, and
    median / quantile, check their arguments first (the reference's messages) and then refuse CPU data
    loudly instead of computing on the host."""
    from numpower_amd.ndarray import Error, NDArray
    x = np.arange(12, dtype=np.float32).reshape(1, 3, 1, 4)
    a = NDArray.array(x)
    assert NDArray.squeeze(a).shape() == [3, 4] and NDArray.squeeze(a, 0).shape() == [3, 1, 4]
    assert NDArray.squeeze(a, [0, 2]).toArray() == x.reshape(3, 4).tolist()
    assert NDArray.atleast_1d(NDArray.array(np.float32(2.0))).shape() == [1]
    assert NDArray.atleast_2d(NDArray.array(np.float32([1, 2, 3]))).shape() == [1, 3]
    assert NDArray.atleast_3d(NDArray.array(np.float32([1, 2, 3]))).shape() == [1, 3, 1]
    assert NDArray.atleast_3d(NDArray.array(np.zeros((2, 5), np.float32))).shape() == [2, 5, 1]
    with pytest.raises(Error, match="cannot select an axis to squeeze out which has size not equal to one"):
        NDArray.squeeze(a, 1)
    with pytest.raises(Error, match="duplicate value in 'axis'"):
        NDArray.squeeze(a, [0, 0])
    m = NDArray.array(np.zeros((2, 3), np.float32))
    v = NDArray.array(np.zeros((3,), np.float32))
    with pytest.raises(Error, match="same number of dimensions"):
        NDArray.concatenate([m, v], 0)
    with pytest.raises(Error, match="along dimension 1, the array at index 0 has size 3 and the array at index 1 has size 2"):
        NDArray.concatenate([m, NDArray.array(np.zeros((2, 2), np.float32))], 0)
    with pytest.raises(Error, match="zero-dimensional arrays cannot be concatenated"):
        NDArray.concatenate([NDArray.array(np.float32(1.0))], 0)
    with pytest.raises(Error, match="Axis is out of bounds for array dimension"):
        NDArray.swapaxes(m, 0, 2)
    with pytest.raises(Error, match="must have the same number of elements"):
        NDArray.moveaxis(m, [0, 1], [0])
    with pytest.raises(Error, match="Input array must be a vector or 2-dimensional"):
        NDArray.diag(a)
    with pytest.raises(Error, match="Q must be between 0 and 1"):
        NDArray.quantile(v, 2.0)
    with pytest.raises(Error, match="Q must be a scalar"):
        NDArray.quantile(v, [0.1, 0.2])
    for fn in (lambda: NDArray.concatenate([m, m], 0), lambda: NDArray.vstack([v, v]), lambda: NDArray.swapaxes(m, 0, 1),
               lambda: NDArray.diag(v), lambda: NDArray.median(v), lambda: NDArray.quantile(v, 0.5)):
        with pytest.raises(Error, match="only computes on the GPU"):
            fn()


def test_tuning_knobs_validate_their_argument_without_a_device():
    """np_*_set_variant are plain setters (no device work): a bad code is an error with a message, a good one is accepted,
    and the defaults are restored — on a box without a GPU too."""
    import ctypes as C
    from numpower_amd import _lib
    lib = _lib.load()
    lib.np_last_error.restype = C.c_char_p
    assert lib.np_select_set_variant(-1) != 0 and b"np_select_set_variant" in lib.np_last_error()
    for ok in (0, 2048, 767, 1):
        assert lib.np_select_set_variant(ok) == 0
    assert lib.np_runtime_set_variant(3) != 0 and b"np_runtime_set_variant" in lib.np_last_error()
    assert lib.np_runtime_set_variant(-1) != 0
    for ok in (0, 1, 2):
        assert lib.np_runtime_set_variant(ok) == 0
    assert lib.np_select_last_path(None) != 0      # null output: refused before any device call
    # the GEMM planner's switches are setters too (A/B partners of the default forms): accepted and restored
    for code in (-4, -5, -2, -6, -8, -7, -9, -11, -10, -12, -13, -3, -2, 0):
        assert lib.np_sgemm_set_variant(code) == 0
    assert lib.np_sgemm_set_variant(1000) != 0 and b"tuning builds" in lib.np_last_error()      # ablations: tuning builds only
    for code in (9000, 0):
        assert lib.np_elementwise_set_variant(code) == 0
    # a piece larger than the batch it claims to belong to: refused before any device call
    assert lib.np_sgemm_strided_batched_piece(3, 2, 8, 8, 8, None, 64, None, 64, None, 64) != 0
    assert b"a piece of 3 matrices of a batch of 2" in lib.np_last_error()


def test_gemm_planner_choices_on_a_256_cu_device():
    """np_sgemm_debug_plan: the planner is host arithmetic (np_sgemm.hip plan_sgemm / streamk_model), so what a 256-CU device
    would run is checkable here.  Pinned: the forms the measurements in profiles/r04 (gemm_plans.log, gemm_kdeep_ab.log) stand
    on.  cfg 0 = 256 x 128 LDS-DMA tiles, 1 / 2 = register-staged 128 x 128 / 64 x 64, 3 / 4 / 5 = the mid-size LDS-DMA tiles
    128 x 128 / 128 x 64 / 64 x 64, 6 .. 12 = the k-quartered tiles."""
    from numpower_amd import _lib
    lib = _lib.load()
    out = (C.c_double * 11)()

    def plan(m, n, k, batch=1):
        assert lib.np_sgemm_debug_plan(m, n, k, batch, 256, out) == 0, lib.np_last_error()
        o = list(out)
        return {"cfg": int(o[0]), "tail_rows": int(o[1]), "S": int(o[2]), "us": o[3], "streamk": bool(o[4]), "mid_cfg": int(o[6]), "other_cfg": int(o[9])}

    p = plan(4096, 4096, 4096)          # the headline: whole-K 256 x 128 tiles, two per CU, no stream-K
    assert (p["cfg"], p["tail_rows"], p["S"], p["streamk"]) == (0, 0, 1, False) and 850 < p["us"] < 1100
    # the k-quartered tiles (sgemm_kq_kernel): cfg 6 + shape, shapes 48x48, 32x32, 64x64, 48x32, 64x32, 64x48, 80x48 — the one whose
    # tile count fits the 256 CUs best
    for shape, cfg in (((256,) * 3, 7), ((512,) * 3, 7), ((576,) * 3, 9), ((640,) * 3, 10), ((768,) * 3, 6), ((768, 768, 3072), 6), ((832,) * 3, 11),
                       ((896,) * 3, 12), ((1000,) * 3, 8), ((1024,) * 3, 8), ((256, 4096, 4096), 8), ((4096, 256, 4096), 8), ((1024, 1024, 4096), 8),
                       ((512, 512, 4096), 7), ((760,) * 3, 6)):
        p = plan(*shape)
        assert (p["cfg"], p["tail_rows"], p["S"], p["streamk"]) == (cfg, 0, 1, False), (shape, p)
    for shape in ((1152,) * 3, (1280,) * 3, (1536,) * 3):   # several co-resident rounds of small tiles beat the larger tiles' ragged rounds
        p = plan(*shape)
        assert p["cfg"] >= 6 and (p["tail_rows"], p["S"], p["streamk"]) == (0, 1, False), (shape, p)
    assert plan(762, 762, 762)["cfg"] == 6 and plan(333, 333, 333)["cfg"] == 7      # rows that are not float4-loadable: the same tiles on dword loads
    assert plan(1001, 1003, 1002)["cfg"] in (5, 8) and plan(100, 100, 2)["cfg"] < 6    # (a tie with the LDS-DMA 64 x 64 tiles; K < 4 is not for that kernel)
    p = plan(1000, 1000, 100000)       # a few tiles and a very deep K: K is split, one way or another
    assert p["S"] >= 2 or p["streamk"] or p["tail_rows"] > 0
    for shape in ((2560,) * 3, (3072,) * 3):   # tile counts that leave a ragged last round: stream-K
        assert plan(*shape)["streamk"], shape
    # a dot-product-like shape: K cut into chunks on the k-quartered tiles, one workgroup per CU, folded by a second launch
    # (round 5, profiles/r05/gemm_deep_k_ab.log; until then the register-staged 64 x 64 tiles in 128 chunks)
    for shape in ((100, 100, 100000), (128, 128, 65536), (64, 64, 100000), (256, 256, 32768), (300, 300, 20000)):
        p = plan(*shape)
        assert p["cfg"] >= 6 and p["tail_rows"] > 0 and p["S"] >= 2 and not p["streamk"], (shape, p)
    # ... unless an operand reaches 4 GiB: the k-quartered kernel's byte offsets are 32-bit, its launcher declines, so the planner
    # must not pick it — the same K-chunking on the register-staged tiles, sized for THEM
    for shape in ((100, 100, 11_000_000), (8192, 64, 200_000)):
        p = plan(*shape)
        assert p["cfg"] < 6 and (p["S"] >= 2 or p["streamk"]), (shape, p)
    lib.np_sgemm_set_variant(-22)
    try:
        p = plan(100, 100, 100000)
        assert p["cfg"] == 2 and p["tail_rows"] > 0 and p["S"] >= 64
    finally:
        lib.np_sgemm_set_variant(-23)
    p = plan(1024, 1024, 1024, 512)    # BASELINE config 5: a batch is never split along K
    assert p["S"] == 1 and p["tail_rows"] == 0 and not p["streamk"]
    assert lib.np_sgemm_debug_plan(0, 4, 4, 1, 256, out) != 0 and lib.np_sgemm_debug_plan(4, 4, 4, 1, 256, None) != 0


def test_gemm_deep_k_switches_validate_and_restore():
    """The round-5 planner switches of include/np_hip_debug.h are host state: each code is accepted or refused as documented, and
    the forced K-chunk plan shows in np_sgemm_debug_plan exactly while it is on (no device needed)."""
    from numpower_amd import _lib
    lib = _lib.load()
    lib.np_last_error.restype = C.c_char_p
    out = (C.c_double * 11)()
    assert lib.np_sgemm_set_variant(-(30000 + 1000 * 7 + 4)) != 0 and b"no such shape" in lib.np_last_error()
    try:
        assert lib.np_sgemm_set_variant(-(30000 + 1000 * 1 + 16)) == 0          # 32 x 32 tiles, 16 chunks
        assert lib.np_sgemm_debug_plan(512, 512, 4096, 1, 256, out) == 0
        assert (int(out[0]), int(out[1]), int(out[2])) == (7, 16, 16), list(out)
        assert lib.np_sgemm_debug_plan(512, 512, 4096, 4, 256, out) == 0       # batches are never K-chunked
        assert int(out[1]) == 0
    finally:
        assert lib.np_sgemm_set_variant(-30000) == 0
    assert lib.np_sgemm_debug_plan(512, 512, 4096, 1, 256, out) == 0
    assert (int(out[0]), int(out[1]), int(out[2])) == (7, 0, 1), list(out)
    for code in (-22, -23, -24, -23, -25, -26, -40, -(40 + 59)):
        assert lib.np_sgemm_set_variant(code) == 0, code


def test_gemm_planner_is_total_on_random_shapes():
    """Whatever the shape, np_sgemm_debug_plan answers with a plan the launchers know: a cfg in range, a positive finite model
    time, no second-launch fold or stream-K for batches, no k-quartered tiles below K = 4 — 3000 random shapes from 1 to 20000 per dimension."""
    from numpower_amd import _lib
    lib = _lib.load()
    out = (C.c_double * 11)()
    rng = np.random.default_rng(5)
    for _ in range(3000):
        m, n, k = (int(10 ** rng.uniform(0, 4.3)) for _ in range(3))
        batch = int(rng.choice([1, 1, 1, 2, 64]))
        assert lib.np_sgemm_debug_plan(m, n, k, batch, 256, out) == 0, (m, n, k, batch, lib.np_last_error())
        cfg, tail, S, us, sk = int(out[0]), int(out[1]), int(out[2]), out[3], bool(out[4])
        assert 0 <= cfg <= 12 and S >= 1 and tail >= 0 and 0.0 < us < 1e9, (m, n, k, batch, list(out))
        if batch > 1:     # batches: no second-launch fold and no stream-K; K may be split INSIDE the launch of the mid-size tiles
            assert tail == 0 and not sk and (S == 1 or 3 <= cfg <= 5), (m, n, k, batch, list(out))
        if k < 4:
            assert cfg < 6, (m, n, k, list(out))
        if cfg >= 6:      # whole K, or (single deep-K products of a few tiles) every tile row cut into K-chunks
            assert (S == 1 and tail == 0) or (batch == 1 and S >= 2 and tail > 0 and k >= 2048), (m, n, k, list(out))


def test_comm_entry_points_without_a_communicator_or_device():
    """The overlapped-gather entry points refuse politely before any device work when no communicator exists, the piece
    arithmetic is pure, and the communicator's tuning knob validates its argument — on a box without a GPU too."""
    import ctypes as C
    from numpower_amd import _lib
    lib = _lib.load()
    lib.np_last_error.restype = C.c_char_p
    assert lib.np_comm_world() == 0 and lib.np_comm_rank() == -1 and lib.np_comm_sync_mode() == -1
    assert not lib.np_comm_stream()
    for call in (lambda: lib.np_allgather_async(1, 2, 4, 4, 0), lambda: lib.np_comm_wait(),
                 lambda: lib.np_sgemm_strided_batched_allgather(4, 8, 8, 8, 1, 64, 2, 64, 3, 2, 0),
                 lambda: lib.np_comm_debug_sendrecv_self(1, 2, 4), lambda: lib.np_comm_debug_loopback(None, 0),
                 lambda: lib.np_comm_debug_loopback_timed(1, 2, 4, 1, (C.c_float * 1)())):
        assert call() != 0 and b"no communicator" in lib.np_last_error()
    lo, count = C.c_size_t(0), C.c_size_t(0)
    got = []
    for c in range(3):
        assert lib.np_comm_piece(64, 3, c, C.byref(lo), C.byref(count)) == 0
        got.append((lo.value, count.value))
    assert got == [(0, 22), (22, 21), (43, 21)]
    assert lib.np_comm_piece(64, 0, 0, C.byref(lo), C.byref(count)) != 0 and b"np_comm_piece" in lib.np_last_error()
    assert lib.np_comm_piece(64, 3, 3, C.byref(lo), C.byref(count)) != 0
    assert lib.np_comm_piece(64, 3, 0, None, C.byref(count)) != 0
    assert lib.np_comm_set_variant(4) != 0 and b"np_comm_set_variant" in lib.np_last_error()
    for ok in (1, 2, 3, 0):
        assert lib.np_comm_set_variant(ok) == 0


def test_loopback_stand_in_for_rccl_is_test_infrastructure_and_complete():
    """tests/loopback_rccl/lib/librccl.so.1 (ranks that share one GPU: tests/test_gpu_comm_loopback_peers.py) exports every entry
    point np_comm.hip looks up in the collective library, answers ncclGetVersion with its own number without a device — and
    nothing of the product names it: np_comm.hip asks the loader for "librccl.so.1", a test's workers get this one through
    LD_LIBRARY_PATH."""
    import ctypes as C
    lib_path = ROOT / "tests" / "loopback_rccl" / "lib" / "librccl.so.1"
    assert lib_path.exists(), "not built: python -m numpower_amd.build"
    wanted = re.findall(r'NP_SYM\(\w+, "(ncc\w+)"\)', (ROOT / "numpower_amd" / "csrc" / "np_comm.hip").read_text())
    wanted += ["ncclCommAbort", "ncclGetVersion"]                      # the two np_comm.hip treats as optional
    assert len(wanted) >= 12
    lib = C.CDLL(str(lib_path))
    for name in wanted:
        assert hasattr(lib, name), name
    version = C.c_int(0)
    assert lib.ncclGetVersion(C.byref(version)) == 0 and version.value == 99901
    for path in list((ROOT / "numpower_amd").rglob("*")) + list((ROOT / "ext").rglob("*")) + [ROOT / "bench.py", ROOT / "__graft_entry__.py"]:
        if path.suffix in (".py", ".hip", ".cpp", ".h", ".c") and path.name != "build.py":
            assert "loopback_rccl" not in path.read_text(), "%s names the tests' stand-in for RCCL" % path
