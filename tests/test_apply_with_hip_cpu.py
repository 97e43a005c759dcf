"""tools/apply_with_hip.py — INTEGRATION.md section 2a as a mechanically checkable artefact.

Runs the tool on a scratch copy of the reference checkout (this container only: /root/reference does not exist on
the GPU box, where these tests skip; nothing of the reference is committed or shipped) and asserts that
  * every edit's anchor is found exactly as many times as the tool expects (no anchor missing or ambiguous),
  * the DEFAULT output is a HIP-only tree: no CUDA-runtime / cuBLAS name is left anywhere in the text of a source the
    build compiles (comments and inactive preprocessor branches included), no `#ifdef HAVE_NP_HIP ... #else <CUDA>` pair,
  * with --keep-cuda each replaced statement is kept on the #else side (that tree still builds --with-cuda) and, with
    HAVE_NP_HIP + HAVE_CUBLAS defined, no preprocessor-visible line names the CUDA runtime or cuBLAS,
  * the glue travels with the tree (src/hip/),
and that the tool fails loudly — not silently — when an anchor has moved."""
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference")
sys.path.insert(0, str(ROOT / "tools"))

needs_reference = pytest.mark.skipif(not (REF / "numpower.c").exists(), reason="reference checkout not present on this box")


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    if not (REF / "numpower.c").exists():
        pytest.skip("reference checkout not present on this box")
    import apply_with_hip as tool
    out = tmp_path_factory.mktemp("with_hip") / "numpower"
    applied = tool.apply(REF, out)
    return tool, out, applied


@pytest.fixture(scope="module")
def patched_keep_cuda(tmp_path_factory):
    if not (REF / "numpower.c").exists():
        pytest.skip("reference checkout not present on this box")
    import apply_with_hip as tool
    out = tmp_path_factory.mktemp("with_hip_keep_cuda") / "numpower"
    applied = tool.apply(REF, out, keep_cuda=True)
    return tool, out, applied


def test_every_edit_applies_exactly_as_often_as_expected(patched):
    tool, out, applied = patched
    assert len(applied) == len(tool.EDITS)
    for e in tool.EDITS:
        assert applied[e.what] == e.expect, e.what
    # 7 header swaps + the call sites of INTEGRATION.md section 2a
    assert sum(1 for e in tool.EDITS if "CUDA headers" in e.what) == 8      # + php_numpower.h, which numpower.c includes


def test_the_default_tree_is_hip_only(patched):
    """No CUDA / HIP dual path: the raw text of every source a --with-hip build compiles — the headers numpower.c pulls in
    included — is free of CUDA-runtime and cuBLAS names, and no `#ifdef HAVE_NP_HIP` block has an #else side."""
    tool, out, _ = patched
    assert tool.check_tree(out, raw=True) == []
    compiled = [p for p in list(out.glob("*.c")) + list(out.glob("*.h")) + list(out.glob("src/**/*.c")) + list(out.glob("src/**/*.h"))
                if not p.relative_to(out).as_posix().startswith(tool.REPLACED_BY_GLUE + ("src/hip/",))]
    assert len(compiled) > 30
    for p in compiled:
        text = p.read_text(errors="replace")
        assert not re.search(r"cuda[A-Z]|cublas|<cuda_runtime\.h>", text), p
    # the HAVE_NP_HIP blocks that DO have an #else side are feature guards around device-independent reference code — what a
    # build WITHOUT --with-hip compiles there: reduce() / single_reduce()'s slice loops, exp2's NDArray_Map, and section 2c's
    # appenders (the plain operand lookups and `rtn = NDArray_Add_Float(nda, ndb);` ...).  Each is the reference's own text.
    else_sides = []
    for p in compiled:
        ref_text = re.sub(r"\s+", " ", (REF / p.relative_to(out)).read_text(errors="replace")) if (REF / p.relative_to(out)).exists() else ""
        for m in re.finditer(r"^#ifdef HAVE_NP_HIP\n(.*?)^#endif", p.read_text(errors="replace"), flags=re.S | re.M):
            if "\n#else\n" in m.group(1):
                side = m.group(1).split("\n#else\n")[1].strip()
                else_sides.append(side)
                assert re.sub(r"\s+", " ", side) in ref_text, (p, side)
                assert not tool._CUDA_NAME.search(side)
    assert {"_reduce(0, 0, axis, array, rtn, operation);", "_single_reduce(0, 0, axis, array, rtn, operation);",
            "rtn = NDArray_Map(nda, float_exp2);", "rtn = NDArray_Add_Float(nda, ndb);"} <= set(else_sides)
    assert len(else_sides) == 3 + 43 + 12 + 5 + 6
    # the two things that are left are not compiled by a --with-hip build (config.m4 swaps them for the glue)
    assert (out / "src" / "gpu_alloc.c").exists() and "src/hip/gpu_alloc_hip.c" in (out / "config.m4").read_text()
    # the checker is not vacuous: the --keep-cuda tree fails the raw check at its #else sides
    import apply_with_hip
    assert apply_with_hip.raw_cuda_names("x\n#else\n    cudaSetDevice(deviceId);\n#endif\n") == [(3, "cudaSetDevice(deviceId);")]


def test_no_cuda_runtime_name_is_visible_to_a_hip_build(patched_keep_cuda):
    tool, out, _ = patched_keep_cuda
    assert tool.check_tree(out, raw=False) == []
    assert tool.check_tree(out, raw=True) != []          # ... while its text still holds every CUDA statement
    # the checker is not vacuous: the UNPATCHED tree fails it, at the very call sites the table lists
    problems = []
    for rel in ("numpower.c", "src/ndarray.c", "src/initializers.c", "src/ndmath/arithmetics.c", "src/ndmath/linalg.c", "src/debug.c"):
        problems += ["%s:%d" % (rel, no) for no, _ in tool.hip_visible_cuda_names((REF / rel).read_text())]
    for site in ("src/ndarray.c:1055", "src/ndarray.c:1090", "src/initializers.c:439", "src/initializers.c:443",
                 "src/initializers.c:758", "src/ndmath/arithmetics.c:218", "src/ndmath/arithmetics.c:883",
                 "src/ndmath/linalg.c:68", "numpower.c:623", "numpower.c:633"):
        assert site in problems, site


def test_the_cuda_side_is_kept_verbatim(patched_keep_cuda):
    """--keep-cuda: each wrapped edit leaves the reference's statement on the #else side: with HAVE_NP_HIP undefined the
    patched file preprocesses back to the reference's text."""
    tool, out, _ = patched_keep_cuda

    def without_hip(text):
        keep, stack = [], []          # stack entries: True = inside the HAVE_NP_HIP side of one of OUR blocks, False = its #else side
        for line in text.split("\n"):
            s = line.strip()
            if s == "#ifdef HAVE_NP_HIP":
                stack.append(True)
                continue
            if stack and s == "#else" and stack[-1] is True:
                stack[-1] = False
                continue
            if stack and s == "#endif":          # closes the #else side of a wrapped edit, or an inserted (#else-less) block
                stack.pop()
                continue
            if any(side is True for side in stack):      # (blocks nest where 2c's call edit meets 2a's rsqrt pair)
                continue
            keep.append(line)
        return "\n".join(keep)

    for rel in sorted({e.file for e in tool.EDITS if e.wrap}):
        assert without_hip((out / rel).read_text()) == (REF / rel).read_text(), rel


def test_the_glue_and_the_configure_option_travel_with_the_tree(patched):
    tool, out, _ = patched
    for g in tool.GLUE_FILES:
        assert (out / "src" / "hip" / Path(g).name).read_bytes() == (ROOT / g).read_bytes()
    m4 = (out / "config.m4").read_text()
    assert "PHP_ARG_WITH([hip]" in m4 and "$NP_GPU_ALLOC_SOURCES \\" in m4 and "src/gpu_alloc.c \\" not in m4
    assert 'NP_GPU_ALLOC_SOURCES="src/gpu_alloc.c"' in m4          # the default keeps the CUDA / CPU builds as they were
    for src in re.search(r'NP_GPU_ALLOC_SOURCES="(src/hip/[^"]+)"', m4).group(1).split():
        assert (out / src).exists(), src
    numpower = (out / "numpower.c").read_text()
    # rsqrt / exp2 get their own device functions (section 2a), which section 2c then turns into appenders like the other 34
    assert "NPH_LazyElementWise(nda, cuda_float_rsqrt)" in numpower and "NPH_LazyElementWise(nda, cuda_float_exp2)" in numpower
    assert "void cuda_float_rsqrt(int nblocks, float *d_array);" in numpower


def test_the_new_statements_compile_against_the_c_abi(tmp_path):
    """The HAVE_NP_HIP side of every statement-level edit, each with the locals it uses declared as in the reference,
    compiled on its own with -Wall -Wextra -Werror against include/np_hip.h, include/numpower_host.h and ext/hip_math.h:
    the replacement text is well-typed C against the ABI it calls (needs no reference checkout)."""
    import apply_with_hip as tool
    src = tmp_path / "snippets.c"
    src.write_text(tool.snippet_check_source())
    n = sum(1 for e in tool.EDITS if e.context)
    assert n >= 14 + 12 + 2          # section 2a's statements, the twelve early-outs, reduce(), single_reduce()
    proc = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
                           "-I", str(ROOT / "include"), "-I", str(ROOT / "ext"), str(src)], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    # and the check is not vacuous: a wrong argument order in one snippet is refused
    bad = src.read_text().replace("np_memset0(rtn->data, rtn->descriptor->numElements * sizeof(float));",
                                  "np_memset0(rtn->descriptor->numElements * sizeof(float), rtn->data);")
    assert bad != src.read_text()
    (tmp_path / "bad.c").write_text(bad)
    proc = subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Wno-unused-variable", "-Wno-unused-but-set-variable",
                           "-I", str(ROOT / "include"), "-I", str(ROOT / "ext"), str(tmp_path / "bad.c")], capture_output=True, text=True)
    assert proc.returncode != 0


@needs_reference
def test_a_moved_anchor_is_a_hard_error(tmp_path):
    import apply_with_hip as tool
    broken = tmp_path / "broken"
    shutil.copytree(REF, broken, ignore=shutil.ignore_patterns(".git"))
    text = (broken / "src" / "ndarray.c").read_text()
    assert "cudaMemcpyDeviceToHost);" in text
    (broken / "src" / "ndarray.c").write_text(text.replace("cudaMemcpy(rtn->data, NDArray_FDATA(target)", "cudaMemcpy(rtn->data , NDArray_FDATA(target)"))
    with pytest.raises(tool.PatchError, match=r"ndarray\.c:1090 .* matched 0 time"):
        tool.apply(broken, tmp_path / "out1")
    # ... and so is an ambiguous one (the same statement twice)
    twice = tmp_path / "twice"
    shutil.copytree(REF, twice, ignore=shutil.ignore_patterns(".git"))
    t = (twice / "src" / "initializers.c").read_text()
    line = "            cudaMemset(rtn->data, 0, rtn->descriptor->numElements * sizeof(float));\n"
    assert t.count(line) == 1
    (twice / "src" / "initializers.c").write_text(t.replace(line, line + line))
    with pytest.raises(tool.PatchError, match="matched 2 time"):
        tool.apply(twice, tmp_path / "out2")
    # the command line reports it with exit status 2
    proc = subprocess.run([sys.executable, str(ROOT / "tools" / "apply_with_hip.py"), str(broken), str(tmp_path / "out3")],
                          capture_output=True, text=True)
    assert proc.returncode == 2 and "matched 0 time" in proc.stderr


def test_the_fast_path_inserts_replace_nothing(patched):
    """Section 2b: the GPU early-outs are INSERTED behind each function's own device-mismatch check — no reference symbol
    is replaced, the whole body below (scalar expand, broadcast, AVX2 loop) is still there for CPU operands — and the tree
    compiles ext/hip_fast.c, which holds NPH_Binary_Float / NPH_ReduceAxisInto."""
    tool, out, _ = patched
    assert len(tool.FAST_BINARY) == 12
    for rel, ref_rel in (("src/ndmath/arithmetics.c", "src/ndmath/arithmetics.c"), ("src/logic.c", "src/logic.c")):
        text, ref = (out / rel).read_text(), (REF / ref_rel).read_text()
        n = sum(1 for _, e in tool.FAST_BINARY if e.file == rel)
        assert text.count("if (NPH_TAKES(") == n == 6
        # every line of the reference file that does not name CUDA is still in the patched one, in order (insert-only for
        # these two files' early-outs; section 2a replaced the CUDA includes and the five cudaDeviceSynchronize() calls)
        it = iter(text.split("\n"))
        # (... and, since the end of round 6, the calls that hand a `long` element count to the back end's `int`: cuda_prod_float / cuda_sum_float
        # in arithmetics.c:41,63,86, cuda_equal_float in logic.c:683 — back-end calls, not reference arithmetic: section 2a)
        int_count_call = re.compile(r"\bcuda_(?:prod|sum|equal)_float\(")
        kept = [line for line in ref.split("\n") if not tool._CUDA_NAME.search(line) and not int_count_call.search(line)]
        assert len(kept) >= len(ref.split("\n")) - 10
        assert all(any(line == cand for cand in it) for line in kept), rel
    for name, e in tool.FAST_BINARY:
        text = (out / e.file).read_text()
        head = text.index(name + "(NDArray*")
        body = text[head:head + 1500]
        # order inside the function: the reference's device check, then the early-out, then the scalar expand
        assert body.index("mismatch") < body.index("NPH_TAKES") < body.index("// If a or b are scalars, reshape"), name
    nd = (out / "src/ndarray.c").read_text()
    assert nd.count("NPH_ReduceAxisInto(") == 2 and nd.count(" _reduce(0, 0, axis, array, rtn, operation);") == 2   # HIP side's else + the #else side (a build without --with-hip)
    assert nd.count("_single_reduce(0, 0, axis, array, rtn, operation);") == 2 and "operation == NDArray_Mean_Float" in nd
    assert "src/hip/hip_fast.c" in (out / "config.m4").read_text()
    assert (out / "src/hip/hip_fast.c").exists() and (out / "src/hip/hip_fast.h").exists()


def test_cpu_operands_still_reach_the_reference_code(tmp_path):
    """The inserted text, verbatim, inside functions with the reference's signatures, as a program (the tool generates it):
    with CPU operands — BASELINE config 1 — every call falls through to the stand-in for the reference's own body and
    nothing touches a device.  Needs no reference checkout and no GPU."""
    import apply_with_hip as tool
    from numpower_amd import build
    build.build_all()
    src = tmp_path / "fast_path_bodies.c"
    src.write_text(tool.fast_path_program_source())
    for name, e in tool.FAST_BINARY:
        assert e.new in src.read_text().replace("\n    ", "\n"), name       # the inserted text is in there as the tool holds it
    exe = tmp_path / "fast_path_bodies"
    lib = ROOT / "numpower_amd" / "lib"
    proc = subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), "-I", str(ROOT / "ext"),
                           str(src), "-o", str(exe), "-L", str(lib), "-lnumpower_host", "-lnp_hip", "-Wl,-rpath," + str(lib)],
                          capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    proc = subprocess.run([str(exe), "cpu"], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    m = re.search(r"(\d+) calls, (\d+) reached the reference's own code", proc.stdout)
    assert m and m.group(1) == m.group(2) and int(m.group(1)) >= 80


@needs_reference
def test_hip_fast_uses_only_what_the_reference_headers_declare():
    """ext/hip_fast.c is compiled inside the patched tree against the reference's OWN headers (src/initializers.h -> src/ndarray.h):
    every NDArray_* / NDARRAY_* name it uses must be a macro or a prototype there, with the argument list the call sites assume —
    the one check of that build that needs no PHP."""
    src = re.sub(r"/\*.*?\*/", "", (ROOT / "ext" / "hip_fast.c").read_text(), flags=re.S)
    used = sorted(set(re.findall(r"\b(NDArray_\w+|NDARRAY_\w+)\b", src)) - {"NDArray_Binary"})
    headers = (REF / "src" / "ndarray.h").read_text() + (REF / "src" / "initializers.h").read_text()
    assert {"NDArray_EmptyLike", "NDArray_FREE", "NDArray_IsBroadcastable", "NDArray_NDIM", "NDArray_DEVICE", "NDARRAY_DEVICE_GPU"} <= set(used)
    for name in used:
        assert re.search(r"#define\s+%s\b|\b%s\s*\(" % (name, name), headers), "%s is not declared by the reference's headers" % name
    protos = {"NDArray_EmptyLike": r"NDArray\s*\*\s*NDArray_EmptyLike\(NDArray \*a\);",
              "NDArray_FREE": r"void NDArray_FREE\(NDArray \*array\);",
              "NDArray_IsBroadcastable": r"int NDArray_IsBroadcastable\(const NDArray \*arr1, const NDArray \*arr2\);"}
    for name, rx in protos.items():
        assert re.search(rx, headers), name
    # the same names, the same shapes, in the header the stand-alone build compiles it against
    ours = (ROOT / "include" / "numpower_host.h").read_text()
    for name in used:
        assert re.search(r"#define\s+%s\b|\b%s\s*\(" % (name, name), ours), "%s missing from include/numpower_host.h" % name


def test_pending_chains_flush_at_the_one_marshalling_point(patched):
    """Section 2c: buffer_get — the only function through which a PHP handle becomes an NDArray* — gets the flush; the
    arithmetic operator handler, the six static arithmetic methods and the 36 unary methods look their operands up inside an
    appender scope and append; NOTHING else in numpower.c changes how it marshals, so every other method (the remaining
    ZVAL_TO_NDARRAY / buffer_get call sites) flushes by construction."""
    tool, out, _ = patched
    buf, ref_buf = (out / "src/buffer.c").read_text(), (REF / "src/buffer.c").read_text()
    assert buf.count("NPH_OnBufferGet(MAIN_MEM_STACK.buffer[uuid]);") == 1
    body = buf[buf.index("NDArray* buffer_get(int uuid) {"):]
    assert body.index("assert(") < body.index("NPH_OnBufferGet(") < body.index("return MAIN_MEM_STACK.buffer[uuid];")
    # the object table is read in exactly the places it was read before: buffer_get, and the free / insert paths that must NOT flush
    assert buf.count("MAIN_MEM_STACK.buffer[") == ref_buf.count("MAIN_MEM_STACK.buffer[") + 1
    ref_np, new_np = (REF / "numpower.c").read_text(), (out / "numpower.c").read_text()
    other = [p for p in list(REF.glob("*.c")) + list(REF.glob("src/**/*.c")) if p.name not in ("buffer.c",)]
    assert sum(p.read_text(errors="replace").count("MAIN_MEM_STACK.buffer[") for p in other) == 0      # nobody reads the table behind buffer_get's back
    assert ref_np.count("buffer_get(") == 6 == new_np.count("buffer_get(")
    # appender scopes: 1 operator handler + 6 static methods + 36 unary methods, each closed right behind the lookups
    # ... + the five full reductions (sum, min, max, prod, mean), consumers that know chains
    assert new_np.count("NPH_LAZY_MARSHAL_BEGIN();") == 43 + 5 == new_np.count("NPH_LAZY_MARSHAL_END();")
    for m in re.finditer(r"NPH_LAZY_MARSHAL_BEGIN\(\);\n(.*?)NPH_LAZY_MARSHAL_END\(\);", new_np, flags=re.S):
        lines = [ln.strip() for ln in m.group(1).strip().split("\n")]
        assert 1 <= len(lines) <= 2 and all(re.fullmatch(r"NDArray \*nd[ab] = ZVAL_TO_NDARRAY\((op1|op2|a|b|array)\);", ln) for ln in lines), lines
    # every marshalling call of the reference is still there (the appenders' inside a scope + kept on the guard's #else side)
    assert new_np.count("ZVAL_TO_NDARRAY(") == ref_np.count("ZVAL_TO_NDARRAY(") + 2 + 12 + 36 + 5
    assert new_np.count("NPH_ReduceAll(") == 6          # sum, min, max, prod + mean's two branches
    for fn, op in (("NDArray_Sum_Float", "NP_SUM"), ("NDArray_Min", "NP_MIN"), ("NDArray_Max", "NP_MAX"), ("NDArray_Float_Prod", "NP_PROD")):
        assert "= NPH_ReduceAll(%s, %s, nda);" % (op, fn) in new_np
    # the axis forms reach reduce() / single_reduce() with an operand that may still be pending: both compute it first
    nd_c = (out / "src/ndarray.c").read_text()
    assert nd_c.count("if (NPH_Flush(array) != 0) {") == 2
    assert new_np.count("rtn = NPH_LazyBinary(") == 12 and len(re.findall(r"rtn = NPH_LazyElementWise(?:1F|2F)?\(nda, cuda_float_\w+", new_np)) == 36
    assert "rtn = NPH_LazyElementWise2F(nda, cuda_float_clip, (float)min, (float)max);" in new_np
    assert "rtn = NPH_LazyElementWise1F(nda, cuda_float_round, (float)precision);" in new_np
    assert "rtn = NPH_LazyElementWise(nda, cuda_float_rsqrt);" in new_np and "rtn = NPH_LazyElementWise(nda, cuda_float_exp2);" in new_np
    # square ($a * $a inside the method), arctan2 and the comparisons are not appenders: they marshal with the flush
    assert "rtn = NDArray_Multiply_Float(nda, nda);" in new_np and "NDArrayMathGPU_ElementWise1N(ndx, cuda_float_arctan2, ndy)" in new_np
    nd = (out / "src/ndarray.c").read_text()
    free_body = nd[nd.index("\nNDArray_FREE(NDArray *array) {"):]
    assert free_body.index("return;") < free_body.index("NPH_OnFree(array);") < free_body.index("NDArray_DELREF(array);")
    for f in ("hip_lazy.c", "hip_lazy.h"):
        assert (out / "src/hip" / f).exists()
    assert "src/hip/hip_lazy.c" in (out / "config.m4").read_text()


def test_the_pending_chain_text_runs_as_a_program_and_cpu_operands_reach_the_reference(tmp_path):
    """The text section 2c inserts, verbatim, inside functions with the shape of ndarray_do_operation_ex / PHP_METHOD(add ...) /
    the unary PHP_METHODs around a restated Zend object table (the tool generates the program; the GPU tier runs its `gpu`
    mode): with CPU operands — BASELINE config 1 — every appender hands the call to the stand-in for the reference's own
    code, nothing becomes pending, no device is touched.  Needs no reference checkout and no GPU."""
    import apply_with_hip as tool
    from numpower_amd import build
    build.build_all()
    text = tool.lazy_program_source()
    flat = text.replace("\n        ", "\n").replace("\n    ", "\n")
    for prefix in ("buffer.c:80-81", "numpower.c:194-195", "numpower.c:3374-3540", "numpower.c:1608-3357"):
        e = next(x for x in tool.EDITS if x.what.startswith(prefix))
        assert e.new in flat, prefix
    for e in tool.EDITS:
        if "`rtn = NDArray_" in e.what:
            assert e.new in flat, e.what
    assert "rtn = NPH_LazyElementWise(nda, cuda_float_exp);" in text and "rtn = NPH_LazyElementWise2F(nda, cuda_float_clip, (float)min, (float)max);" in text
    # the NDArray_FREE hook of the host library (which the program links) is the statement the tool inserts into ndarray.c
    hook = next(x for x in tool.EDITS if x.what.startswith("ndarray.c:588-591")).new.split("\n")[-1]
    assert hook == "NPH_OnFree(array);" and hook in (ROOT / "numpower_amd" / "host" / "numpower_host.cpp").read_text()
    src = tmp_path / "lazy_bodies.c"
    src.write_text(text)
    exe = tmp_path / "lazy_bodies"
    lib = ROOT / "numpower_amd" / "lib"
    proc = subprocess.run(["gcc", "-std=c99", "-O2", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), "-I", str(ROOT / "ext"),
                           str(src), "-o", str(exe), "-L", str(lib), "-lnumpower_host", "-lnp_hip", "-Wl,-rpath," + str(lib)],
                          capture_output=True, text=True)
    assert proc.returncode == 0, proc.stderr
    proc = subprocess.run([str(exe), "cpu"], capture_output=True, text=True, timeout=120)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    m = re.search(r"(\d+) expressions, (\d+) reached the reference's own code, 0 pending", proc.stdout)
    assert m and int(m.group(1)) >= 14 and int(m.group(2)) >= int(m.group(1)) + 5          # + the five reductions of a CPU array


@needs_reference
def test_hip_lazy_uses_only_what_the_reference_headers_declare():
    """ext/hip_lazy.c is compiled inside the patched tree against the reference's OWN headers: every NDArray_* / NDARRAY_* name
    and every struct field it touches must exist there with the meaning the code assumes."""
    src = re.sub(r"/\*.*?\*/", "", (ROOT / "ext" / "hip_lazy.c").read_text(), flags=re.S)
    used = sorted(set(re.findall(r"\b(NDArray_\w+|NDARRAY_\w+)\b", src)))
    headers = (REF / "src" / "ndarray.h").read_text() + (REF / "src" / "initializers.h").read_text()
    assert {"NDArray_EmptyLike", "NDArray_FREE", "NDArray_ADDREF", "NDArray_NDIM", "NDArray_DEVICE", "NDArray_FDATA"} <= set(used)
    for name in used:
        assert re.search(r"#define\s+%s\b|\b%s\s*\(" % (name, name), headers), "%s is not declared by the reference's headers" % name
    # the two struct fields it reads directly (the root of a view chain, the last-reference test of NPH_OnFree)
    assert "a->base" in src and "a->refcount" in src
    nd_h = (REF / "src" / "ndarray.h").read_text()
    assert re.search(r"struct NDArray\* base;", nd_h) and re.search(r"int refcount;", nd_h)
    assert re.search(r"#define NDArray_ADDREF\(a\) \(\(a\)->refcount\+\+\)", headers)
    drivers = (REF / "src" / "ndmath" / "cuda" / "cuda_math.h").read_text()
    for fn in ("NDArrayMathGPU_ElementWise", "NDArrayMathGPU_ElementWise1F", "NDArrayMathGPU_ElementWise2F"):
        assert re.search(r"NDArray\s*\*\s*%s\(" % fn, drivers), fn
    ours = (ROOT / "include" / "numpower_host.h").read_text()
    for name in used:
        assert re.search(r"#define\s+%s\b|\b%s\s*\(" % (name, name), ours), "%s missing from include/numpower_host.h" % name


def test_byte_counts_are_size_t_in_the_emitted_tree(patched, tmp_path):
    """The reference's vmalloc / vmemcpy* take `unsigned int size` (gpu_alloc.h:8,10-11) and every call site passes
    `numElements * sizeof(float)`: a result of 4 GiB or more is truncated, not refused.  The emitted tree declares the three with
    size_t, its m4 block hands NP_GPU_ALLOC_WIDE to the glue, and the glue compiles -Werror in both forms — with the emitted header in
    front of it (so a mismatch between prototype and definition is a compile error) and stand-alone with the reference's types."""
    tool, out, _ = patched
    header = (out / "src" / "gpu_alloc.h").read_text()
    assert "unsigned int size" not in header
    assert header.count("size_t size") == 3 and "#include <stddef.h>" in header
    assert "-DNP_GPU_ALLOC_WIDE=1" in (out / "config.m4").read_text()
    glue = ROOT / "ext" / "gpu_alloc_hip.c"
    inc = ["-I", str(ROOT / "include"), "-I", str(ROOT / "ext")]
    wide = tmp_path / "wide.c"
    wide.write_text('#include "%s"\n#include "%s"\n' % (out / "src" / "gpu_alloc.h", glue))
    for cmd in (["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-DNP_GPU_ALLOC_WIDE=1", *inc, "-c", str(wide), "-o", str(tmp_path / "w.o")],
                ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", *inc, "-c", str(glue), "-o", str(tmp_path / "n.o")]):
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    # ... and the reference's own header in front of the stand-alone form: the signatures objects built against it expect
    narrow = tmp_path / "narrow.c"
    narrow.write_text('#include "%s"\n#include "%s"\n' % (REF / "src" / "gpu_alloc.h", glue))
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", *inc, "-c", str(narrow), "-o", str(tmp_path / "r.o")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # the wrong pairing must NOT compile (the check above is not vacuous)
    r = subprocess.run(["gcc", "-std=c99", "-Werror", *inc, "-c", str(wide), "-o", str(tmp_path / "x.o")], capture_output=True, text=True)
    assert r.returncode != 0 and "conflicting types" in r.stderr
