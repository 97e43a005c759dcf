"""AddressSanitizer + UndefinedBehaviorSanitizer over the HOST code of the repository, on the CPU build (GPU ASan is not
available on the pool): numpower_host.cpp, the ext/ glue (gpu_alloc_hip.c, hip_math.c, hip_math_drivers.c, hip_fast.c and
hip_lazy.c — the pending chains: reference counts, the side table, flushes on write, lifetimes) and the three C programs the GPU
tier runs (method_bodies, and fast_path_bodies / lazy_bodies = the text tools/apply_with_hip.py inserts, verbatim), compiled with
-fsanitize=address,undefined and linked with tests/null_device/null_device.c — the entry points of np_hip.h over malloc, a
"device" that computes nothing but touches every byte a kernel would read or write, so that every extent the host code passes
down is checked.  The programs run their GPU mode end to end; what is asserted is: exit status 0 (their own checks of launch
counts, pending counts, error texts and leaks of device allocations hold on the null device too), and not a word from a
sanitizer — no overflow, no use after free, no host-side leak (LeakSanitizer), no undefined behaviour.  Values are not looked at:
there are none."""
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
INC = ["-I%s" % (ROOT / "include"), "-I%s" % (ROOT / "ext"), "-I%s" % (ROOT / "numpower_amd" / "host")]
CFLAGS = ["-std=c99", "-Wall", "-Wextra", "-Werror"]


def _run(cmd, **kw):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, **kw)
    assert proc.returncode == 0, "%s\n%s\n%s" % (" ".join(map(str, cmd)), proc.stdout[-3000:], proc.stderr[-3000:])
    return proc


@pytest.fixture(scope="module")
def programs(tmp_path_factory):
    gcc, gxx = shutil.which("gcc"), shutil.which("g++")
    if not gcc or not gxx:
        pytest.skip("no gcc / g++")
    from numpower_amd import build
    build.build_fast_path_bodies()        # (re)generates build/gen/*.c from the tool when it is newer
    build.build_lazy_bodies()
    gen = ROOT / "build" / "gen"
    d = tmp_path_factory.mktemp("san")
    probe = d / "probe.c"
    probe.write_text("int main(void) { return 0; }\n")
    if subprocess.run([gcc, *SAN, str(probe), "-o", str(d / "probe")], capture_output=True).returncode != 0:
        pytest.skip("this gcc has no libasan / libubsan")
    units = [([gxx, *SAN, "-std=c++17", *INC, "-c", str(ROOT / "numpower_amd" / "host" / "numpower_host.cpp")], "host.o")]
    for name in ("hip_math", "gpu_alloc_hip", "hip_math_drivers", "hip_fast", "hip_lazy", "method_bodies"):
        units.append(([gcc, *SAN, *CFLAGS, *INC, "-c", str(ROOT / "ext" / (name + ".c"))], name + ".o"))
    units.append(([gcc, *SAN, *CFLAGS, *INC, "-c", str(ROOT / "tests" / "null_device" / "null_device.c")], "null_device.o"))
    for name in ("fast_path_bodies", "lazy_bodies"):
        units.append(([gcc, *SAN, *CFLAGS, *INC, "-c", str(gen / (name + ".c"))], name + ".o"))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda u: _run(u[0] + ["-o", str(d / u[1])]), units))
    common = [str(d / n) for n in ("host.o", "hip_math.o", "gpu_alloc_hip.o", "hip_math_drivers.o", "hip_fast.o", "hip_lazy.o", "null_device.o")]
    for name in ("method_bodies", "fast_path_bodies", "lazy_bodies"):
        _run([gxx, *SAN, *common, str(d / (name + ".o")), "-o", str(d / name), "-lm"])
    return d


@pytest.mark.parametrize("name,args", [("method_bodies", []), ("fast_path_bodies", ["gpu"]), ("lazy_bodies", ["gpu"])])
def test_host_code_is_clean_under_asan_and_ubsan(programs, name, args):
    out = programs / (name + ".bin")
    env = {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0:halt_on_error=1", "UBSAN_OPTIONS": "print_stacktrace=1", "PATH": "/usr/bin:/bin"}
    proc = subprocess.run([str(programs / name), *args, str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    report = proc.stderr
    assert "Sanitizer" not in report and "runtime error" not in report, report[-4000:]
    assert proc.returncode == 0, (proc.stdout[-1500:], report[-1500:])
    if name == "lazy_bodies":
        assert "lazy_bodies gpu: ok" in proc.stdout
        assert "exp_mul_add   3 steps: 1 launch(es) with chains (0 before the value was asked for), 3 without" in proc.stdout


def test_the_null_device_is_test_infrastructure():
    """Nothing of the product names or links it, and its header says what it is."""
    text = (ROOT / "tests" / "null_device" / "null_device.c").read_text()
    assert "TEST INFRASTRUCTURE" in text and "COMPUTES\n * NOTHING" in text and "never part of the product" in text
    for path in list((ROOT / "numpower_amd").rglob("*")) + list((ROOT / "ext").rglob("*")) + [ROOT / "bench.py", ROOT / "__graft_entry__.py"]:
        if path.suffix in (".py", ".hip", ".cpp", ".h", ".c"):
            assert "null_device" not in path.read_text(), "%s names the tests' null device" % path


def test_the_harness_sees_a_seeded_lifetime_bug(programs):
    """Sensitivity: ext/hip_lazy.c with the one line removed that makes a pending chain hold its inputs
    (`NDArray_ADDREF(c.inputs[i])`) — `$c = 1 / $t; unset($t);` then reads freed memory, and the run must say so."""
    src = (ROOT / "ext" / "hip_lazy.c").read_text()
    lines = [ln for ln in src.split("\n") if "NDArray_ADDREF(c.inputs[i])" not in ln]
    assert len(lines) == len(src.split("\n")) - 1
    broken = programs / "hip_lazy_broken.c"
    broken.write_text("\n".join(lines))
    gcc, gxx = shutil.which("gcc"), shutil.which("g++")
    _run([gcc, *SAN, "-std=c99", *INC, "-c", str(broken), "-o", str(programs / "hip_lazy_broken.o")])
    objs = [str(programs / n) for n in ("host.o", "hip_math.o", "gpu_alloc_hip.o", "hip_math_drivers.o", "hip_fast.o", "hip_lazy_broken.o",
                                        "null_device.o", "lazy_bodies.o")]
    _run([gxx, *SAN, *objs, "-o", str(programs / "lazy_broken"), "-lm"])
    proc = subprocess.run([str(programs / "lazy_broken"), "gpu", str(programs / "broken.bin")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          text=True, timeout=600)
    assert proc.returncode != 0 and "heap-use-after-free" in proc.stderr, proc.stderr[-2000:]
