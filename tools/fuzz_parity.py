"""The random-shape parity sweep of tests/test_gpu_fuzz.py, long: NP_FUZZ_CASES cases per family
(default 400), generator seeds offset by [seed].  Usage: python tools/fuzz_parity.py [cases] [seed]"""
import os
import subprocess
import sys
from pathlib import Path
root = Path(__file__).resolve().parent.parent
env = dict(os.environ, NP_FUZZ_CASES=sys.argv[1] if len(sys.argv) > 1 else "400",
           NP_FUZZ_SEED=sys.argv[2] if len(sys.argv) > 2 else "0")
sys.exit(subprocess.call([sys.executable, "-m", "pytest", str(root / "tests" / "test_gpu_fuzz.py"), "-x", "-q", "-m", "gpu"], env=env, cwd=root))
