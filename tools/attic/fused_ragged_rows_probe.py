"""sum(exp(X), axis 1) as one fused launch with row lengths that are / are not multiples of 4: ms, GB/s over 4 B/elem.
NP_HIP_LIB selects the build for a same-box A/B."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from numpower_amd import _lib, synth
from numpower_amd.lazy import Lazy   # noqa: F401
from numpower_amd.ndarray import NDArray

lib = _lib.load()
_lib.check(lib.np_init(0))
import os
variants = [int(v) for v in os.environ.get("NP_PROBE_VARIANTS", "0").split(",")]
shapes = ((25000, 4000), (25000, 4001), (33333, 3001), (400_000, 250), (400_000, 256), (1_000_000, 100), (1_600_000, 64), (2_000_000, 50), (800_000, 127),
          (800_000, 128), (500_000, 192), (260_000, 384), (200_000, 500), (200_000, 512), (160_000, 640), (130_000, 768), (100_000, 1000), (100_000, 1001), (98_000, 1024), (50_000, 2000), (5000, 20_001), (300, 333_335))
for variant in variants:
  _lib.check(lib.np_elementwise_set_variant(variant))
  print("== np_elementwise_set_variant(%d)" % variant)
  for rows, cols in shapes:
      x = synth.uniform((rows, cols), 3, -1.0, 1.0)
      gx = NDArray.array(x).gpu()
      build = lambda: gx.lazy().exp().sum(axis=1)
      for _ in range(3):
          y = build()
      _lib.check(lib.np_sync())
      t = _lib.Timer(); t.start()
      for _ in range(10):
          y = build()
      t.stop(); _lib.check(lib.np_sync())
      ms = t.elapsed_ms() / 10
      want = np.exp(x.astype(np.float64)).sum(1)
      ok = bool((np.abs(y.cpu().numpy() - want) <= 1e-5 * want).all())
      print("%7d x %-7d  sum(exp(X),1) %.4f ms %5.0f GB/s %s" % (rows, cols, ms, 4.0 * rows * cols / ms / 1e6, "ok" if ok else "WRONG"), flush=True)
