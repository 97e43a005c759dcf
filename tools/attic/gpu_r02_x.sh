#!/bin/bash
# full GPU suite + smoke, then the evidence run
mkdir -p gpurun_out/r02x
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02x/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02x/pytest.log
tail -3 gpurun_out/r02x/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02x/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r02x/smoke.log
bash tools/gpu_r02_final.sh
