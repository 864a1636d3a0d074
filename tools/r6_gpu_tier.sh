#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -m gpu -q --durations=5 2>&1 | tail -14 > gpurun_out/r6g_gpu_pytest.log; cat gpurun_out/r6g_gpu_pytest.log
python -c "import __graft_entry__ as g; g.smoke()"
