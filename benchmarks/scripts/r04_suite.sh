#!/bin/bash
# full GPU suite + smoke
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04_suite
python -m pytest tests -q -m gpu 2>&1 | grep -v "Warning\|warnings.warn\|pin_memory\|^$" | tail -25 > gpurun_out/r04_suite/pytest.log; cat gpurun_out/r04_suite/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
