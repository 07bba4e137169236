#!/bin/bash
# the whole GPU suite (no -x: every failure in one lease) + smoke
TAG=${1:-r03f}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
(timeout 2400 python -m pytest tests -q -m gpu --durations=15 2>&1 | tail -120) > $O/pytest.log; tail -40 $O/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.log
