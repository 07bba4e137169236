#!/bin/bash
# lease: bias-as-accumulator-init build: drift per form, the whole GPU suite, bench c2 / c3 / c4 / c5
TAG=${1:-r03g}
O=gpurun_out/$TAG; mkdir -p $O
cd "$(dirname "$0")/.."
timeout 600 python scripts/drift_forms.py > $O/drift_forms.txt 2>&1; grep -v amdgpu.ids $O/drift_forms.txt | cut -c1-150
(timeout 2400 python -m pytest tests -q -m gpu --durations=8 2>&1 | tail -80) > $O/pytest.log; tail -15 $O/pytest.log
(timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1) | tee $O/smoke.log
for c in 2 3 4 5; do timeout 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline 2>>$O/bench.err | tee $O/bench_c$c.json | cut -c1-330; done
