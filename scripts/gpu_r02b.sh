#!/bin/bash
O=gpurun_out/r02d; mkdir -p $O
(timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -8) | tee $O/pytest.log
bash scripts/variants.sh $O
