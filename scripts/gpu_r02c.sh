#!/bin/bash
O=gpurun_out/r02c; mkdir -p $O
bash scripts/variants.sh $O
L2O_HIP_LIB=$PWD/build/lib_phases.so python scripts/phase_profile.py 2>&1 | tee $O/phases.txt
