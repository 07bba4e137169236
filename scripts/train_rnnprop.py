#!/usr/bin/env python
"""L2O-RNNProp meta-training (the reference's DM/train_rnnprop.py flags and schedule) on open_l2o_amd:
    python scripts/train_rnnprop.py --problem=quadratic --num_epochs=200 --save_path=out [--beta1 .95 --beta2 .95]
"""
from _train_common import main

if __name__ == "__main__":
    main(rnnprop=True)
