#!/usr/bin/env python
"""L2O-DM meta-training (the reference's DM/train_dm.py flags and schedule) on open_l2o_amd:
    python scripts/train_dm.py --problem=quadratic --num_epochs=200 --save_path=out [--if_cl] [--if_scale]
"""
from _train_common import main

if __name__ == "__main__":
    main(rnnprop=False)
