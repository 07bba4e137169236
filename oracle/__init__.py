"""CPU oracle for the Open-L2O model-free inner unroll loop.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / the timed CPU baseline.
The product path (``open_l2o_amd``) never imports this package and fails loudly
when the HIP extension is missing.

Parity status: the optimizee closed forms, preprocess and unroll semantics are
pinned by the reference's own known-answer tests (see tests/test_oracle_kat.py).
The LSTM cell arithmetic lives in dm-sonnet==1.11 (pinned in the reference's
requirements.txt, NOT vendored under /root/reference and not installable here);
it is restated from Sonnet's published algorithm and cross-checked against
torch.nn.LSTMCell on CPU.  No reference test pins the cell with a non-zero
width, so for the LSTM cell: **parity unpinned** (see DESIGN.md).
"""
from .l2o_oracle import *  # noqa: F401,F403
