"""Learning 2 Learn preprocessing modules -- the reference's ``DM/preprocess.py``
(``Clamp`` :26-39, ``LogAndSign`` :42-70).

Inside the optimizer networks the preprocessing is fused into the HIP kernels
(``preprocess_grad`` in csrc/l2o_common.h).  These classes exist for API parity /
standalone use; their ``__call__`` is plain elementwise tensor code (not on the hot path).
"""
import numpy as np
import torch


class Clamp(object):
    def __init__(self, min_value=None, max_value=None, name="clamp"):
        self.name = name
        self._min = min_value
        self._max = max_value

    def __call__(self, inputs):
        output = inputs
        if self._min is not None:
            output = torch.clamp(output, min=self._min)
        if self._max is not None:
            output = torch.clamp(output, max=self._max)
        return output


class LogAndSign(object):
    """Log and sign preprocessing (https://arxiv.org/pdf/1606.04474v1.pdf, Appendix A)."""

    def __init__(self, initializer=None, k=5, name="preprocess_log"):
        self.name = name
        self._k = k

    def __call__(self, gradients):
        eps = float(np.finfo(np.float32).eps)
        ndims = gradients.dim()
        log = torch.log(torch.abs(gradients) + eps)
        clamped_log = Clamp(min_value=-1.0)(log / self._k)
        sign = Clamp(min_value=-1.0, max_value=1.0)(gradients * float(np.float32(np.exp(self._k))))
        return torch.cat([clamped_log, sign], ndims - 1)
