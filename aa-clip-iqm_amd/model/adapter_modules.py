"""Adapter parameter containers, reference model/adapter_modules.py:6-26.
`fc` keeps the reference's key names ('fc.0.weight' with the LeakyReLU wrapper,
'fc.weight' without)."""
from torch import nn

from aaclip_hip import engine

from .transformer import Linear


class _LinearLeaky(nn.Sequential):
    """Sequential(Linear(no bias), LeakyReLU) whose forward is one HIP GEMM with
    the LeakyReLU fused in the epilogue."""

    def __init__(self, c_in, c_out):
        super().__init__(Linear(c_in, c_out, bias=False), nn.LeakyReLU())

    def forward(self, x):
        code = engine.dtype_code(getattr(self[0], "precision", "fp32"))
        return engine.linear(x, self[0].weight, None, True, code).to(x.dtype)


class SimpleAdapter(nn.Module):
    def __init__(self, c_in, c_out=768):
        super().__init__()
        self.fc = _LinearLeaky(c_in, c_out)

    @property
    def weight(self):
        return self.fc[0].weight

    def forward(self, x):
        return self.fc(x)


class SimpleProj(nn.Module):
    def __init__(self, c_in, c_out=768, relu=True):
        super().__init__()
        self.relu = relu
        self.fc = _LinearLeaky(c_in, c_out) if relu else Linear(c_in, c_out, bias=False)

    @property
    def weight(self):
        return self.fc[0].weight if self.relu else self.fc.weight

    def forward(self, x):
        return self.fc(x)
