"""Builds a TorchScript archive whose state_dict has given keys (stand-in for OpenAI's released
.pt format, which is a scripted module; the real file is not available offline)."""
import torch
from torch import nn


class Node(nn.Module):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x


def save_scripted_state_dict(sd, path, half=True):
    root = Node()
    for k, v in sd.items():
        parts = k.split(".")
        m = root
        for p in parts[:-1]:
            if not hasattr(m, p):
                m.add_module(p, Node())
            m = getattr(m, p)
        v = v.clone()
        if half and torch.is_floating_point(v):
            v = v.half()                      # OpenAI ships fp16 weights
        m.register_parameter(parts[-1], nn.Parameter(v, requires_grad=False))
    torch.jit.script(root).save(path)
