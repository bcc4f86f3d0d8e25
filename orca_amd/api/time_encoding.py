"""model/time_encoding.py:6-28 -- TimeEncode."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


class TimeEncode(nn.Module):
  def __init__(self, dimension):
    super().__init__()
    if dimension != ops.D:
      raise ValueError(f"TimeEncode: the HIP path is built for dimension {ops.D}, got {dimension}")
    self.dimension = dimension
    self.w = nn.Linear(1, dimension)
    self.reset_parameters()

  def reset_parameters(self):
    self.w.weight = nn.Parameter(torch.from_numpy(1 / 10 ** np.linspace(0, 9, self.dimension, dtype=np.float32))
                                 .reshape(self.dimension, -1), requires_grad=False)
    self.w.bias = nn.Parameter(torch.zeros(self.dimension), requires_grad=False)

  @torch.no_grad()
  def forward(self, t):
    """t [B, K] float32 on the device -> cos(t*w + b) [B, K, dimension]."""
    return ops.time_encode(t.float().contiguous(), self.w.weight.reshape(-1).contiguous(), self.w.bias.contiguous())
