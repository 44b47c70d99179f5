"""Restatement of the ONE class of loralib==0.1.1 the reference uses (requirements.txt:11,
models/ynet.py:4,143-144): ``loralib.Conv2d``.

TEST INFRASTRUCTURE ONLY (used by oracle/gen_goldens.py so the reference imports here).
PARITY UNPINNED: loralib is a third-party dependency that is neither vendored under
/root/reference nor installed in this image, and no reference test pins its results.  This file
restates its published 0.1.1 algorithm:

  * parameters live directly on the conv: weight, bias, lora_A [r*k, Cin*k], lora_B [Cout*k, r*k]
  * scaling = lora_alpha / r with lora_alpha = 1 (the reference never passes lora_alpha)
  * forward: conv2d(x, weight + (lora_B @ lora_A).view(weight.shape) * scaling, bias)
  * init: conv reset, then lora_A ~ kaiming_uniform(a=sqrt(5)), lora_B = 0; base weight frozen
  * .eval() merges into weight / .train() un-merges (never reached through model.eval(), which
    calls children's train(False))
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class Conv2d(nn.Conv2d):
    def __init__(self, in_channels, out_channels, kernel_size, r=0, lora_alpha=1,
                 lora_dropout=0.0, merge_weights=True, **kwargs):
        self._lora_ready = False
        nn.Conv2d.__init__(self, in_channels, out_channels, kernel_size, **kwargs)
        assert type(kernel_size) is int
        self.r = r
        self.lora_alpha = lora_alpha
        self.merged = False
        self.merge_weights = merge_weights
        if r > 0:
            self.lora_A = nn.Parameter(self.weight.new_zeros((r * kernel_size, in_channels * kernel_size)))
            self.lora_B = nn.Parameter(self.weight.new_zeros((out_channels * kernel_size, r * kernel_size)))
            self.scaling = self.lora_alpha / self.r
            self.weight.requires_grad = False
        self._lora_ready = True
        self.reset_parameters()

    def reset_parameters(self):
        nn.Conv2d.reset_parameters(self)
        if getattr(self, "_lora_ready", False) and hasattr(self, "lora_A"):
            nn.init.kaiming_uniform_(self.lora_A, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B)

    def _delta(self):
        return (self.lora_B @ self.lora_A).view(self.weight.shape) * self.scaling

    def train(self, mode=True):
        nn.Conv2d.train(self, mode)
        if self.merge_weights and self.merged and self.r > 0:
            self.weight.data -= self._delta()
            self.merged = False
        return self

    def eval(self):
        nn.Conv2d.eval(self)
        if self.merge_weights and not self.merged and self.r > 0:
            self.weight.data += self._delta()
            self.merged = True
        return self

    def forward(self, x):
        if self.r > 0 and not self.merged:
            return F.conv2d(x, self.weight + self._delta(), self.bias, self.stride,
                            self.padding, self.dilation, self.groups)
        return nn.Conv2d.forward(self, x)
