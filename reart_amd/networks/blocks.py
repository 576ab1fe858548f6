"""Minimal counterpart of the reference's ``networks/blocks.py:99-118`` (MLPConv1d): only the
configuration BaseModel uses (no norm, ReLU, bias-free last layer).  Module / parameter names
match the reference so its checkpoints load (``seg_head.model.0.weight`` ...)."""
import torch.nn as nn


class MLPConv1d(nn.Module):
    def __init__(self, in_channel, mlp, bn=False, gn=False, activation="relu", last_activation="none"):
        super().__init__()
        if bn or gn or activation != "relu" or last_activation != "none":
            raise NotImplementedError("only the BaseModel seg-head configuration is built")
        layers, last = [], in_channel
        for i, out in enumerate(mlp):
            is_last = i == len(mlp) - 1
            layers.append(nn.Conv1d(last, out, kernel_size=1, bias=not is_last))
            if not is_last:
                layers.append(nn.ReLU(inplace=True))
            last = out
        self.model = nn.Sequential(*layers)
        self.out_channel = last

    def forward(self, input):  # [B, C, n] -> [B, out, n]   (plain PyTorch; the fused path
        return self.model(input)  # in BaseModel.forward reads the weights directly)
