"""tlx_Identity / tlx_Dropout as the reference uses them (swin_transformer.py:8,79,166; mobilenetv2.py:98;
mobilenetv3.py:167): identity in eval mode."""
from oracle.tlx_cpu import nn


class tlx_Identity(nn.Module):
    def __init__(self, *args, **kwargs):
        super().__init__()

    def forward(self, x):
        return x


class tlx_Dropout(nn.Module):
    def __init__(self, p=0.5, *args, **kwargs):
        super().__init__()
        self.p = p

    def forward(self, x):
        assert not self.is_train or not self.p, "oracle stand-in is eval-only"
        return x
