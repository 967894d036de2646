"""Host-side feature extractor (PyTorch-ROCm / MIOpen; NOT part of the HIP path).

north_star keeps "the ResNet/FPN backbone ... as PyTorch-ROCm host code".  The
reference gets its trunk from torchvision (reference
``src/models/backbone.py:137-154``) which is not installed here, so this file
carries a plain-torch ResNet bottleneck trunk whose parameter/buffer names are
the ones a reference checkpoint holds (``backbone.encoder.*`` for the whole
torchvision-shaped net including the unused ``layer4``/``fc``, and the
``backbone.layer0..3`` aliases that re-register the same tensors), the
multi-kernel ``PatchMerging`` neck (reference ``backbone.py:18-67``) and the
sine position table with the reference's exponent quirk (reference
``src/models/utils.py:174-205``).
"""
import math

import torch
import torch.nn as nn


class _Bottleneck(nn.Module):
    """1x1 -> 3x3(stride) -> 1x1 residual block, stride on the 3x3 conv."""
    expansion = 4

    def __init__(self, cin, width, stride, project):
        super().__init__()
        cout = width * self.expansion
        self.conv1 = nn.Conv2d(cin, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, stride=stride, padding=1,
                               bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, cout, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, cout, 1, stride=stride, bias=False),
                nn.BatchNorm2d(cout))

    def forward(self, x):
        skip = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + skip)


class _ResNetTrunk(nn.Module):
    """conv1/bn1/maxpool/layer1-4/fc with torchvision's attribute names."""

    _DEPTHS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}

    def __init__(self, num_layers=50):
        super().__init__()
        if num_layers not in self._DEPTHS:
            raise ValueError(
                f'ResNet-{num_layers} trunk not available (bottleneck '
                f'depths {sorted(self._DEPTHS)} only)')
        depths = self._DEPTHS[num_layers]
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        for i, (width, n) in enumerate(zip((64, 128, 256, 512), depths)):
            blocks = []
            for j in range(n):
                stride = 2 if (j == 0 and i > 0) else 1
                blocks.append(_Bottleneck(cin, width, stride, project=(j == 0)))
                cin = width * _Bottleneck.expansion
            setattr(self, f'layer{i + 1}', nn.Sequential(*blocks))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(cin, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out',
                                        nonlinearity='relu')


class ResnetEncoder(nn.Module):
    """Image [N,H,W,3] in [0,1] -> stride-16 feature map [N,1024,H/16,W/16].

    Mirrors reference ``src/models/backbone.py:130-174``: NHWC->NCHW,
    ``(x-0.45)/0.225`` when ``cfg.NORM_INPUT``, conv1..layer3 (or layer4).
    ImageNet weights are not downloadable here, so the trunk is random-init
    until ``load_state_dict``.
    """

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.last_layer = cfg.BACKBONE.LAST_LAYER
        trunk = _ResNetTrunk(cfg.BACKBONE.NUM_LAYERS)
        self.encoder = trunk
        self.layer0 = nn.Sequential(trunk.conv1, trunk.bn1, trunk.relu)
        self.layer1 = nn.Sequential(trunk.maxpool, trunk.layer1)
        self.layer2 = trunk.layer2
        self.layer3 = trunk.layer3
        if cfg.BACKBONE.LAYER == 'layer4':
            self.layer4 = trunk.layer4

    def forward(self, image_nhwc):
        x = image_nhwc.permute(0, 3, 1, 2).contiguous()
        if self.cfg.NORM_INPUT:
            x = (x - 0.45) / 0.225
        x = self.layer2(self.layer1(self.layer0(x)))
        layer = self.cfg.BACKBONE.LAYER       # reference backbone.py:168-172: layer3 only runs
        if layer in ('layer3', 'layer4'):     # for 'layer3' / 'layer4'
            x = self.layer3(x)
        if layer == 'layer4':
            x = self.layer4(x)
        return x


class PatchMerging(nn.Module):
    """LayerNorm over channels, then parallel stride-2 convs with kernel sizes
    ``patch_size`` whose outputs are concatenated (2*dim channels in total).
    Reference ``src/models/backbone.py:18-67``."""

    def __init__(self, input_resolution, dim, norm_layer=nn.LayerNorm,
                 patch_size=(2,)):
        super().__init__()
        self.input_resolution = input_resolution
        self.dim = dim
        self.patch_size = list(patch_size)
        self.reductions = nn.ModuleList()   # registered before `norm`, as in
        self.norm = norm_layer(dim)         # the reference's state-dict order
        last = len(self.patch_size) - 1
        for i, ps in enumerate(self.patch_size):
            out_dim = (2 * dim) // 2 ** (i if i == last else i + 1)
            self.reductions.append(
                nn.Conv2d(dim, out_dim, kernel_size=ps, stride=2,
                          padding=(ps - 2) // 2))

    def forward(self, x):
        # LayerNorm wants channels last; convs want them first.
        x = self.norm(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2).contiguous()
        return torch.cat([conv(x) for conv in self.reductions], dim=1)


def sine_position_table(d_model, max_shape):
    """[1, d_model, H, W] table with the VALUES the reference produces.

    Reference ``src/models/utils.py:185-195``: positions are 1-based cumsums;
    the frequency exponent is written ``-math.log(10000.0) / d_model // 2``
    which Python parses as ``floor((-ln 1e4 / d_model) / 2)`` = -1.0 for
    d_model=256, so ``div_term = exp(-k)`` for k = 0, 2, 4, ... (SURVEY.md
    §8a a1).  Channels interleave sin(x), cos(x), sin(y), cos(y) period 4.
    """
    h, w = max_shape
    ys = torch.arange(1, h + 1, dtype=torch.float32).view(1, h, 1).expand(1, h, w)
    xs = torch.arange(1, w + 1, dtype=torch.float32).view(1, 1, w).expand(1, h, w)
    slope = (-math.log(10000.0) / d_model) // 2
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * slope)
    div = div.view(-1, 1, 1)
    pe = torch.zeros(d_model, h, w)
    pe[0::4] = torch.sin(xs * div)
    pe[1::4] = torch.cos(xs * div)
    pe[2::4] = torch.sin(ys * div)
    pe[3::4] = torch.cos(ys * div)
    return pe.unsqueeze(0)


class PositionEncodingSine(nn.Module):
    """Returns the top-left ``[1,C,h,w]`` window of the precomputed table
    (non-persistent buffer ``pe``, as in the reference, so it is absent from
    the state dict)."""

    def __init__(self, d_model, max_shape=(256, 256)):
        super().__init__()
        self.register_buffer('pe', sine_position_table(d_model, max_shape),
                             persistent=False)

    def forward(self, x):
        return self.pe[:, :, :x.size(2), :x.size(3)]
