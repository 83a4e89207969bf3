"""ResNet-50 + FPN as the r3det configs instantiate them through mmdet
(configs/r3det/r3det_r50_fpn_1x_dota_v1.py:8-25).  mmdet/torchvision are not part of the
reference tree nor installed here, so the two modules are restated in plain torch.nn with
mmdet's parameter names (``layer1.0.conv1``, ``lateral_convs.0.conv``, ``fpn_convs.0.conv``)
so that reference checkpoints would load."""
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        # style='pytorch': the stride sits on the 3x3 conv
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNet50(nn.Module):
    """depth=50, num_stages=4, out_indices=(0,1,2,3), frozen_stages=1, norm_eval=True.  Registered as ``ResNet``
    with mmdet's keywords (configs/r3det/r3det_r50_fpn_1x_dota_v1.py:8-18): what this restatement does not
    implement (another depth / style, fewer stages) raises instead of being ignored; ``init_cfg`` / ``pretrained``
    are accepted and not acted on (there are no checkpoints here: random init)."""

    def __init__(self, frozen_stages=1, norm_eval=True, depth=50, num_stages=4, out_indices=(0, 1, 2, 3),
                 zero_init_residual=False, norm_cfg=None, style='pytorch', init_cfg=None, pretrained=None):
        super().__init__()
        norm_cfg = dict(norm_cfg or dict(type='BN', requires_grad=True))
        if depth != 50 or num_stages != 4 or tuple(out_indices) != (0, 1, 2, 3) or style != 'pytorch' or \
                norm_cfg.get('type', 'BN') != 'BN' or zero_init_residual:
            raise NotImplementedError('ResNet: only the shipped configuration (depth 50, 4 stages, pytorch style, BN)')
        self.frozen_stages, self.norm_eval = frozen_stages, norm_eval
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        inplanes = 64
        for i, (planes, blocks, stride) in enumerate([(64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)]):
            down = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                 nn.BatchNorm2d(planes * 4))
            layers = [Bottleneck(inplanes, planes, stride, down)]
            inplanes = planes * 4
            layers += [Bottleneck(inplanes, planes) for _ in range(1, blocks)]
            setattr(self, f'layer{i + 1}', nn.Sequential(*layers))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        self._freeze()

    def _freeze(self):
        if self.frozen_stages >= 0:
            for m in (self.conv1, self.bn1):
                for p in m.parameters():
                    p.requires_grad = False
        for i in range(1, self.frozen_stages + 1):
            for p in getattr(self, f'layer{i}').parameters():
                p.requires_grad = False

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        return self

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        outs = []
        for i in range(4):
            x = getattr(self, f'layer{i + 1}')(x)
            outs.append(x)
        return tuple(outs)


class ConvModule(nn.Module):
    """mmcv ConvModule without norm: ``.conv`` (+ optional ReLU)."""

    def __init__(self, cin, cout, k, stride=1, padding=0, act=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=padding)
        self.act = act

    def forward(self, x):
        x = self.conv(x)
        return F.relu(x, inplace=True) if self.act else x


class FPN(nn.Module):
    """in_channels=[256,512,1024,2048], out 256, start_level=1, add_extra_convs='on_input',
    num_outs=5: P3-P5 from C3-C5, P6 = 3x3/s2 conv on C5, P7 = 3x3/s2 conv on P6."""

    def __init__(self, in_channels=(256, 512, 1024, 2048), out_channels=256, start_level=1, num_outs=5,
                 add_extra_convs='on_input', end_level=-1, relu_before_extra_convs=False, no_norm_on_lateral=False,
                 conv_cfg=None, norm_cfg=None, act_cfg=None, upsample_cfg=None, init_cfg=None):
        super().__init__()
        if add_extra_convs != 'on_input' or end_level != -1 or relu_before_extra_convs or conv_cfg or norm_cfg or \
                act_cfg or (upsample_cfg and dict(upsample_cfg).get('mode', 'nearest') != 'nearest'):
            raise NotImplementedError("FPN: only the shipped configuration (add_extra_convs='on_input', no norm)")
        self.start_level = start_level
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for c in in_channels[start_level:]:
            self.lateral_convs.append(ConvModule(c, out_channels, 1))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1))
        used = len(in_channels) - start_level
        for i in range(num_outs - used):
            cin = in_channels[-1] if i == 0 else out_channels
            self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)

    def forward(self, inputs):
        lat = [l(inputs[i + self.start_level]) for i, l in enumerate(self.lateral_convs)]
        for i in range(len(lat) - 1, 0, -1):
            lat[i - 1] = lat[i - 1] + F.interpolate(lat[i], size=lat[i - 1].shape[2:], mode='nearest')
        outs = [self.fpn_convs[i](lat[i]) for i in range(len(lat))]
        outs.append(self.fpn_convs[len(lat)](inputs[-1]))  # on_input: from C5
        for i in range(len(lat) + 1, len(self.fpn_convs)):
            outs.append(self.fpn_convs[i](outs[-1]))
        return tuple(outs)
