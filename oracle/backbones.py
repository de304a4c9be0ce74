"""CPU oracle: fp32 PyTorch restatement of the reference backbones.

TEST INFRASTRUCTURE ONLY.  Nothing under ``face-recognition_amd/`` may import this module; it is
the checker used by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg.  It is pinned against the real reference (imported unmodified in the build container by
``tests/golden/make_golden.py``): identical ``state_dict`` keys/shapes, identical weights for a
given ``torch.manual_seed`` (sha256 fixtures) and identical embeddings (golden vectors).

Every builder cites the reference lines it restates (paths relative to /root/reference).  The
restatement is table-driven: each architecture is a list of rows, ``_cna`` builds the
conv -> norm -> activation triple whose child indices ("0", "1", "2") appear in checkpoint keys.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import nn

MODEL_NAMES = (
    "sphere20", "sphere36", "sphere64",
    "mobilenetv1", "mobilenetv2", "mobilenetv3_small", "mobilenetv3_large",
)


def round_channels(v: float, divisor: int = 8) -> int:
    """utils/layers.py:6-12 (_make_divisible)."""
    r = max(divisor, int(v + divisor / 2) // divisor * divisor)
    return r + divisor if r < 0.9 * v else r


def _cna(cin: int, cout: int, k: int = 3, stride: int = 1, groups: int = 1,
         norm: bool = True, act: Optional[str] = "prelu", bias: bool = True,
         padding: Optional[int] = None) -> nn.Sequential:
    """utils/layers.py:15-84: Conv2d(bias=True by default) -> BatchNorm2d? -> activation?.

    PReLU is per-channel (num_parameters=cout, :78-79); Hardswish is parameter free.
    """
    pad = (k - 1) // 2 if padding is None else padding          # :59-60 (dilation is always 1)
    mods: List[nn.Module] = [nn.Conv2d(cin, cout, k, stride, pad, groups=groups, bias=bias)]
    if norm:
        mods.append(nn.BatchNorm2d(cout))
    if act == "prelu":
        mods.append(nn.PReLU(num_parameters=cout))
    elif act == "hswish":
        mods.append(nn.Hardswish(inplace=True))
    elif act is not None:
        raise ValueError(act)
    return nn.Sequential(*mods)


class SqueezeExcite(nn.Module):
    """utils/layers.py:87-122 with scale_activation=Hardsigmoid (models/mobilenetv3.py:46)."""

    def __init__(self, channels: int, squeeze: int) -> None:
        super().__init__()
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        s = x.mean(dim=(2, 3), keepdim=True)                    # AdaptiveAvgPool2d(1), :114
        s = F.relu(self.fc1(s))                                 # :115-116
        s = F.hardsigmoid(self.fc2(s))                          # :117-118
        return s * x                                            # :122


class _LinearBlock(nn.Module):
    """utils/layers.py:156-173: conv(bias=False) + BN under the attribute name ``layers``."""

    def __init__(self, cin: int, cout: int, k: int, groups: int) -> None:
        super().__init__()
        self.layers = nn.Sequential(
            nn.Conv2d(cin, cout, k, 1, 0, groups=groups, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.layers(x)


class GlobalDepthwiseHead(nn.Module):
    """utils/layers.py:176-191 (GDC): dw7x7 valid + BN2d -> flatten -> Linear(no bias) -> BN1d."""

    def __init__(self, channels: int, embedding_dim: int) -> None:
        super().__init__()
        self.features = nn.Sequential(_LinearBlock(channels, channels, 7, channels), nn.Flatten())
        self.fc = nn.Sequential(nn.Linear(channels, embedding_dim, bias=False),
                                nn.BatchNorm1d(embedding_dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc(self.features(x))


def _mobilenet_init(net: nn.Module) -> None:
    """models/mobilenetv1.py:47-59, mobilenetv2.py:148-159, mobilenetv3.py:163-175 (same rule)."""
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out")
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)
        elif isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0, 0.01)
            if m.bias is not None:
                nn.init.zeros_(m.bias)


# ----------------------------------------------------------------------------- MobileNetV1
class MobileNetV1(nn.Module):
    """models/mobilenetv1.py:10-67.  width_mult defaults to 0.25 (:11)."""

    # (out filter index, stride) of the 13 depthwise-separable blocks, :25-42
    _ROWS = ((1, 1), (2, 2), (2, 1), (3, 2), (3, 1),
             (4, 2), (4, 1), (4, 1), (4, 1), (4, 1), (4, 1),
             (5, 2), (5, 1))

    def __init__(self, embedding_dim: int = 512, width_mult: float = 0.25) -> None:
        super().__init__()
        f = [round_channels(c * width_mult) for c in (32, 64, 128, 256, 512, 1024)]   # :14-15
        blocks: List[nn.Module] = [_cna(3, f[0], 3, 1)]                               # stem, stride 1 (:22)
        cin = f[0]
        for idx, stride in self._ROWS:
            # utils/layers.py:125-153: dw3x3(stride)+BN+PReLU then pw1x1+BN+PReLU
            blocks.append(nn.Sequential(_cna(cin, cin, 3, stride, groups=cin), _cna(cin, f[idx], 1)))
            cin = f[idx]
        self.stage1 = nn.Sequential(*blocks[:6])
        self.stage2 = nn.Sequential(*blocks[6:12])
        self.stage3 = nn.Sequential(*blocks[12:])
        self.output_layer = GlobalDepthwiseHead(f[5], embedding_dim)
        _mobilenet_init(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.output_layer(self.stage3(self.stage2(self.stage1(x))))


# ----------------------------------------------------------------------------- MobileNetV2
class _V2Block(nn.Module):
    """models/mobilenetv2.py:11-57."""

    def __init__(self, cin: int, cout: int, stride: int, t: int) -> None:
        super().__init__()
        hidden = int(round(cin * t))
        self.residual = stride == 1 and cin == cout                                   # :19
        mods: List[nn.Module] = []
        if t != 1:
            mods.append(_cna(cin, hidden, 1))                                         # :24-32
        mods += [_cna(hidden, hidden, 3, stride, groups=hidden),                      # :36-43
                 nn.Conv2d(hidden, cout, 1, 1, 0, bias=False),                        # :45
                 nn.BatchNorm2d(cout)]                                                # :46
        self.conv = nn.Sequential(*mods)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x + self.conv(x) if self.residual else self.conv(x)                    # :54-57


class MobileNetV2(nn.Module):
    """models/mobilenetv2.py:60-165 (width_mult 1.0, round_nearest 8)."""

    _TCNS = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2),
             (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))                          # :91-100

    def __init__(self, embedding_dim: int = 512) -> None:
        super().__init__()
        cin, last = 32, 512                                                           # :87-88
        feats: List[nn.Module] = [_cna(3, cin, 3, 1)]                                 # :112-118
        for t, c, n, s in self._TCNS:
            for i in range(n):
                feats.append(_V2Block(cin, c, s if i == 0 else 1, t))                 # :121-134
                cin = c
        feats.append(_cna(cin, last, 1))                                              # :136-139
        self.features = nn.Sequential(*feats)
        self.output_layer = GlobalDepthwiseHead(last, embedding_dim)                  # :145
        _mobilenet_init(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.output_layer(self.features(x))


# ----------------------------------------------------------------------------- MobileNetV3
# (cin, kernel, expanded, cout, use_se, activation, stride) -- models/mobilenetv3.py:197-230
_V3_ROWS = {
    "large": ((16, 3, 16, 16, False, "RE", 1), (16, 3, 64, 24, False, "RE", 2),
              (24, 3, 72, 24, False, "RE", 1), (24, 5, 72, 40, True, "RE", 2),
              (40, 5, 120, 40, True, "RE", 1), (40, 5, 120, 40, True, "RE", 1),
              (40, 3, 240, 80, False, "HS", 2), (80, 3, 200, 80, False, "HS", 1),
              (80, 3, 184, 80, False, "HS", 1), (80, 3, 184, 80, False, "HS", 1),
              (80, 3, 480, 112, True, "HS", 1), (112, 3, 672, 112, True, "HS", 1),
              (112, 5, 672, 160, True, "HS", 2), (160, 5, 960, 160, True, "HS", 1),
              (160, 5, 960, 160, True, "HS", 1)),
    "small": ((16, 3, 16, 16, True, "RE", 2), (16, 3, 72, 24, False, "RE", 2),
              (24, 3, 88, 24, False, "RE", 1), (24, 5, 96, 40, True, "HS", 2),
              (40, 5, 240, 40, True, "HS", 1), (40, 5, 240, 40, True, "HS", 1),
              (40, 5, 120, 48, True, "HS", 1), (48, 5, 144, 48, True, "HS", 1),
              (48, 5, 288, 96, True, "HS", 2), (96, 5, 576, 96, True, "HS", 1),
              (96, 5, 576, 96, True, "HS", 1)),
}


class _V3Block(nn.Module):
    """models/mobilenetv3.py:41-98.  "RE" rows use per-channel PReLU, not ReLU (:55)."""

    def __init__(self, cin: int, k: int, exp: int, cout: int, se: bool, kind: str, stride: int) -> None:
        super().__init__()
        act = "hswish" if kind == "HS" else "prelu"
        self.residual = stride == 1 and cin == cout                                   # :52
        mods: List[nn.Module] = []
        if exp != cin:
            mods.append(_cna(cin, exp, 1, act=act))                                   # :58-66
        mods.append(_cna(exp, exp, k, stride, groups=exp, act=act))                   # :70-80
        if se:
            mods.append(SqueezeExcite(exp, round_channels(exp // 4, 8)))              # :81-83
        mods.append(_cna(exp, cout, 1, act=None))                                     # :86-88
        self.block = nn.Sequential(*mods)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = self.block(x)
        return y + x if self.residual else y                                          # :94-98


class MobileNetV3(nn.Module):
    """models/mobilenetv3.py:101-181; rows already width-adjusted (width_mult 1.0)."""

    def __init__(self, arch: str, embedding_dim: int = 512) -> None:
        super().__init__()
        rows = _V3_ROWS[arch]
        feats: List[nn.Module] = [_cna(3, rows[0][0], 3, 1, act="hswish")]            # :133-141
        feats += [_V3Block(*r) for r in rows]                                         # :144-145
        last = 6 * rows[-1][3]                                                        # :148-149
        feats.append(_cna(rows[-1][3], last, 1, act="hswish"))                        # :150-157
        self.features = nn.Sequential(*feats)
        self.output_layer = GlobalDepthwiseHead(last, embedding_dim)                  # :160
        _mobilenet_init(self)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.output_layer(self.features(x))


# ----------------------------------------------------------------------------- SphereFace
class _SphereResidual(nn.Module):
    """models/sphereface.py:10-48: two bias-free 3x3 conv + PReLU, no norm; x + block(x)."""

    residual = True

    def __init__(self, c: int) -> None:
        super().__init__()
        self.block = nn.Sequential(_cna(c, c, 3, norm=False, bias=False),
                                   _cna(c, c, 3, norm=False, bias=False))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x + self.block(x)


class SphereNet(nn.Module):
    """models/sphereface.py:51-129 (norm_layer=False: no BatchNorm anywhere)."""

    def __init__(self, blocks: Sequence[int], embedding_dim: int = 512, in_channels: int = 3) -> None:
        super().__init__()
        cin = in_channels
        for i, (c, n) in enumerate(zip((64, 128, 256, 512), blocks), start=1):        # :75-81
            stage = [_cna(cin, c, 3, 2, norm=False)] + [_SphereResidual(c) for _ in range(n)]   # :107-117
            setattr(self, f"layer{i}", nn.Sequential(*stage))
            cin = c
        self.fc = nn.Linear(512 * 7 * 7, embedding_dim)                               # :84
        for m in self.modules():                                                      # :89-96
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                if m.bias is not None:
                    nn.init.xavier_uniform_(m.weight)
                    nn.init.constant_(m.bias, 0.0)
                else:
                    nn.init.normal_(m.weight, 0, 0.01)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(x, 1))                                           # NCHW flatten, :126-127


_SPHERE_BLOCKS = {"sphere20": (1, 2, 4, 1), "sphere36": (2, 4, 8, 2), "sphere64": (3, 8, 16, 3)}  # :144,159,173


def build(name: str, embedding_dim: int = 512) -> nn.Module:
    """Same name -> constructor map as inference.py:39-47 (but builds only the one requested)."""
    if name in _SPHERE_BLOCKS:
        return SphereNet(_SPHERE_BLOCKS[name], embedding_dim, 3)
    if name == "mobilenetv1":
        return MobileNetV1(embedding_dim)
    if name == "mobilenetv2":
        return MobileNetV2(embedding_dim)
    if name == "mobilenetv3_small":
        return MobileNetV3("small", embedding_dim)
    if name == "mobilenetv3_large":
        return MobileNetV3("large", embedding_dim)
    raise ValueError(f"Unsupported network '{name}'! Available options: {list(MODEL_NAMES)}")


def seeded(name: str, seed: int = 0) -> nn.Module:
    """``torch.manual_seed(seed); ctor()`` in eval mode -- the random-init weights of BASELINE configs."""
    torch.manual_seed(seed)
    return build(name).eval()


# ----------------------------------------------------------------------------- taps / utilities
def unit_names(net: nn.Module) -> List[str]:
    """Names of every conv->norm->act unit, SE block, bare project conv+BN pair and head stage, in
    execution order.  These are the points at which the HIP plan materialises an activation."""
    out = []
    for name, m in net.named_modules():
        if isinstance(m, (SqueezeExcite, _LinearBlock)):
            out.append(name)
        elif isinstance(m, nn.Sequential) and len(m) and isinstance(m[0], nn.Conv2d) and not name.endswith(".layers"):
            out.append(name)
        elif isinstance(m, nn.BatchNorm2d) and name.split(".")[-2] == "conv":   # mnv2 project conv's BN
            out.append(name)
    return out


@torch.no_grad()
def forward_with_taps(net: nn.Module, x: torch.Tensor, names: Sequence[str]) -> Tuple[torch.Tensor, Dict[str, torch.Tensor]]:
    """Run ``net`` and capture the output of every module in ``names`` (fp32 NCHW clones)."""
    taps: Dict[str, torch.Tensor] = OrderedDict()
    mods = dict(net.named_modules())
    hooks = [mods[n].register_forward_hook(lambda _m, _i, o, n=n: taps.__setitem__(n, o.detach().clone()))
             for n in names]
    try:
        y = net(x)
    finally:
        for h in hooks:
            h.remove()
    return y, taps


def state_dict_sha256(sd) -> str:
    """sha256 over (key, shape, raw little-endian bytes) of every entry, in key order of the dict."""
    import hashlib
    h = hashlib.sha256()
    for k, v in sd.items():
        t = v.detach().cpu().contiguous()
        h.update(k.encode())
        h.update(str(tuple(t.shape)).encode())
        h.update(str(t.dtype).encode())
        h.update(t.numpy().tobytes())
    return h.hexdigest()


@torch.no_grad()
def calibrate_bn(net: nn.Module, batches: Sequence[torch.Tensor]) -> None:
    """Give BatchNorm layers data-dependent ("trained-like") running statistics: train-mode passes
    with momentum=None (cumulative average), then back to eval.  Not a reference function; it only
    produces a second, harder weight set for parity tests."""
    bns = [m for m in net.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d))]
    for m in bns:
        m.reset_running_stats()
        m.momentum = None
    net.train()
    for b in batches:
        net(b)
    net.eval()
    for m in bns:
        m.momentum = 0.1


def trained_like(sd, seed: int = 2024):
    """A deterministic, seed-regenerable "trained-like" variant of a state_dict.  Not a
    reference function: it only produces a harder weight set than the reference's init
    (models/sphereface.py:89-96: N(0, 0.01) bias-free convs, PReLU slopes 0.25, zero biases; mobilenet init
    models/mobilenetv3.py:163-175) so that the precision policy is tested on weights with O(1) differences in
    dynamic range between channels.  The SAME function is applied to the reference model's state_dict in
    tests/golden/make_golden.py and to the product model's in the tests (their seeded state_dicts are identical,
    tests/golden/state_sha256.json), so only the embeddings are committed.

      * conv / linear weights: rescaled per layer to unit gain (std = g / sqrt(fan_in); g = 0.5 for the second conv
        of a SphereFace residual block so that a stage of 16 blocks stays O(1..10)), then multiplied by a per-output-
        channel log-normal factor exp(0.5 N(0,1));
      * PReLU slopes ~ U[0.05, 0.5] per channel;
      * conv / linear biases ~ N(0, 0.1); BatchNorm weight ~ U[0.5, 1.5], BatchNorm bias ~ N(0, 0.3)
        (running statistics are left alone: calibrate them afterwards with ``calibrate_bn``).
    Keys are visited in the dict's order with one generator, so the result depends only on (sd, seed)."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    keys = list(sd.keys())
    is_bn = {k.rsplit(".", 1)[0] for k in keys if k.endswith("running_mean")}
    for k in keys:
        v = sd[k].detach().clone()
        mod, leaf = k.rsplit(".", 1)
        if mod in is_bn:
            if leaf == "weight":
                v = torch.empty_like(v).uniform_(0.5, 1.5, generator=g)
            elif leaf == "bias":
                v = torch.empty_like(v).normal_(0.0, 0.3, generator=g)
        elif leaf == "weight" and v.dim() == 1:                      # PReLU slopes (one per channel)
            v = torch.empty_like(v).uniform_(0.05, 0.5, generator=g)
        elif leaf == "weight" and v.dim() in (2, 4):
            fan_in = v[0].numel()
            gain = 0.5 if ".block.1.0" in k and k.startswith("layer") else 1.0
            v = v / v.std().clamp_min(1e-30) * (gain / fan_in ** 0.5)
            ch = torch.exp(0.5 * torch.randn(v.shape[0], generator=g))
            v = v * ch.view(-1, *([1] * (v.dim() - 1)))
        elif leaf == "bias":
            v = torch.empty_like(v).normal_(0.0, 0.1, generator=g)
        out[k] = v
    return out
