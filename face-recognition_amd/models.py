"""Backbones of the hot path as *parameter containers that lower themselves to a HIP plan*.

Each class mirrors the constructor signature and the ``state_dict`` key layout of the reference
(models/mobilenetv1.py, mobilenetv2.py, mobilenetv3.py, sphereface.py, utils/layers.py) so that
reference checkpoints load unchanged (strict) and ``torch.manual_seed(s); ctor()`` reproduces the
reference's random init bit for bit.  There is NO PyTorch arithmetic here: sub-modules only hold
parameters; ``HipBackbone.forward`` runs libfr_hip.so (include/fr_hip.h) and raises if the input
is not on a GPU or the library is missing.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import _lib
from . import ops

__all__ = ["MobileNetV1", "MobileNetV2", "mobilenet_v3_small", "mobilenet_v3_large",
           "sphere20", "sphere36", "sphere64", "HipBackbone", "default_precision"]

_ACT = {None: _lib.FR_ACT_NONE, "prelu": _lib.FR_ACT_PRELU, "hswish": _lib.FR_ACT_HSWISH}


def default_precision() -> str:
    """Plan precision used when a model does not set one: env FR_PRECISION or "auto".

    "auto" picks, per backbone family, the cheapest mode that meets the parity bar (embedding cosine >= 0.9999
    against the reference, BASELINE.json north_star) on three weight sets -- raw init, BN-calibrated, and the
    seeded "trained-like" transform of oracle/backbones.py:trained_like -- as measured on the goldens of tests/golden
    (tools/precision_probe.py, MI355X; profiles/r03_precision_probe.txt):

        family       bf16 raw     bf16 BN-calibrated   bf16 trained-like   f32 (split-bf16, 3 MFMA passes), all three sets
        MobileNetV1  0.99986 (fails)  0.9961 (fails)   0.9974 (fails)      >= 0.9999999
        MobileNetV2  0.99990 (fails)  0.9973 (fails)   0.9975 (fails)      >= 0.9999999
        V3 small     0.99995          0.974  (fails)   0.965  (fails)      >= 0.9999993
        V3 large     0.99995          0.946  (fails)   0.9925 (fails)      >= 0.9999993
        SphereFace   0.99997          (no BatchNorm)   0.99997             1.0000000

    so MobileNets default to "f32" and SphereFace to "bf16".  "bf16" stays available as an explicit choice (it is
    the mode BASELINE.json configs[1] names, and it meets the bar on that config's random-init weights).  "f24"
    (MobileNets only) is the f32 arithmetic on 3-byte activation storage: >= 0.9999985 on all three sets, 25 % less
    workspace, the same speed as "f32" ("What round 3 taught"); opt-in."""
    return os.environ.get("FR_PRECISION", "auto")


def graph_max_batch() -> int:
    """Largest batch that ``model(x)`` runs as a hipGraph replay (fr_embed_graph_*).  Off by default (0): measured on
    mobilenetv3_large, a replay takes the same GPU time as the eager launches (0.60 ms at batch 1: the 59 dependent
    kernels, ~10 us each, set the latency, not the launches) and a third of the HOST time (0.07 vs 0.18-0.22 ms per
    forward), so it pays only for a serving loop whose host thread is the bottleneck: ``FR_GRAPH_MAX=64``, or
    ``model.plan(device).graph(batch)`` for explicit static buffers."""
    return int(os.environ.get("FR_GRAPH_MAX", "0"))


def default_fuse() -> bool:
    """Cross-layer fusion (depthwise + following 1x1 conv in one kernel, k_dw.hip:dwp_kernel) is
    bit-identical to the separate kernels and on by default; env FR_FUSE=0 turns it off.
    ``forward_tap`` always uses an unfused plan so that every layer output exists."""
    return os.environ.get("FR_FUSE", "1") != "0"


def _make_divisible(v: float, divisor: int = 8) -> int:
    """utils/layers.py:6-12."""
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


class _Graph:
    """Collects fr_layer_desc records; keeps the host tensors alive until the plan is created."""

    def __init__(self) -> None:
        self.descs: List[_lib.LayerDesc] = []
        self.names: List[str] = []
        self._keep: List[torch.Tensor] = []

    def _ptr(self, t: Optional[torch.Tensor]):
        if t is None:
            return None
        h = t.detach().to("cpu", torch.float32).contiguous()
        self._keep.append(h)
        return h.data_ptr()

    def add(self, name: str, **kw) -> int:
        d = _lib.LayerDesc()
        d.residual = -1
        d.kernel, d.stride, d.groups, d.bn_eps = 1, 1, 1, 1e-5
        for k, v in kw.items():
            setattr(d, k, self._ptr(v) if isinstance(v, torch.Tensor) else v)
        self.descs.append(d)
        self.names.append(name)
        return len(self.descs) - 1

    def conv(self, name: str, conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d], act: Optional[nn.Module],
             src: int, residual: int = -1) -> int:
        kind = None if act is None else ("prelu" if isinstance(act, nn.PReLU) else "hswish")
        kw = dict(op=_lib.FR_OP_CONV, src=src, residual=residual, cin=conv.in_channels, cout=conv.out_channels,
                  kernel=conv.kernel_size[0], stride=conv.stride[0], pad=conv.padding[0], groups=conv.groups,
                  act=_ACT[kind], weight=conv.weight, bias=conv.bias)
        if bn is not None:
            kw.update(bn_weight=bn.weight, bn_bias=bn.bias, bn_mean=bn.running_mean, bn_var=bn.running_var,
                      bn_eps=float(bn.eps))
        if kind == "prelu":
            kw["prelu"] = act.weight
        return self.add(name, **kw)


class ConvUnit(nn.Sequential):
    """Conv2d -> BatchNorm2d? -> PReLU(per channel)|Hardswish?  (utils/layers.py:15-84).
    Child indices "0","1","2" are the checkpoint keys."""

    def __init__(self, cin: int, cout: int, kernel_size: int = 3, stride: int = 1, groups: int = 1,
                 norm: bool = True, act: Optional[str] = "prelu", bias: bool = True) -> None:
        mods: List[nn.Module] = [nn.Conv2d(cin, cout, kernel_size, stride, (kernel_size - 1) // 2,
                                           groups=groups, bias=bias)]
        if norm:
            mods.append(nn.BatchNorm2d(cout))
        if act == "prelu":
            mods.append(nn.PReLU(num_parameters=cout))
        elif act == "hswish":
            mods.append(nn.Hardswish(inplace=True))
        super().__init__(*mods)

    def lower(self, g: _Graph, name: str, src: int, residual: int = -1) -> int:
        bn = next((m for m in self if isinstance(m, nn.BatchNorm2d)), None)
        act = next((m for m in self if isinstance(m, (nn.PReLU, nn.Hardswish))), None)
        return g.conv(name, self[0], bn, act, src, residual)


class SEUnit(nn.Module):
    """utils/layers.py:87-122 (Hardsigmoid scale, models/mobilenetv3.py:46)."""

    def __init__(self, channels: int, squeeze: int) -> None:
        super().__init__()
        self.fc1 = nn.Conv2d(channels, squeeze, 1)
        self.fc2 = nn.Conv2d(squeeze, channels, 1)

    def lower(self, g: _Graph, name: str, src: int) -> int:
        return g.add(name, op=_lib.FR_OP_SE, src=src, cin=self.fc1.in_channels, cout=self.fc1.out_channels,
                     weight=self.fc1.weight, bias=self.fc1.bias, weight2=self.fc2.weight, bias2=self.fc2.bias)


class _LinearBlock(nn.Module):
    """utils/layers.py:156-173."""

    def __init__(self, cin: int, cout: int, kernel_size: int, groups: int) -> None:
        super().__init__()
        self.layers = nn.Sequential(nn.Conv2d(cin, cout, kernel_size, 1, 0, groups=groups, bias=False),
                                    nn.BatchNorm2d(cout))


class GDC(nn.Module):
    """utils/layers.py:176-191."""

    def __init__(self, in_channels: int, embedding_dim: int) -> None:
        super().__init__()
        self.features = nn.Sequential(_LinearBlock(in_channels, in_channels, 7, in_channels), nn.Flatten())
        self.fc = nn.Sequential(nn.Linear(in_channels, embedding_dim, bias=False), nn.BatchNorm1d(embedding_dim))

    def lower(self, g: _Graph, name: str, src: int) -> int:
        lb = self.features[0].layers
        i = g.conv(f"{name}.features.0", lb[0], lb[1], None, src)
        lin, bn = self.fc[0], self.fc[1]
        return g.add(f"{name}.fc", op=_lib.FR_OP_LINEAR, src=i, cin=lin.in_features, cout=lin.out_features,
                     weight=lin.weight, bias=lin.bias, bn_weight=bn.weight, bn_bias=bn.bias,
                     bn_mean=bn.running_mean, bn_var=bn.running_var, bn_eps=float(bn.eps))


# ------------------------------------------------------------------------------------------------
class HipBackbone(nn.Module):
    """float32[N,3,112,112] (NCHW, cuda) -> float32[N,embedding_dim], computed by libfr_hip.so."""

    precision: Optional[str] = None      # "bf16" | "f32" | "auto"; None -> default_precision()
    auto_precision: str = "f32"          # what "auto" resolves to for this family (see default_precision)

    def __init__(self) -> None:
        super().__init__()
        self._plans: Dict[Tuple[int, str, bool], "_Plan"] = {}

    # -- lowering ------------------------------------------------------------------------------
    def lower(self, g: _Graph) -> None:  # pragma: no cover - overridden
        raise NotImplementedError

    def layer_names(self) -> List[str]:
        g = _Graph()
        self.lower(g)
        return g.names

    def invalidate(self) -> None:
        """Drop compiled plans (call after mutating parameters in place)."""
        for p in self._plans.values():
            p.close()
        self._plans = {}

    def _apply(self, fn, *a, **kw):
        self.invalidate()
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        self.invalidate()
        return super().load_state_dict(*a, **kw)

    def plan(self, device: torch.device, precision: Optional[str] = None, fuse: Optional[bool] = None) -> "_Plan":
        prec = precision or self.precision or default_precision()
        if prec == "auto":
            prec = self.auto_precision
        fuse = default_fuse() if fuse is None else fuse
        if prec not in _lib.PRECISIONS:
            raise ValueError(f"unknown precision '{prec}' (expected one of {list(_lib.PRECISIONS)})")
        key = (device.index if device.index is not None else torch.cuda.current_device(), prec, bool(fuse))
        if key not in self._plans:
            g = _Graph()
            self.lower(g)
            self._plans[key] = _Plan(g, key[0], prec, fuse)
        return self._plans[key]

    def forward(self, x: torch.Tensor, precision: Optional[str] = None, fuse: Optional[bool] = None) -> torch.Tensor:
        if self.training:
            raise RuntimeError("HipBackbone runs inference only: call .eval() first (reference: inference.py:92)")
        if not x.is_cuda:
            raise RuntimeError("HipBackbone.forward needs a GPU tensor: the hot path has no CPU fallback")
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, 112, 112):
            raise ValueError(f"expected input [N,3,112,112], got {tuple(x.shape)}")
        return self.plan(x.device, precision, fuse).run(x)

    def forward_u8(self, x: torch.Tensor, precision: Optional[str] = None, fuse: Optional[bool] = None) -> torch.Tensor:
        """uint8 [N,112,112,3] RGB faces on the GPU -> embeddings: ``model(transform(img))`` of
        inference.py:98-106,:120-122 with ToTensor + Normalize applied inside the stem kernel's load
        (fr_embed_u8).  Bit-identical to ``forward`` on the transformed tensor."""
        if self.training:
            raise RuntimeError("HipBackbone runs inference only: call .eval() first (reference: inference.py:92)")
        if not x.is_cuda:
            raise RuntimeError("HipBackbone.forward_u8 needs a GPU tensor: the hot path has no CPU fallback")
        if x.dtype != torch.uint8 or x.dim() != 4 or tuple(x.shape[1:]) != (112, 112, 3):
            raise ValueError(f"expected uint8 input [N,112,112,3], got {x.dtype} {tuple(x.shape)}")
        return self.plan(x.device, precision, fuse).run_u8(x)

    def forward_tap(self, x: torch.Tensor, layer: int, precision: Optional[str] = None):
        """(embedding, output of plan layer ``layer`` as fp32 NCHW) -- parity tests."""
        return self.plan(x.device, precision, fuse=False).run(x, tap=layer)


class _Plan:
    def __init__(self, g: _Graph, device_index: int, precision: str, fuse: bool = True) -> None:
        self.lib = _lib.lib()
        self.device = torch.device("cuda", device_index)
        self.names = list(g.names)
        self.srcs = [int(d.src) for d in g.descs]
        self.ops = [int(d.op) for d in g.descs]
        self.n_layers = len(g.descs)
        arr = (_lib.LayerDesc * self.n_layers)(*g.descs)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.fr_plan_create(arr, self.n_layers, 3, 112, 112, _lib.PRECISIONS[precision] | (0 if fuse else _lib.FR_PLAN_NO_FUSE),
                                               C.byref(handle)), "fr_plan_create")
        self.handle = handle
        self.precision = precision
        self._ws: Optional[torch.Tensor] = None
        self._graphs: Dict[Tuple[int, bool, int], "EmbedGraph"] = {}
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        self.dims = []
        for i in range(self.n_layers):
            _lib.check(self.lib.fr_plan_layer_dims(self.handle, i, C.byref(c), C.byref(h), C.byref(w)), "fr_plan_layer_dims")
            self.dims.append((c.value, h.value, w.value))

    def close(self) -> None:
        if self.handle:
            for g in self._graphs.values():
                g.close()
            self._graphs = {}
            ops.drop_workspace(int(self.handle.value))
            self.lib.fr_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def workspace(self, batch: int) -> torch.Tensor:
        need = C.c_size_t()
        _lib.check(self.lib.fr_plan_workspace_bytes(self.handle, batch, C.byref(need)), "fr_plan_workspace_bytes")
        if self._ws is None or self._ws.numel() < need.value:
            self._ws = None
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=self.device)
        return self._ws

    def profile(self, x: torch.Tensor):
        """Per-layer device milliseconds of one fr_embed call: [(name, kernel, ms, src_index)]."""
        x = x.to(torch.float32).contiguous()
        n = x.shape[0]
        out = torch.empty((n, self.dims[-1][0]), dtype=torch.float32, device=x.device)
        ws = self.workspace(n)
        ms = (C.c_float * self.n_layers)()
        with torch.cuda.device(x.device):
            _lib.check(self.lib.fr_embed_profile(self.handle, x.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 torch.cuda.current_stream(x.device).cuda_stream, ms), "fr_embed_profile")
        rows = []
        nm = C.c_char_p()
        for i in range(self.n_layers):
            _lib.check(self.lib.fr_plan_layer_kernel(self.handle, i, C.byref(nm)), "fr_plan_layer_kernel")
            rows.append((self.names[i], nm.value.decode(), float(ms[i]), self.srcs[i]))
        return rows

    def graph(self, batch: int, u8: bool = False) -> "EmbedGraph":
        """The hipGraph replay of this plan at ``batch`` for the current stream (created on first use)."""
        key = (batch, u8, torch.cuda.current_stream(self.device).cuda_stream)
        g = self._graphs.get(key)
        if g is None:
            g = self._graphs[key] = EmbedGraph(self, batch, u8)
        return g

    def _replay(self, n: int) -> bool:
        return 0 < n <= graph_max_batch() and not torch.cuda.is_current_stream_capturing()

    def run_u8(self, x: torch.Tensor) -> torch.Tensor:
        if self._replay(x.shape[0]):
            with torch.cuda.device(self.device):
                return self.graph(x.shape[0], True)(x).clone()
        c, h, w = self.dims[-1]
        return torch.ops.fr_hip.embed_u8(x, int(self.handle.value), c, h, w)        # ops.py -> fr_embed_u8

    def run(self, x: torch.Tensor, tap: int = -1):
        x = x.to(torch.float32).contiguous()
        n = x.shape[0]
        out = torch.empty((n,) + ((self.dims[-1][0],) if self.dims[-1][1:] == (1, 1) else self.dims[-1]),
                          dtype=torch.float32, device=x.device)
        if n == 0:
            return out if tap < 0 else (out, None)
        if tap < 0:
            if self._replay(n):
                with torch.cuda.device(self.device):
                    return self.graph(n, False)(x).clone()
            c, h, w = self.dims[-1]
            return torch.ops.fr_hip.embed(x, int(self.handle.value), c, h, w)      # ops.py -> fr_embed
        ws = self.workspace(n)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            tap_out = torch.empty((n,) + self.dims[tap], dtype=torch.float32, device=x.device)
            _lib.check(self.lib.fr_embed_tap(self.handle, x.data_ptr(), n, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                             stream, tap, tap_out.data_ptr()), "fr_embed_tap")
            return out, tap_out


class EmbedGraph:
    """One plan at one batch size as a hipGraph bound to static buffers (include/fr_hip.h: fr_embed_graph_*): the
    serving-loop form of ``model(x)`` for the reference's per-image loops (inference.py:150-153, evaluate.py:517-555).
    ``g(x)`` copies ``x`` (GPU or pinned/pageable CPU tensor of the static shape) into ``g.x``, enqueues one replay on the
    current stream and returns ``g.out`` -- the SAME tensor every call, overwritten by the next one."""

    def __init__(self, plan: "_Plan", batch: int, u8: bool = False) -> None:
        self.plan, self.batch, self.u8 = plan, batch, u8
        dev = plan.device
        self.x = (torch.zeros((batch, 112, 112, 3), dtype=torch.uint8, device=dev) if u8
                  else torch.zeros((batch, 3, 112, 112), dtype=torch.float32, device=dev))
        d = plan.dims[-1]
        self.out = torch.empty((batch,) + ((d[0],) if d[1:] == (1, 1) else d), dtype=torch.float32, device=dev)
        need = C.c_size_t()
        _lib.check(plan.lib.fr_plan_workspace_bytes(plan.handle, batch, C.byref(need)), "fr_plan_workspace_bytes")
        self.ws = torch.empty(need.value, dtype=torch.uint8, device=dev)
        self.handle = C.c_void_p()
        torch.cuda.current_stream(dev).synchronize()          # the buffers exist before the private capture stream touches them
        with torch.cuda.device(dev):
            _lib.check(plan.lib.fr_embed_graph_create(plan.handle, self.x.data_ptr(), int(u8), batch, self.out.data_ptr(),
                                                      self.ws.data_ptr(), self.ws.numel(), C.byref(self.handle)), "fr_embed_graph_create")
        self.nodes = int(plan.lib.fr_embed_graph_nodes(self.handle))

    def launch(self) -> torch.Tensor:
        with torch.cuda.device(self.plan.device):
            _lib.check(self.plan.lib.fr_embed_graph_launch(self.handle, torch.cuda.current_stream(self.plan.device).cuda_stream),
                       "fr_embed_graph_launch")
        return self.out

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if tuple(x.shape) != tuple(self.x.shape) or x.dtype != self.x.dtype:
            raise ValueError(f"EmbedGraph: expected {self.x.dtype} {tuple(self.x.shape)}, got {x.dtype} {tuple(x.shape)}")
        self.x.copy_(x, non_blocking=True)
        return self.launch()

    def close(self) -> None:
        if self.handle:
            self.plan.lib.fr_embed_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _mobilenet_init(net: nn.Module) -> None:
    """models/mobilenetv1.py:47-59 == mobilenetv2.py:148-159 == mobilenetv3.py:163-175."""
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


# ----------------------------------------------------------------------------------- MobileNetV1
class _DWSeparable(nn.Sequential):
    """utils/layers.py:125-153."""

    def __init__(self, cin: int, cout: int, stride: int) -> None:
        super().__init__(ConvUnit(cin, cin, 3, stride, groups=cin), ConvUnit(cin, cout, 1))

    def lower(self, g: _Graph, name: str, src: int) -> int:
        return self[1].lower(g, f"{name}.1", self[0].lower(g, f"{name}.0", src))


class MobileNetV1(HipBackbone):
    """models/mobilenetv1.py:10-67."""

    def __init__(self, embedding_dim: int = 512, width_mult: float = 0.25) -> None:
        super().__init__()
        f = [_make_divisible(c * width_mult) for c in (32, 64, 128, 256, 512, 1024)]
        self.stage1 = nn.Sequential(ConvUnit(3, f[0], 3, 1), _DWSeparable(f[0], f[1], 1), _DWSeparable(f[1], f[2], 2),
                                    _DWSeparable(f[2], f[2], 1), _DWSeparable(f[2], f[3], 2), _DWSeparable(f[3], f[3], 1))
        self.stage2 = nn.Sequential(_DWSeparable(f[3], f[4], 2), *[_DWSeparable(f[4], f[4], 1) for _ in range(5)])
        self.stage3 = nn.Sequential(_DWSeparable(f[4], f[5], 2), _DWSeparable(f[5], f[5], 1))
        self.output_layer = GDC(f[5], embedding_dim)
        _mobilenet_init(self)

    def lower(self, g: _Graph) -> None:
        cur = -1
        for sname in ("stage1", "stage2", "stage3"):
            for i, m in enumerate(getattr(self, sname)):
                cur = m.lower(g, f"{sname}.{i}", cur)
        self.output_layer.lower(g, "output_layer", cur)


# ----------------------------------------------------------------------------------- MobileNetV2
class _V2InvertedResidual(nn.Module):
    """models/mobilenetv2.py:11-57."""

    def __init__(self, cin: int, cout: int, stride: int, expand_ratio: int) -> None:
        super().__init__()
        hidden = int(round(cin * expand_ratio))
        self.use_res_connect = stride == 1 and cin == cout
        mods: List[nn.Module] = []
        if expand_ratio != 1:
            mods.append(ConvUnit(cin, hidden, 1))
        mods += [ConvUnit(hidden, hidden, 3, stride, groups=hidden),
                 nn.Conv2d(hidden, cout, 1, 1, 0, bias=False), nn.BatchNorm2d(cout)]
        self.conv = nn.Sequential(*mods)

    def lower(self, g: _Graph, name: str, src: int) -> int:
        cur = src
        n = len(self.conv)
        for j in range(n - 2):
            cur = self.conv[j].lower(g, f"{name}.conv.{j}", cur)
        return g.conv(f"{name}.conv.{n - 1}", self.conv[n - 2], self.conv[n - 1], None, cur,
                      src if self.use_res_connect else -1)


class MobileNetV2(HipBackbone):
    """models/mobilenetv2.py:60-165."""

    def __init__(self, embedding_dim: int = 512, width_mult: float = 1.0, round_nearest: int = 8,
                 dropout: float = 0.2) -> None:
        super().__init__()
        setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2], [6, 320, 1, 1]]
        cin = _make_divisible(32 * width_mult, round_nearest)
        self.last_channel = _make_divisible(512 * max(1.0, width_mult), round_nearest)
        feats: List[nn.Module] = [ConvUnit(3, cin, 3, 1)]
        for t, c, n, s in setting:
            cout = _make_divisible(c * width_mult, round_nearest)
            for i in range(n):
                feats.append(_V2InvertedResidual(cin, cout, s if i == 0 else 1, t))
                cin = cout
        feats.append(ConvUnit(cin, self.last_channel, 1))
        self.features = nn.Sequential(*feats)
        self.output_layer = GDC(self.last_channel, embedding_dim)
        _mobilenet_init(self)

    def lower(self, g: _Graph) -> None:
        cur = -1
        for i, m in enumerate(self.features):
            cur = m.lower(g, f"features.{i}", cur)
        self.output_layer.lower(g, "output_layer", cur)


# ----------------------------------------------------------------------------------- MobileNetV3
class _V3InvertedResidual(nn.Module):
    """models/mobilenetv3.py:41-98."""

    def __init__(self, cin: int, kernel: int, exp: int, cout: int, use_se: bool, activation: str, stride: int) -> None:
        super().__init__()
        act = "hswish" if activation == "HS" else "prelu"
        self.use_res_connect = stride == 1 and cin == cout
        mods: List[nn.Module] = []
        if exp != cin:
            mods.append(ConvUnit(cin, exp, 1, act=act))
        mods.append(ConvUnit(exp, exp, kernel, stride, groups=exp, act=act))
        if use_se:
            mods.append(SEUnit(exp, _make_divisible(exp // 4, 8)))
        mods.append(ConvUnit(exp, cout, 1, act=None))
        self.block = nn.Sequential(*mods)

    def lower(self, g: _Graph, name: str, src: int) -> int:
        cur = src
        last = len(self.block) - 1
        for j, m in enumerate(self.block):
            if j == last:
                cur = m.lower(g, f"{name}.block.{j}", cur, src if self.use_res_connect else -1)
            else:
                cur = m.lower(g, f"{name}.block.{j}", cur)
        return cur


_V3_SETTINGS = {
    "mobilenet_v3_large": [
        (16, 3, 16, 16, False, "RE", 1), (16, 3, 64, 24, False, "RE", 2), (24, 3, 72, 24, False, "RE", 1),
        (24, 5, 72, 40, True, "RE", 2), (40, 5, 120, 40, True, "RE", 1), (40, 5, 120, 40, True, "RE", 1),
        (40, 3, 240, 80, False, "HS", 2), (80, 3, 200, 80, False, "HS", 1), (80, 3, 184, 80, False, "HS", 1),
        (80, 3, 184, 80, False, "HS", 1), (80, 3, 480, 112, True, "HS", 1), (112, 3, 672, 112, True, "HS", 1),
        (112, 5, 672, 160, True, "HS", 2), (160, 5, 960, 160, True, "HS", 1), (160, 5, 960, 160, True, "HS", 1)],
    "mobilenet_v3_small": [
        (16, 3, 16, 16, True, "RE", 2), (16, 3, 72, 24, False, "RE", 2), (24, 3, 88, 24, False, "RE", 1),
        (24, 5, 96, 40, True, "HS", 2), (40, 5, 240, 40, True, "HS", 1), (40, 5, 240, 40, True, "HS", 1),
        (40, 5, 120, 48, True, "HS", 1), (48, 5, 144, 48, True, "HS", 1), (48, 5, 288, 96, True, "HS", 2),
        (96, 5, 576, 96, True, "HS", 1), (96, 5, 576, 96, True, "HS", 1)],
}


class MobileNetV3(HipBackbone):
    """models/mobilenetv3.py:101-181."""

    def __init__(self, arch: str, embedding_dim: int = 512) -> None:
        super().__init__()
        rows = _V3_SETTINGS[arch]
        feats: List[nn.Module] = [ConvUnit(3, rows[0][0], 3, 1, act="hswish")]
        feats += [_V3InvertedResidual(*r) for r in rows]
        last = 6 * rows[-1][3]
        feats.append(ConvUnit(rows[-1][3], last, 1, act="hswish"))
        self.features = nn.Sequential(*feats)
        self.output_layer = GDC(last, embedding_dim)
        _mobilenet_init(self)

    def lower(self, g: _Graph) -> None:
        cur = -1
        for i, m in enumerate(self.features):
            cur = m.lower(g, f"features.{i}", cur)
        self.output_layer.lower(g, "output_layer", cur)


def mobilenet_v3_large(embedding_dim: int = 512, **kwargs) -> MobileNetV3:
    """models/mobilenetv3.py:237-240."""
    return MobileNetV3("mobilenet_v3_large", embedding_dim)


def mobilenet_v3_small(embedding_dim: int = 512, **kwargs) -> MobileNetV3:
    """models/mobilenetv3.py:243-246."""
    return MobileNetV3("mobilenet_v3_small", embedding_dim)


# ------------------------------------------------------------------------------------ SphereFace
class _SphereResidual(nn.Module):
    """models/sphereface.py:10-48."""

    def __init__(self, channels: int) -> None:
        super().__init__()
        self.block = nn.Sequential(ConvUnit(channels, channels, 3, norm=False, bias=False),
                                   ConvUnit(channels, channels, 3, norm=False, bias=False))

    def lower(self, g: _Graph, name: str, src: int) -> int:
        a = self.block[0].lower(g, f"{name}.block.0", src)
        return self.block[1].lower(g, f"{name}.block.1", a, src)


class SphereNet(HipBackbone):
    """models/sphereface.py:51-129."""

    auto_precision = "bf16"              # no BatchNorm, K >= 576 dense convs: bf16 measures cos >= 0.99997 on raw-init AND on
    # trained-like weights (tests/golden/trained_sphere*.npz; test_default_precision_meets_the_bar_on_trained_like_weights)

    def __init__(self, layers: Sequence[int], embedding_dim: int = 512, in_channels: int = 3) -> None:
        super().__init__()
        cin = in_channels
        for i, (c, n) in enumerate(zip((64, 128, 256, 512), layers), start=1):
            setattr(self, f"layer{i}", nn.Sequential(ConvUnit(cin, c, 3, 2, norm=False),
                                                     *[_SphereResidual(c) for _ in range(n)]))
            cin = c
        self.fc = nn.Linear(512 * 7 * 7, embedding_dim)
        for m in self.modules():
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                if m.bias is not None:
                    nn.init.xavier_uniform_(m.weight)
                    nn.init.constant_(m.bias, 0.0)
                else:
                    nn.init.normal_(m.weight, 0, 0.01)

    def lower(self, g: _Graph) -> None:
        cur = -1
        for i in (1, 2, 3, 4):
            for j, m in enumerate(getattr(self, f"layer{i}")):
                cur = m.lower(g, f"layer{i}.{j}", cur)
        g.add("fc", op=_lib.FR_OP_LINEAR, src=cur, cin=self.fc.in_features, cout=self.fc.out_features,
              weight=self.fc.weight, bias=self.fc.bias)


def sphere20(embedding_dim, **kwargs):
    """models/sphereface.py:132-144."""
    return SphereNet([1, 2, 4, 1], embedding_dim, **kwargs)


def sphere36(embedding_dim, **kwargs):
    """models/sphereface.py:147-159."""
    return SphereNet([2, 4, 8, 2], embedding_dim, **kwargs)


def sphere64(embedding_dim, **kwargs):
    """models/sphereface.py:162-173."""
    return SphereNet([3, 8, 16, 3], embedding_dim, **kwargs)
