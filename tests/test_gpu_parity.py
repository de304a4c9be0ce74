"""GPU parity tests: the HIP path (through the C ABI of libfr_hip.so) against the CPU oracle and the
golden vectors captured from the real reference.  Bars: embedding cosine >= 0.9999,
similarity |delta| <= 1e-3, decisions at 0.35 identical, top-5 indices identical where the score gap
exceeds 1e-3; integer/bit work (synthetic rows) bit-exact."""
import os

import numpy as np
import pytest
import torch

import face_recognition_amd as fr
from oracle import backbones as ob
from oracle import scoring as osc
from oracle import search as osr

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
COS_BAR = 0.9999


def cos_rows(a, b):
    a = np.asarray(a, np.float64).reshape(len(a), -1); b = np.asarray(b, np.float64).reshape(len(b), -1)
    return (a * b).sum(1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1) + 1e-300)


def product(name, state_dict=None):
    torch.manual_seed(0)
    m = fr.get_network(name)
    if state_dict is not None:
        m.load_state_dict(state_dict)
    return m.to(DEV).eval()


def layer_report(model, oracle, x, precision):
    """Per-plan-layer cosine / relative max error against oracle forward hooks."""
    names = model.layer_names()
    mods = dict(oracle.named_modules())
    # a unit that closes a residual block is compared at the block's output (the oracle hook on the
    # unit itself fires before the skip connection is added)
    where = {}
    for n in names:
        if n not in mods:
            continue
        parent = n.rsplit(".", 2)[0]
        closes = getattr(mods.get(parent), "residual", False) and n == [k for k in names if k.startswith(parent + ".")][-1]
        where[n] = parent if closes else n
    _, taps_raw = ob.forward_with_taps(oracle, x, sorted(set(where.values())))
    taps = {n: taps_raw[w] for n, w in where.items()}
    rows = []
    for i, n in enumerate(names):
        if n not in taps or isinstance(mods[n], ob.SqueezeExcite):
            continue
        _, t = model.forward_tap(x.to(DEV), i, precision)
        ref = taps[n].numpy().reshape(len(x), -1)
        got = t.cpu().numpy().reshape(len(x), -1)
        rel = np.abs(got - ref).max() / (np.abs(ref).max() + 1e-300)
        rows.append((i, n, cos_rows(got, ref).min(), rel))
    return rows


@pytest.mark.parametrize("precision,bar,relbar", [("f32", 0.999999, 1e-3), ("f24", 0.999999, 1e-3), ("bf16", 0.9995, 0.1)])
@pytest.mark.parametrize("name", ["mobilenetv3_large", "mobilenetv3_small", "mobilenetv2", "mobilenetv1", "sphere20"])
def test_every_layer_against_oracle_hooks(name, precision, bar, relbar, golden_x):
    """Each fused unit (conv+BN+act, project+residual, SE-scaled project, GDC) vs the oracle's hook at
    the same point.  f32 mode must track to ~1e-6 (f24: the same arithmetic on activations stored with 16
    significant bits in two planes); bf16 mode within bf16 rounding accumulation."""
    if precision == "f24" and name.startswith("sphere"):
        pytest.skip("f24 storage is the depthwise-separable families' mode (dense 3x3 convs have no f24 kernel)")
    x = golden_x[:3]
    rows = layer_report(product(name), ob.seeded(name, 0), x, precision)
    bad = [r for r in rows if not (r[2] >= bar and r[3] <= relbar)]
    msg = "\n".join(f"{i:3d} {n:40s} cos={c:.8f} relmax={e:.3e}" for i, n, c, e in rows)
    assert not bad, f"{name} {precision}: layers off:\n{msg}"


@pytest.mark.parametrize("name", ["mobilenetv1", "mobilenetv2", "mobilenetv3_small", "mobilenetv3_large"])
def test_embeddings_vs_reference_golden_f32(name, golden_dir, golden_x):
    g = np.load(os.path.join(golden_dir, f"emb_{name}.npz"))
    with torch.no_grad():
        y = product(name)(golden_x.to(DEV), precision="f32").cpu().numpy()
    c = cos_rows(y, g["emb"])
    assert c.min() >= COS_BAR, c
    assert np.abs(np.linalg.norm(y, axis=1) / np.linalg.norm(g["emb"], axis=1) - 1).max() < 1e-2


@pytest.mark.parametrize("name", ["sphere20", "sphere36", "sphere64"])
def test_sphereface_embeddings_vs_reference_golden(name, golden_dir, golden_x):
    """BASELINE config 3 family: dense 3x3 MFMA path + PReLU + residual + NCHW-flatten fc."""
    g = np.load(os.path.join(golden_dir, f"emb_{name}.npz"))
    m = product(name)
    with torch.no_grad():
        for prec in ("bf16", "f32"):
            y = m(golden_x.to(DEV), precision=prec).cpu().numpy()
            c = cos_rows(y, g["emb"])
            assert c.min() >= COS_BAR, (prec, c)
            assert np.abs(np.linalg.norm(y, axis=1) / np.linalg.norm(g["emb"], axis=1) - 1).max() < 1e-2


@pytest.mark.parametrize("name", ["mobilenetv3_small", "mobilenetv3_large"])
def test_embeddings_vs_reference_golden_bf16(name, golden_dir, golden_x):
    """BASELINE configs 1-2 name the bf16 path: single-pass bf16 operands, fp32 accumulate."""
    g = np.load(os.path.join(golden_dir, f"emb_{name}.npz"))
    with torch.no_grad():
        y = product(name)(golden_x.to(DEV), precision="bf16").cpu().numpy()
    c = cos_rows(y, g["emb"])
    assert c.min() >= COS_BAR, c


@pytest.mark.parametrize("name,batch", [("mobilenetv3_large", 1024), ("sphere64", 512)])
def test_baseline_batch_sizes_by_position_invariance(name, batch, golden_dir, golden_inputs, golden_x):
    """BASELINE.json configs[1] / configs[2] at their full batch sizes, checked through a size-independent property:
    an image's embedding does not depend on where in the batch it sits.  The batch repeats the 8 golden inputs (plus
    3 noise images), so every repeat must equal the first occurrence BIT FOR BIT -- across the two sub-batches that
    fr_embed runs on two streams, across workgroup tile boundaries and the XCD-aware block orders -- and the first
    occurrences must match the reference's golden embeddings to the cosine bar.  Same for the uint8 entry point."""
    g = np.load(os.path.join(golden_dir, f"emb_{name}.npz"))
    m = product(name)
    rng = np.random.default_rng(9)
    u8 = np.concatenate([golden_inputs, rng.integers(0, 256, size=(3, 112, 112, 3), dtype=np.uint8)])      # period 11: odd against every tile size
    P = len(u8)
    reps = (batch + P - 1) // P
    big_u8 = torch.from_numpy(np.ascontiguousarray(np.tile(u8, (reps, 1, 1, 1))[:batch])).to(DEV)
    xp = torch.stack([osc.to_model_input(im) for im in u8]).to(DEV)
    big = xp.repeat(reps, 1, 1, 1)[:batch].contiguous()
    with torch.no_grad():
        y = m(big, precision="bf16")
        y8 = m.forward_u8(big_u8, precision="bf16")
        y_small = m(xp, precision="bf16")
    want = y[:P].repeat(reps, 1)[:batch]
    assert torch.equal(y, want)
    assert torch.equal(y8, y)
    assert torch.equal(y[:P], y_small)                      # and not on the batch size either (11 images: one stream)
    c = cos_rows(y[:8].cpu().numpy(), g["emb"])
    assert c.min() >= COS_BAR, c


@pytest.mark.parametrize("name", ["mobilenetv1", "mobilenetv2", "mobilenetv3_small", "mobilenetv3_large"])
def test_bn_calibrated_weights_f32(name, golden_dir, golden_x):
    """'Trained-like' BatchNorm statistics captured from the reference (harder: real cancellation)."""
    g = np.load(os.path.join(golden_dir, f"bncal_{name}.npz"))
    sd = ob.seeded(name, 0).state_dict()
    for i, k in enumerate(g["keys"]):
        sd[str(k)] = torch.from_numpy(g[f"s{i}"])
    with torch.no_grad():
        y = product(name, sd)(golden_x.to(DEV), precision="f32").cpu().numpy()
    c = cos_rows(y, g["emb"])
    assert c.min() >= COS_BAR, c
    assert np.abs(y - g["emb"]).max() <= 2e-3 * np.abs(g["emb"]).max()


def _calibrated_state(name, golden_dir):
    g = np.load(os.path.join(golden_dir, f"bncal_{name}.npz"))
    sd = ob.seeded(name, 0).state_dict()
    for i, k in enumerate(g["keys"]):
        sd[str(k)] = torch.from_numpy(g[f"s{i}"])
    return sd, g["emb"]


@pytest.mark.parametrize("calibrated", [False, True])
@pytest.mark.parametrize("name", ob.MODEL_NAMES)
def test_default_precision_meets_the_bar_on_every_backbone(name, calibrated, golden_dir, golden_x):
    """What load_model(...) -> model(x) runs when nobody chooses a precision ("auto", models.default_precision):
    every backbone x {raw init, BN-calibrated 'trained-like' statistics} against the reference's golden embeddings
    at the north_star bar, cosine >= 0.9999.  (SphereFace has no BatchNorm, hence no calibrated variant.)"""
    if calibrated and name.startswith("sphere"):
        pytest.skip("SphereFace has no BatchNorm: there is no calibrated variant")
    if calibrated:
        sd, want = _calibrated_state(name, golden_dir)
    else:
        sd, want = None, np.load(os.path.join(golden_dir, f"emb_{name}.npz"))["emb"]
    m = product(name, sd)
    assert m.precision is None and fr.default_precision() == "auto"
    with torch.no_grad():
        y = m(golden_x.to(DEV)).cpu().numpy()
    c = cos_rows(y, want)
    assert c.min() >= COS_BAR, (m.plan(DEV).precision, c)
    assert m.plan(DEV).precision == ("bf16" if name.startswith("sphere") else "f32")


@pytest.mark.parametrize("name", ob.MODEL_NAMES)
def test_default_precision_meets_the_bar_on_trained_like_weights(name, golden_dir, golden_x):
    """The default precision on the "trained-like" weight set (oracle.backbones.trained_like, seed 2024: per-channel
    log-normal weight scales, PReLU slopes U[0.05, 0.5], non-zero biases, BatchNorm affine away from (1, 0) and
    statistics calibrated on THOSE weights) against the REFERENCE's embeddings on the same state_dict
    (tests/golden/trained_<name>.npz).  This is the SphereFace case the raw init cannot give: residual branches as
    large as the trunk, O(1) differences in dynamic range between channels.  Also checked with the common component
    of the 8 embeddings removed (their mean), which is what separates two faces."""
    from test_oracle_golden import trained_like_state
    sd, want = trained_like_state(name, golden_dir)
    m = product(name, sd)
    with torch.no_grad():
        y = m(golden_x.to(DEV)).cpu().numpy()
    c = cos_rows(y, want)
    mu = want.astype(np.float64).mean(0)
    cc = cos_rows(y - mu, want - mu)
    assert c.min() >= COS_BAR, (m.plan(DEV).precision, c)
    assert cc.min() >= 0.999, (m.plan(DEV).precision, cc)
    assert np.abs(np.linalg.norm(y, axis=1) / np.linalg.norm(want, axis=1) - 1).max() < 1e-2


@pytest.mark.parametrize("weights", ["raw", "bncal", "trained"])
@pytest.mark.parametrize("name", ["mobilenetv1", "mobilenetv2", "mobilenetv3_small", "mobilenetv3_large"])
def test_f24_storage_meets_the_bar(name, weights, golden_dir, golden_x, golden_inputs):
    """FR_PREC_F24 (activations stored as the top 24 bits of fp32 in a bf16 plane + a byte plane, the arithmetic of the
    f32 plans) against the reference's goldens on all three weight sets, at the north_star bar; the uint8 entry point
    gives the same bits; a batch of 300 (two sub-batches, ragged tiles) repeats the 8-image result bit for bit; and
    the dense-3x3 families refuse the mode with a clear error."""
    from test_oracle_golden import trained_like_state
    if weights == "raw":
        sd, want = None, np.load(os.path.join(golden_dir, f"emb_{name}.npz"))["emb"]
    elif weights == "bncal":
        sd, want = _calibrated_state(name, golden_dir)
    else:
        sd, want = trained_like_state(name, golden_dir)
    m = product(name, sd)
    x = golden_x.to(DEV)
    with torch.no_grad():
        y = m(x, precision="f24")
        y8 = m.forward_u8(torch.from_numpy(golden_inputs).to(DEV), precision="f24")
        big = m(x.repeat(38, 1, 1, 1)[:300].contiguous(), precision="f24")
        y32 = m(x, precision="f32")
    c = cos_rows(y.cpu().numpy(), want)
    assert c.min() >= COS_BAR, c
    assert 1 - c.min() <= 5e-6, c                              # measured <= 1.5e-6 (profiles/r03_precision_probe.txt)
    assert torch.equal(y8, y)
    assert torch.equal(big, y.repeat(38, 1)[:300])
    assert cos_rows(y.cpu().numpy(), y32.cpu().numpy()).min() >= 0.99999        # (both sit ~1e-6 from the reference on the hardest set)
    if name == "mobilenetv1" and weights == "raw":
        with pytest.raises(RuntimeError, match="no f24 kernel"):
            product("sphere20")(x, precision="f24")


# measured on MI355X (tools/precision_probe.py): min cosine of the single-pass bf16 mode over the 8 golden inputs
_BF16_MEASURED = {("mobilenetv1", False): 0.99986, ("mobilenetv2", False): 0.99990, ("mobilenetv3_small", False): 0.99995,
                  ("mobilenetv3_large", False): 0.99995, ("mobilenetv1", True): 0.9961, ("mobilenetv2", True): 0.9973,
                  ("mobilenetv3_small", True): 0.974, ("mobilenetv3_large", True): 0.946}


@pytest.mark.parametrize("name,calibrated", sorted(_BF16_MEASURED))
def test_bf16_mode_against_the_bar_is_what_the_policy_says(name, calibrated, golden_dir, golden_x):
    """The explicit bf16 mode (the one BASELINE.json configs[1] names) against the same goldens.  It meets the bar on
    raw-init MobileNetV3 (the benchmarked configuration) and misses it on MobileNetV1/V2 and on every BN-calibrated
    MobileNet -- which is WHY "auto" resolves to f32 for the MobileNets.  The test pins both facts: where bf16 is
    claimed to pass it must pass, and elsewhere 1 - cos must stay within 1.5x of what was measured, so that a
    change in the bf16 kernels that shifts its accuracy is seen."""
    if calibrated:
        sd, want = _calibrated_state(name, golden_dir)
    else:
        sd, want = None, np.load(os.path.join(golden_dir, f"emb_{name}.npz"))["emb"]
    with torch.no_grad():
        y = product(name, sd)(golden_x.to(DEV), precision="bf16").cpu().numpy()
    c = cos_rows(y, want).min()
    if not calibrated and name.startswith("mobilenetv3"):
        assert c >= COS_BAR, c
    else:
        assert c < COS_BAR, f"{name}: bf16 now meets the bar ({c}); revisit auto_precision"
    assert 1 - c <= 1.5 * (1 - _BF16_MEASURED[(name, calibrated)]), c


@pytest.mark.parametrize("precision", ["bf16", "f32", "f24"])
@pytest.mark.parametrize("name", ["mobilenetv3_large", "mobilenetv3_small", "mobilenetv2", "mobilenetv1"])
def test_fused_depthwise_pointwise_equals_unfused(name, precision, golden_x):
    """dwp_kernel (depthwise + following 1x1 conv, depthwise result kept in LDS) rounds and accumulates
    exactly like the two separate kernels: fused and unfused plans must agree bit for bit -- in bf16 plans and in
    fp32 plans (depthwise result split into bf16 hi + lo planes in LDS, three MFMA passes)."""
    m = product(name)
    x = golden_x.to(DEV)
    with torch.no_grad():
        a = m(x, precision=precision, fuse=True)
        b = m(x, precision=precision, fuse=False)
    assert torch.equal(a, b)
    names = [m.plan(DEV, precision, True).profile(x)[i][1] for i in range(len(m.layer_names()))]
    assert "dwp_kernel" in names


@pytest.mark.parametrize("prec", ["bf16", "f32", "f24"])
@pytest.mark.parametrize("name", ["mobilenetv3_large", "mobilenetv3_small", "mobilenetv2"])
def test_fused_expand_depthwise_pointwise_equals_unfused(name, prec, golden_x, monkeypatch):
    """The expand 1x1 conv of an inverted-residual block is computed inside dwp_kernel as well (expand ->
    depthwise -> project in one launch): same MFMA k order and the same bf16 rounding points as the three
    separate kernels, so bit-identical -- to the unfused plan and to the plan with FR_FUSE3=0 (two-stage fusion)."""
    x = golden_x.to(DEV)
    m = product(name)
    with torch.no_grad():
        b = m(x, precision=prec, fuse=False)
        a = m(x, precision=prec, fuse=True)
        big = m(x.repeat(5, 1, 1, 1), precision=prec, fuse=True)
    names = [r[1] for r in m.plan(DEV, prec, True).profile(x)]
    assert torch.equal(a, b)
    assert torch.equal(big.view(5, len(x), -1), b.unsqueeze(0).expand(5, -1, -1))
    # (layer 0 is the stem conv, which has a fused form of its own: test_fused_stem_depthwise_pointwise_equals_unfused)
    fused = [i for i in range(3, len(names)) if names[i] in ("dwp_kernel", "dwc_kernel") and names[i - 1].startswith("(fused") and names[i - 2].startswith("(fused")]
    assert fused, names
    monkeypatch.setenv("FR_FUSE3", "0")
    m.invalidate()                                   # plans read FR_FUSE3 when they are created
    try:
        with torch.no_grad():
            c = m(x, precision=prec, fuse=True)
        names2 = [r[1] for r in m.plan(DEV, prec, True).profile(x)]
    finally:
        m.invalidate()
    assert torch.equal(c, b)
    if name != "mobilenetv3_small":                  # (its depthwise layers have 72+ channels: the two-stage form is not taken)
        assert "dwp_kernel" in names2
    assert not [i for i in range(3, len(names2)) if names2[i] in ("dwp_kernel", "dwc_kernel") and names2[i - 2].startswith("(fused")]


@pytest.mark.parametrize("prec", ["bf16", "f32"])
@pytest.mark.parametrize("name", ["mobilenetv3_large", "mobilenetv3_small", "mobilenetv2"])
def test_chunked_expand_depthwise_pointwise_is_bit_identical(name, prec, golden_x, monkeypatch):
    """dwc_kernel walks the expanded channels of an inverted-residual block in chunks of 32 (expand chunk -> depthwise chunk -> that
    chunk's two k-steps of the 1x1 conv into register accumulators): same k order and rounding points as dwp_kernel and as the three
    separate kernels.  Bit for bit: the default plan (chunked on its measured shapes) == FR_DWP_CHUNK=0 (dwp_kernel everywhere) ==
    FR_DWP_CHUNK=2 (chunked on every eligible shape with the cost model's tiles -- including the 72- and 88-channel blocks whose last
    chunk holds 1 or 3 of 4 channel vectors, and stride 2) == the unfused plan, at 8 images and at 131 (two-stream split, ragged)."""
    x = golden_x.to(DEV)
    m = product(name)
    big_x = x.repeat(17, 1, 1, 1)[:131]
    outs = {}
    for mode in ("1", "0", "2"):
        monkeypatch.setenv("FR_DWP_CHUNK", mode)
        m.invalidate()
        with torch.no_grad():
            outs[mode] = (m(x, precision=prec, fuse=True), m(big_x, precision=prec, fuse=True))
        kern = [r[1] for r in m.plan(DEV, prec, True).profile(x)]
        if mode == "0":
            assert "dwc_kernel" not in kern, kern
        elif mode == "2" or not (name == "mobilenetv2" and prec == "f32"):        # (mobilenetv2 f32: no measured win, not in the table)
            assert "dwc_kernel" in kern, kern
    with torch.no_grad():
        unfused = m(x, precision=prec, fuse=False)
    m.invalidate()
    for mode in ("1", "0", "2"):
        assert torch.equal(outs[mode][0], unfused), mode
        assert torch.equal(outs[mode][1], unfused.repeat(17, 1)[:131]), mode


class _StemBlockNet(fr.models.HipBackbone):
    """Not a reference model: a 3x3 stem conv, a depthwise 3x3 conv and a 1x1 conv (the pattern stem -> depthwise -> project that
    opens mobilenetv2 and mobilenetv3_large), with the channel count, the depthwise stride, the activations and the residual
    (the stem output added to the 1x1 output, as mobilenetv3_large's first block does) as parameters, then a Linear."""

    def __init__(self, C, cout, stride, stem_act, residual):
        super().__init__()
        U = fr.models.ConvUnit
        self.residual = residual
        self.c0 = U(3, C, 3, 1, act=stem_act)
        self.d1 = U(C, C, 3, stride, groups=C, act="prelu")
        self.p1 = U(C, cout, 1, act=None)
        self.p2 = U(cout, 8, 1)
        hw = 112 // stride
        self.fc = torch.nn.Linear(8 * hw * hw, 32)
        g = torch.Generator().manual_seed(7 * C + stride)
        for q in self.parameters():
            with torch.no_grad():
                q.copy_(torch.randn(q.shape, generator=g) * (0.15 if q.dim() > 1 else 0.2))
        with torch.no_grad():
            self.fc.weight.mul_(0.02)
        for m in self.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                with torch.no_grad():
                    m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                    m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                    m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)

    def lower(self, g):
        a0 = self.c0.lower(g, "c0", -1)
        a1 = self.d1.lower(g, "d1", a0)
        a2 = self.p1.lower(g, "p1", a1, a0 if self.residual else -1)
        a3 = self.p2.lower(g, "p2", a2)
        g.add("fc", op=fr._lib.FR_OP_LINEAR, src=a3, cin=self.fc.in_features, cout=self.fc.out_features,
              weight=self.fc.weight, bias=self.fc.bias)

    def reference(self, x):
        y0 = self.c0(x)
        p1 = self.p1(self.d1(y0))
        if self.residual:
            p1 = p1 + y0
        return p1, self.fc(self.p2(p1).flatten(1))


@pytest.mark.parametrize("precision,bar", [("bf16", 2e-2), ("f32", 2e-5)])
@pytest.mark.parametrize("C,cout,stride,stem_act,residual", [(32, 16, 1, "prelu", False), (16, 16, 1, "hswish", True),
                                                            (32, 64, 2, "hswish", False), (24, 24, 1, None, True)])
def test_fused_stem_depthwise_pointwise_equals_unfused(C, cout, stride, stem_act, residual, precision, bar, monkeypatch):
    """dwp_kernel's stem form (3x3 stem conv on the fp32 NCHW input -> depthwise -> 1x1 in one launch; the stem output, which is
    also the residual of mobilenetv3_large's first block, stays in LDS): same im2col order, MFMA sequence and rounding points as
    stem_kernel followed by the depthwise + 1x1 kernel, so bit for bit equal to the plan with FR_FUSE_STEM=0 and to the unfused
    plan -- 1x1 output and embedding, 5 images and 131 (both halves of the two-stream split), stride 1 (7 x 7 whole tiles) and
    stride 2 (56 x 56 output: ragged last tile row and column), 16 / 24 / 32 channels, residual from LDS or none -- and close to
    a plain PyTorch fp32 forward on the CPU (relative L2 <= 2e-2 in bf16, 2e-5 in f32).  A tap on the stem and a uint8 input
    take the separate stem kernel from the same plan."""
    m = _StemBlockNet(C, cout, stride, stem_act, residual).eval()
    torch.manual_seed(C + stride)
    x = torch.rand(5, 3, 112, 112) * 2 - 1
    with torch.no_grad():
        want_p1, want_emb = m.reference(x)
    m = m.to(DEV)
    xg = x.to(DEV)
    names = m.layer_names()
    ip1 = names.index("p1")
    got = {}
    for sw in ("1", "0"):
        monkeypatch.setenv("FR_FUSE_STEM", sw)
        m.invalidate()                                   # plans read FR_FUSE_STEM when they are created
        kern = [r[1] for r in m.plan(DEV, precision, True).profile(xg)]
        assert kern[ip1] == "dwp_kernel", kern
        # (plans with fewer than 32 stem channels keep the stem kernel: measured slower fused, plan.hip)
        assert kern[0].startswith("(fused") == (sw == "1" and C >= 32), kern
        with torch.no_grad():
            emb, p1 = m.plan(DEV, precision, True).run(xg, tap=ip1)     # (forward_tap takes the unfused plan)
            big = m(xg.repeat(27, 1, 1, 1)[:131], precision=precision)
            emb0, c0 = m.plan(DEV, precision, True).run(xg, tap=0)      # tap on the stem: the separate stem kernel, same plan
            assert torch.equal(emb0, emb)
        got[sw] = (emb, p1, big, c0)
    with torch.no_grad():
        unfused = m(xg, precision=precision, fuse=False)
    m.invalidate()
    for a, b in zip(got["1"], got["0"]):
        assert torch.equal(a, b)
    assert torch.equal(got["1"][0], unfused)
    assert torch.equal(got["1"][2], got["1"][0].repeat(27, 1)[:131])
    for g_, w_ in ((got["1"][1].cpu(), want_p1), (got["1"][0].cpu(), want_emb)):
        assert g_.shape == w_.shape and bool(torch.isfinite(g_).all())
        assert float((g_ - w_).norm() / w_.norm()) <= bar


@pytest.mark.parametrize("name,batch", [("sphere20", 8), ("sphere36", 131), ("sphere20", 300)])
def test_shared_pixel_conv_equals_per_tap_conv(name, batch, golden_x, monkeypatch):
    """conv_halo_kernel (one staged pixel range shared by the nine taps of a stride-1 3x3 conv) against
    conv_glds_kernel with the same 32-channel K slices (FR_CONV_HALO=0; FR_CONV_BK64_MIN above every layer): same
    (chunk, tap, k-step) accumulation order, so bit for bit -- at 8 images (128-pixel blocks everywhere), at 131
    (256-pixel blocks on the large maps, 128 on the 14x14 ones, ragged last tile, both sub-batches of the two-stream
    split) and at 300 images.  Borders, row wrap and image boundaries inside a staged range are all in play: the
    golden faces are 8 different images repeated, and every row must equal the first occurrence of its image."""
    x8 = golden_x.to(DEV)
    x = x8.repeat((batch + 7) // 8, 1, 1, 1)[:batch].contiguous()
    m = product(name)
    monkeypatch.setenv("FR_CONV_BK64_MIN", "1000000")   # the layers that stay on conv_glds_kernel: 32-channel slices in both plans
    m.invalidate()                                   # plans read both variables when they are created
    with torch.no_grad():
        a = m(x, precision="bf16")
    names = [r[1] for r in m.plan(DEV, "bf16", True).profile(x8)]
    assert "conv_halo16_kernel" in names and "conv_glds_kernel" in names, names
    monkeypatch.setenv("FR_CONV_HALO", "0")
    m.invalidate()
    try:
        with torch.no_grad():
            b = m(x, precision="bf16")
        names2 = [r[1] for r in m.plan(DEV, "bf16", True).profile(x8)]
    finally:
        m.invalidate()
    assert not [n for n in names2 if n.startswith("conv_halo")]
    assert torch.equal(a, b)
    assert torch.equal(a, a[:8].repeat((batch + 7) // 8, 1)[:batch])


@pytest.mark.parametrize("name,batch", [("sphere20", 131), ("sphere36", 9)])
def test_conv_slices_per_barrier_are_bit_identical(name, batch, golden_x, monkeypatch):
    """conv_halo16_kernel with 1, 2 (default) and 3 K slices per barrier (FR_CONV_SPB, read when a plan is created): the
    weight ring only changes how many slices one DMA + barrier covers, not the accumulation order, so the embeddings are
    bit-identical -- on 131 images (256-pixel blocks, both sub-batches, ragged tiles) and on 9 (128-pixel blocks).  With
    2 slices per barrier a stage straddles the 9-tap chunk boundary on every other chunk: the frame-staging schedule
    (next frame only while all slices of an iteration read the current one) is what this pins."""
    x8 = golden_x.to(DEV)
    x = x8.repeat((batch + 7) // 8, 1, 1, 1)[:batch].contiguous()
    m = product(name)
    outs = {}
    try:
        for spb in ("1", "2", "3"):
            monkeypatch.setenv("FR_CONV_SPB", spb)
            m.invalidate()
            with torch.no_grad():
                outs[spb] = m(x, precision="bf16").clone()
            assert "conv_halo16_kernel" in [r[1] for r in m.plan(DEV, "bf16", True).profile(x8)]
    finally:
        monkeypatch.delenv("FR_CONV_SPB")
        m.invalidate()
    assert torch.equal(outs["1"], outs["2"]) and torch.equal(outs["1"], outs["3"])
    assert torch.equal(outs["2"], outs["2"][:8].repeat((batch + 7) // 8, 1)[:batch])


@pytest.mark.parametrize("precision", ["bf16", "f32"])
def test_split_k_linear_against_whole_k_and_batch_invariance(precision, golden_x, monkeypatch):
    """The SphereFace fc (25088 -> 512) on fc_splitk_kernel (8 K slices + ordered reduce, the default for K >= 8192)
    against fc_kernel (FR_FC_SPLITK=0, read when a plan is created): same products, another summation order, so equal
    to fp32 round-off (|d| <= 2e-5 of the embedding norm), and the embedding cosine 1 to 1e-6.  The slice count
    depends on K only: an image's embedding is bit-identical at batch 1, 8, 70 and 300 (one and two sub-batches)."""
    x8 = golden_x.to(DEV)
    m = product("sphere20")
    with torch.no_grad():
        a = m(x8, precision=precision).clone()
        names = [r[1] for r in m.plan(DEV, precision, True).profile(x8)]
        assert names[-1] == "fc_splitk_kernel(+reduce)", names[-1]
        for batch in (1, 70, 300):
            xb = x8.repeat((batch + 7) // 8, 1, 1, 1)[:batch].contiguous()
            assert torch.equal(m(xb, precision=precision), a.repeat((batch + 7) // 8, 1)[:batch]), batch
    monkeypatch.setenv("FR_FC_SPLITK", "0")
    m.invalidate()
    try:
        with torch.no_grad():
            b = m(x8, precision=precision).clone()
        assert [r[1] for r in m.plan(DEV, precision, True).profile(x8)][-1] == "fc_kernel"
    finally:
        monkeypatch.delenv("FR_FC_SPLITK")
        m.invalidate()
    assert float((a - b).norm(dim=1).max() / b.norm(dim=1).min()) <= 2e-5
    cos = torch.nn.functional.cosine_similarity(a.double(), b.double())
    assert float(cos.min()) >= 1 - 1e-6


@pytest.mark.parametrize("name", ["mobilenetv3_small", "mobilenetv3_large"])
def test_squeeze_excite_scales_against_oracle(name, golden_dir, golden_x):
    """The channel scale of every squeeze-excite block (pooling folded from the depthwise kernel's per-tile sums, fc1 + ReLU,
    fc2 + hard sigmoid: utils/layers.py:114-118) against the oracle's hardsigmoid(fc2(relu(fc1(mean)))) -- the per-layer
    test above compares SE blocks only through their consumer.  On the trained-like weights, where the scales are spread
    over [0, 1] instead of sitting at 0.5.  The deep blocks of this weight set saturate (pre-activations of tens), so the
    error of their INPUT shows: measured <= 4e-5 on the first blocks, 2.5e-3 on the worst; the bar is 1e-2 absolute and
    1e-4 on the first three.  An image's scales are bit-identical at batch 8 and 70."""
    from test_oracle_golden import trained_like_state
    sd, _ = trained_like_state(name, golden_dir)
    m = product(name, sd)
    oracle = ob.build(name)
    oracle.load_state_dict(sd)
    oracle.eval()
    x8 = golden_x.to(DEV)
    names = m.layer_names()
    mods = dict(oracle.named_modules())
    se_layers = [(i, n) for i, n in enumerate(names) if isinstance(mods.get(n), ob.SqueezeExcite)]
    assert len(se_layers) >= 8
    want = {}
    hooks = [mods[n].fc2.register_forward_hook(lambda mod, a, out, n=n: want.__setitem__(n, torch.nn.functional.hardsigmoid(out).flatten(1)))
             for _, n in se_layers]
    with torch.no_grad():
        oracle(golden_x)
    for hk in hooks:
        hk.remove()
    x70 = x8.repeat(9, 1, 1, 1)[:70].contiguous()
    for k, (i, n) in enumerate(se_layers):
        _, t = m.forward_tap(x8, i, "f32")
        got = t.reshape(-1)[:want[n].numel()].reshape(8, -1)          # the tap of an SE block is its scale [N, C] (plan.hip, embed_impl)
        err = float((got.cpu() - want[n]).abs().max())
        assert err <= (1e-4 if k < 3 else 1e-2), (n, err)
        _, t70 = m.forward_tap(x70, i, "f32")
        assert torch.equal(t70.reshape(-1)[:70 * got.shape[1]].reshape(70, -1), got.repeat(9, 1)[:70]), n
    assert max(float(w.max() - w.min()) for w in want.values()) > 0.5            # the scales are not all ~0.5


class _OddConvNet(fr.models.HipBackbone):
    """Not a reference model: dense 3x3 convs with channel counts the SphereFace nets never use (cout tiles that are
    partly empty, cin not a multiple of 64), stride-1 layers with and without residual on 56x56 and 28x28 maps."""

    def __init__(self):
        super().__init__()
        U = fr.models.ConvUnit
        self.c0 = U(3, 32, 3, 2, norm=False)            # 56x56
        self.c1 = U(32, 96, 3, 1, norm=False, bias=False)
        self.c2 = U(96, 96, 3, 1, norm=False)           # + residual c1
        self.c3 = U(96, 160, 3, 2, norm=False)          # 28x28
        self.c4 = U(160, 160, 3, 1, norm=True, bias=False)   # + residual c3, folded BatchNorm
        self.c5 = U(160, 40, 3, 1, norm=False)
        self.fc = torch.nn.Linear(40 * 28 * 28, 64)
        g = torch.Generator().manual_seed(5)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * (0.05 if p.dim() > 1 else 0.2))
        bn = self.c4[1]
        with torch.no_grad():
            bn.running_mean.copy_(torch.randn(160, generator=g) * 0.1)
            bn.running_var.copy_(torch.rand(160, generator=g) + 0.5)

    def lower(self, g):
        a0 = self.c0.lower(g, "c0", -1)
        a1 = self.c1.lower(g, "c1", a0)
        a2 = self.c2.lower(g, "c2", a1, a1)
        a3 = self.c3.lower(g, "c3", a2)
        a4 = self.c4.lower(g, "c4", a3, a3)
        a5 = self.c5.lower(g, "c5", a4)
        g.add("fc", op=fr._lib.FR_OP_LINEAR, src=a5, cin=self.fc.in_features, cout=self.fc.out_features,
              weight=self.fc.weight, bias=self.fc.bias)

    def reference(self, x):
        """Plain PyTorch fp32 of the same graph (eval-mode BatchNorm), outputs of c0..c5."""
        outs = []
        y0 = self.c0(x); outs.append(y0)
        y1 = self.c1(y0); outs.append(y1)
        y2 = self.c2(y1) + y1; outs.append(y2)
        y3 = self.c3(y2); outs.append(y3)
        y4 = self.c4(y3) + y3; outs.append(y4)
        y5 = self.c5(y4); outs.append(y5)
        return outs


@pytest.mark.parametrize("batch", [3, 70])
def test_dense_conv_odd_shapes_against_torch_fp32(batch, monkeypatch):
    """conv_halo_kernel / conv_glds_kernel on shapes outside the SphereFace nets: cout 96 and 40 (64-cout tiles, the last
    one partly empty), cout 160 (128 + 32), cin 96 / 160 (32-channel K slices only), residuals, folded BatchNorm.
    Every conv output (fr_embed_tap, fp32 NCHW) against a plain PyTorch fp32 forward of the same graph on the CPU:
    relative L2 error <= 2e-2 per layer (bf16 activations and weights, six layers deep; measured ~4e-3), and the
    shared-pixel kernel bit for bit against the per-tap kernel."""
    m = _OddConvNet().eval()
    torch.manual_seed(batch)
    x = torch.rand(batch, 3, 112, 112) * 2 - 1
    with torch.no_grad():
        want = m.reference(x)
    m = m.to(DEV)
    xg = x.to(DEV)
    monkeypatch.setenv("FR_CONV_BK64_MIN", "1000000")
    taps = {}
    for halo in ("1", "0"):
        monkeypatch.setenv("FR_CONV_HALO", halo)
        m.invalidate()
        names = [r[1] for r in m.plan(DEV, "bf16", False).profile(xg)]
        assert ("conv_halo16_kernel" in names) == (halo == "1"), names
        with torch.no_grad():
            taps[halo] = [m.forward_tap(xg, i, precision="bf16")[1] for i in range(6)]
    m.invalidate()
    for i in range(6):
        got = taps["1"][i].cpu()
        assert got.shape == want[i].shape
        rel = float((got - want[i]).norm() / want[i].norm())
        assert rel <= 2e-2, (i, rel)
        assert torch.equal(taps["1"][i], taps["0"][i]), i


class _OddPointwiseNet(fr.models.HipBackbone):
    """Not a reference model: 1x1 convs with tile counts the MobileNets never produce -- 224 couts behind K = 272 (seven
    32-cout tiles: column groups of 4 + 3 in pw_proj_kernel), 24 couts behind K = 224 (ONE tile: the kernel's second tile
    multiplies LDS it never filled), a squeeze-excite block in front of a projection with a residual, a 40-cout tail."""

    def __init__(self):
        super().__init__()
        U = fr.models.ConvUnit
        self.c0 = U(3, 16, 3, 2, act="hswish")                       # 56x56
        self.e1 = U(16, 272, 1, act="hswish")
        self.d1 = U(272, 272, 3, 2, groups=272, act="hswish")        # 28x28
        self.p1 = U(272, 224, 1, act=None)                           # K 272 -> 7 tiles
        self.d2 = U(224, 224, 5, 1, groups=224)
        self.se = fr.models.SEUnit(224, 56)
        self.p2 = U(224, 224, 1, act=None)                           # SE scale + residual p1, 7 tiles
        self.p3 = U(224, 24, 1, act=None)                            # one tile
        self.p4 = U(24, 40, 1)
        self.fc = torch.nn.Linear(40 * 28 * 28, 64)
        g = torch.Generator().manual_seed(11)
        for p in self.parameters():
            with torch.no_grad():
                p.copy_(torch.randn(p.shape, generator=g) * (0.08 if p.dim() > 1 else 0.2))
        for m in self.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                with torch.no_grad():
                    m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                    m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                    m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)

    def lower(self, g):
        a = self.c0.lower(g, "c0", -1)
        a = self.e1.lower(g, "e1", a)
        a = self.d1.lower(g, "d1", a)
        p1 = self.p1.lower(g, "p1", a)
        a = self.d2.lower(g, "d2", p1)
        a = self.se.lower(g, "se", a)
        p2 = self.p2.lower(g, "p2", a, p1)
        a = self.p3.lower(g, "p3", p2)
        a = self.p4.lower(g, "p4", a)
        g.add("fc", op=fr._lib.FR_OP_LINEAR, src=a, cin=self.fc.in_features, cout=self.fc.out_features,
              weight=self.fc.weight, bias=self.fc.bias)

    def reference(self, x):
        """Plain PyTorch fp32 of the same graph: outputs of p1, p2, p3, p4 and the embedding."""
        F = torch.nn.functional
        y = self.d1(self.e1(self.c0(x)))
        p1 = self.p1(y)
        d2 = self.d2(p1)
        sc = F.hardsigmoid(self.se.fc2(F.relu(self.se.fc1(d2.mean((2, 3), keepdim=True)))))
        p2 = self.p2(d2 * sc) + p1
        p3 = self.p3(p2)
        p4 = self.p4(p3)
        return [p1, p2, p3, p4, self.fc(p4.flatten(1))]


@pytest.mark.parametrize("precision,bar", [("bf16", 3e-2), ("f32", 2e-5)])
def test_pointwise_odd_tile_counts_against_torch_fp32(precision, bar):
    """pw_proj_kernel's compile-time tile counts on shapes outside the MobileNets (see _OddPointwiseNet), 70 images (two
    128-pixel super-tiles per image row group, a ragged last one): the 1x1 outputs and the embedding against a plain
    PyTorch fp32 forward on the CPU, relative L2 error per tensor <= 3e-2 in bf16 and <= 2e-5 in f32 (f24 plans have no
    Linear on a conv output, and share the kernel template); no NaN from the tiles a block multiplies without having filled them."""
    m = _OddPointwiseNet().eval()
    torch.manual_seed(3)
    x = torch.rand(70, 3, 112, 112) * 2 - 1
    with torch.no_grad():
        want = m.reference(x)
    m = m.to(DEV)
    xg = x.to(DEV)
    names = m.layer_names()
    kern = dict((r[0], r[1]) for r in m.plan(DEV, precision, False).profile(xg[:8]))
    assert kern["p1"] == "pw_proj_kernel", kern                  # K = 272: the large-K kernel in every mode (7 tiles: groups of 4 + 3)
    if precision != "bf16":                                      # K = 224: large-K kernel in fp32 plans (more than 8 k-steps) only
        assert kern["p2"] == "pw_proj_kernel" and kern["p3"] == "pw_proj_kernel", kern
    with torch.no_grad():
        got = [m.forward_tap(xg, names.index(n), precision=precision)[1].cpu() for n in ("p1", "p2", "p3", "p4")]
        got.append(m(xg, precision=precision).cpu())
    m.invalidate()
    for i, (g, w) in enumerate(zip(got, want)):
        assert g.shape == w.shape and bool(torch.isfinite(g).all()), i
        rel = float((g - w).norm() / w.norm())
        assert rel <= bar, (i, rel)


@pytest.mark.parametrize("name,precision", [("mobilenetv3_small", "bf16"), ("mobilenetv3_small", "f32"), ("mobilenetv1", "bf16"),
                                            ("sphere20", "bf16"), ("sphere20", "f32")])
def test_uint8_input_equals_transformed_input(name, precision, golden_inputs, golden_x):
    """fr_embed_u8 (uint8 HWC faces, ToTensor + Normalize(0.5, 0.5) inside the stem kernel's load; SURVEY 8(f)
    rank 2) against fr_embed on the oracle's transform of the same images (inference.py:98-106, pinned by
    test_host.py::test_transform_matches_oracle): bit for bit, for the stride-1 (MobileNet) and the stride-2
    (SphereFace) stem, in both precisions, and across the two-stream split with a ragged batch."""
    import numpy as np
    m = product(name)
    u8 = torch.from_numpy(np.ascontiguousarray(golden_inputs)).to(DEV)
    assert u8.dtype == torch.uint8 and tuple(u8.shape[1:]) == (112, 112, 3)
    rng = np.random.default_rng(5)
    noise = torch.from_numpy(rng.integers(0, 256, size=(3, 112, 112, 3), dtype=np.uint8)).to(DEV)   # every byte value, hard edges
    u8 = torch.cat([u8, noise])
    x = torch.stack([osc.to_model_input(im) for im in u8.cpu().numpy()]).to(DEV)
    with torch.no_grad():
        a = m.forward_u8(u8, precision=precision)
        b = m(x, precision=precision)
        big = m.forward_u8(u8.repeat(19, 1, 1, 1)[:131], precision=precision)       # 131 images: split 65 + 66
    assert torch.equal(a, b)
    assert torch.equal(big, b.repeat(19, 1)[:131])
    with pytest.raises(ValueError):
        m.forward_u8(u8[:, :, :111], precision=precision)
    assert m.forward_u8(u8[:0], precision=precision).shape == (0, 512)


def test_batch_sizes_and_determinism(golden_x):
    m = product("mobilenetv3_small")
    x = golden_x.to(DEV)
    with torch.no_grad():
        full = m(x, precision="f32")
        one = torch.cat([m(x[i:i + 1], precision="f32") for i in range(len(x))])
        again = m(x, precision="f32")
        big = m(x.repeat(33, 1, 1, 1), precision="f32")      # 264 images: crosses tile boundaries
        empty = m(x[:0], precision="f32")
        # 259 images: fr_embed cuts batches >= 128 into two sub-batches (129 + 130) on two streams
        odd = m(x.repeat(33, 1, 1, 1)[:259], precision="bf16")
        ref16 = m(x, precision="bf16")
    assert torch.equal(full, again)
    assert torch.equal(full, one)
    assert torch.equal(big.view(33, 8, -1), full.unsqueeze(0).expand(33, -1, -1))
    assert torch.equal(odd[:256].view(32, 8, -1), ref16.unsqueeze(0).expand(32, -1, -1)) and torch.equal(odd[256:], ref16[:3])
    assert empty.shape == (0, 512)


def test_pair_scoring_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "pairs.npz"))
    a, b = torch.from_numpy(g["a"]).to(DEV), torch.from_numpy(g["b"]).to(DEV)
    s0, d0 = fr.cosine_pairs(a, b, eps=0.0, threshold=0.35)
    s1, d1 = fr.cosine_pairs(a, b, eps=1e-5, threshold=0.35)
    assert np.abs(s0.cpu().numpy() - g["sim_np"]).max() <= 1e-3
    assert np.abs(s1.cpu().numpy() - g["sim_lfw"]).max() <= 1e-3
    # decisions: identical wherever the reference similarity is not within fp32 rounding of 0.35
    far = np.abs(g["sim_lfw"] - 0.35) > 2e-6
    assert np.array_equal(d1.cpu().numpy()[far].astype(int), g["dec_lfw"][far])
    far0 = np.abs(g["sim_np"].astype(np.float64) - 0.35) > 2e-6
    assert np.array_equal(d0.cpu().numpy()[far0], g["dec_np"][far0])
    assert np.float32 == type(fr.compute_similarity(g["a"][0], g["b"][0]))
    assert fr.cosine_pairs(a[:0], b[:0])[0].numel() == 0


def _lfw_fixture(tmp_path, n_ids=40, per_id=3, n_pairs=120, seed=3):
    faces = osc.synthetic_faces(n_ids, per_id, seed=1234)
    rng = np.random.default_rng(seed)
    images, lines = {}, ["header line is dropped\n"]
    root = str(tmp_path)
    for i in range(n_ids):
        for j in range(per_id):
            images[os.path.join(root, f"id{i}", f"id{i}_{j + 1:04d}.jpg")] = faces[i, j]
    for p in range(n_pairs):
        i = int(rng.integers(n_ids))
        if p % 2 == 0:
            a, b = rng.choice(per_id, 2, replace=False) + 1
            lines.append(f"id{i} {a} {b}\n")
        else:
            k = int((i + 1 + rng.integers(n_ids - 1)) % n_ids)
            lines.append(f"id{i} {int(rng.integers(per_id)) + 1} id{k} {int(rng.integers(per_id)) + 1}\n")
    lines.append("id0 1 id1 99\n")          # missing image -> pair skipped (evaluate.py:547-549)
    lines.append("malformed\n")
    with open(os.path.join(root, "lfw_ann.txt"), "w") as f:
        f.writelines(lines)
    return root, images, lines[1:]


@pytest.mark.parametrize("calibrated,precision", [(False, None), (True, None), (False, "bf16")])
def test_lfw_pair_verification_config1(tmp_path, golden_dir, calibrated, precision):
    """BASELINE config 1 (reduced pair count so the CPU oracle's batch-1 loop stays short):
    mobilenetv3_small, LFW-layout annotation, eval() triple vs the oracle's restated pair loop."""
    root, images, pair_lines = _lfw_fixture(tmp_path)
    onet = ob.seeded("mobilenetv3_small", 0)
    if calibrated:
        g = np.load(os.path.join(golden_dir, "bncal_mobilenetv3_small.npz"))
        sd = onet.state_dict()
        for i, k in enumerate(g["keys"]):
            sd[str(k)] = torch.from_numpy(g[f"s{i}"])
        onet.load_state_dict(sd)
    model = product("mobilenetv3_small", onet.state_dict())
    model.precision = precision          # None: the product default ("auto" -> f32 for MobileNets), what load_model() -> eval() runs
    assert model.plan(DEV).precision == (precision or "f32")

    def loader(path):
        if path not in images:
            raise FileNotFoundError(path)
        return images[path]

    mean, preds, metrics = fr.evaluate.eval(model, device=DEV, val_root=root, compute_full_metrics=True,
                                            image_loader=loader)
    omean, opreds, _ = osc.eval_pairs(onet, images, pair_lines, root)
    assert preds.shape == opreds.shape == (120, 4)
    assert list(preds[:, 0]) == list(opreds[:, 0]) and list(preds[:, 3]) == list(opreds[:, 3])
    s, o = preds[:, 2].astype(float), opreds[:, 2].astype(float)
    assert np.abs(s - o).max() <= 1e-3
    far = np.abs(o - 0.35) > 1e-3
    assert np.array_equal(osc.decide(s)[far], osc.decide(o)[far])
    assert abs(mean - omean) <= 1e-3 and abs(metrics["mean_similarity"] - omean) <= 1e-3
    if calibrated:
        assert osc.decide(o).sum() > 0                    # non-trivial decisions
    else:
        assert np.all(osc.decide(s) == 0)                 # raw init + 1e-5 eps: all "different" 


def test_synthetic_rows_bit_exact():
    G, npl = 5000, 16
    got = fr.search.synth_rows(7, 1000, 700, DEV, plant_seed=11, n_plant=npl, n_gallery_total=G)
    want = osr.gallery_rows(7, 11, 1000, 700, G, npl)
    assert np.array_equal(got.view(torch.int16).cpu().numpy().view(np.uint16), want)
    q = fr.search.synth_rows(11, 0, 33, DEV)
    assert np.array_equal(q.view(torch.int16).cpu().numpy().view(np.uint16), osr.query_rows(11, 0, 33))


@pytest.mark.parametrize("Q,G", [(64, 4096), (130, 1000), (1, 129), (257, 20011), (511, 63), (512, 64), (16384 + 130, 700),
                                 (32768 + 130, 700)])   # last: two query chunks of the resident-query kernel (the one before: of search_kernel)
def test_topk_search_parity(Q, G):
    npl = min(8, Q)
    g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=npl, n_gallery_total=G)
    q = fr.search.synth_rows(11, 0, Q, DEV)
    s, i = fr.search.Matcher(g, k=5).search(q)
    gf = osr.bf16_bits_to_f32(g.view(torch.int16).cpu().numpy().view(np.uint16))
    qf = osr.bf16_bits_to_f32(q.view(torch.int16).cpu().numpy().view(np.uint16))
    os_, oi, gap = osr.topk_cosine(qf, gf, 5)
    s, i = s.cpu().numpy(), i.cpu().numpy()
    assert np.abs(s - os_).max() <= 1e-3
    # exact index equality rank by rank wherever neighbouring scores are separated by > 2e-4
    # (5x tighter than the 1e-3 similarity bar).  The oracle re-normalises the bf16-rounded rows in
    # float64 while the kernel takes them as unit vectors (their norms are off by ~1e-4 relative), so
    # scores closer than that can legitimately swap; fp32 accumulation adds ~1e-5.
    full = osr.topk_cosine(qf, gf, 6)[0] if G > 5 else None
    for r in range(Q):
        seps = np.abs(np.diff(np.concatenate([os_[r], [os_[r, -1] - gap[r]]])))
        ok = seps > 2e-4
        clear = np.concatenate([[True], ok[:-1]]) & ok
        assert np.array_equal(i[r][clear], oi[r][clear]), (r, i[r], oi[r])
    for t in range(npl):
        assert i[t, 0] == osr.plant_row(t, G, npl)
    assert np.all(np.diff(s, axis=1) <= 0)


def test_resident_query_search_equals_slice_staged_search(monkeypatch):
    """The three forms of the search kernel -- search_rq16_kernel (queries in registers, 16x16x32 MFMA; the default for
    k <= 5), search_rq_kernel (the same on 32x32x16 MFMA; FR_SEARCH_MFMA=32, and always for k = 6..8) and search_kernel
    (slice-staged, FR_SEARCH_RQ=0) -- accumulate a score over K = 512 in the same order, so scores and indices must be
    identical bit for bit: ragged query count over two 256-query blocks, a gallery that ends inside a tile, planted
    near-duplicates, k = 1, 5 and 8."""
    G, Q = 50021, 300
    g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=16, n_gallery_total=G)
    q = fr.search.synth_rows(11, 0, Q, DEV)
    for k in (1, 5, 8):
        monkeypatch.setenv("FR_SEARCH_RQ", "1")
        monkeypatch.setenv("FR_SEARCH_MFMA", "16")
        s16, i16 = fr.search.Matcher(g, k=k).search(q)
        monkeypatch.setenv("FR_SEARCH_MFMA", "32")
        s1, i1 = fr.search.Matcher(g, k=k).search(q)
        monkeypatch.setenv("FR_SEARCH_RQ", "0")
        s0, i0 = fr.search.Matcher(g, k=k).search(q)
        assert torch.equal(i1, i0) and torch.equal(s1, s0)
        assert torch.equal(i16, i0) and torch.equal(s16, s0)


def test_topk_search_edge_cases():
    """Empty gallery, a gallery smaller than k (missing ranks are (-inf, -1)), identical rows (ties go to the
    lower index, as oracle/search.py:topk_cosine orders them), k = 1 and k = 8, and a k the kernel refuses."""
    q = fr.search.synth_rows(11, 0, 70, DEV)
    s, i = fr.search.Matcher(q[:0], k=5).search(q)
    assert s.shape == (70, 5) and torch.isinf(s).all() and (s < 0).all() and (i == -1).all()
    g3 = fr.search.synth_rows(7, 0, 3, DEV)
    s, i = fr.search.Matcher(g3, k=5).search(q)
    assert (i[:, 3:] == -1).all() and torch.isinf(s[:, 3:]).all() and (i[:, :3] >= 0).all()
    assert torch.equal(i[:, :3].sort(dim=1).values, torch.arange(3, device=DEV).expand(70, 3))
    dup = q[:1].repeat(300, 1)                                            # 300 identical rows: pure tie-break
    s, i = fr.search.Matcher(dup, k=5).search(q[:4])
    assert torch.equal(i, torch.arange(5, device=DEV).expand(4, 5))
    g = fr.search.synth_rows(7, 0, 5000, DEV)
    gf = osr.bf16_bits_to_f32(g.view(torch.int16).cpu().numpy().view(np.uint16))
    qf = osr.bf16_bits_to_f32(q.view(torch.int16).cpu().numpy().view(np.uint16))
    for k in (1, 8):
        s, i = fr.search.Matcher(g, k=k).search(q)
        os_, oi, _ = osr.topk_cosine(qf, gf, k)
        assert s.shape == (70, k) and np.abs(s.cpu().numpy() - os_).max() <= 1e-3
        assert np.array_equal(i[:, 0].cpu().numpy(), oi[:, 0])
    with pytest.raises(RuntimeError):
        fr.search.Matcher(g, k=9).search(q)
    s, i = fr.search.Matcher(g, idx_offset=1 << 33, k=5).search(q[:2])     # indices beyond 32 bits (100 M-row shards)
    assert (i >= (1 << 33)).all()


def test_sharded_search_equals_single(golden_x):
    """Fake-shard mode: P shards on one device, merged on the GPU, must equal
    the single-shard search exactly (deterministic tie rule)."""
    G, Q = 30000, 96
    g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=8, n_gallery_total=G)
    q = fr.search.synth_rows(11, 0, Q, DEV)
    s1, i1 = fr.search.Matcher(g, k=5).search(q)
    for P in (2, 3, 8):
        parts = []
        for r in range(P):
            lo, hi = fr.search.shard_range(G, r, P)
            local = fr.search.synth_rows(7, lo, hi - lo, DEV, plant_seed=11, n_plant=8, n_gallery_total=G)
            parts.append(fr.search.Matcher(local, idx_offset=lo, k=5).search(q))
        ms, mi = fr.search.merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), 5)
        assert torch.equal(mi, i1) and torch.equal(ms, s1)


def test_search_at_scale_by_properties():
    """A gallery far beyond what the CPU oracle scores in seconds (2 M rows x 20 000 queries = two query chunks, many
    gallery splits), checked through properties that do not depend on size: every planted query finds its planted
    row first with a score far above the rest, scores are sorted, indices are valid and distinct, 4 shards merged == one shard bit
    for bit, a sub-range of the queries gives the same rows, and a 4 k-row slice agrees with the oracle."""
    G, Q, npl = 2_000_000, 20_000, 64
    g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=npl, n_gallery_total=G)
    q = fr.search.synth_rows(11, 0, Q, DEV)
    s, i = fr.search.Matcher(g, k=5).search(q)
    planted = torch.tensor([osr.plant_row(t, G, npl) for t in range(npl)], device=DEV)
    assert torch.equal(i[:npl, 0], planted) and (s[:npl, 0] > 0.9).all()          # planted rows are noisy copies of the queries
    assert (s[:, 1:] <= s[:, :-1]).all() and (i >= 0).all() and (i < G).all()
    assert (i.sort(dim=1).values.diff(dim=1) > 0).all()                       # five distinct rows per query
    parts = []
    for r in range(4):
        lo, hi = fr.search.shard_range(G, r, 4)
        parts.append(fr.search.Matcher(g[lo:hi], idx_offset=lo, k=5).search(q))
    ms, mi = fr.search.merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), 5)
    assert torch.equal(mi, i) and torch.equal(ms, s)
    s2, i2 = fr.search.Matcher(g, k=5).search(q[16000:17000])                  # other launch geometry, same answers
    assert torch.equal(i2, i[16000:17000]) and torch.equal(s2, s[16000:17000])
    lo = 1_000_000
    gf = osr.bf16_bits_to_f32(g[lo:lo + 4096].view(torch.int16).cpu().numpy().view(np.uint16))
    qf = osr.bf16_bits_to_f32(q[:32].view(torch.int16).cpu().numpy().view(np.uint16))
    os_, oi, _ = osr.topk_cosine(qf, gf, 5)
    ss, si = fr.search.Matcher(g[lo:lo + 4096].contiguous(), idx_offset=lo, k=5).search(q[:32])
    assert np.abs(ss.cpu().numpy() - os_).max() <= 1e-3 and np.array_equal(si[:, 0].cpu().numpy(), oi[:, 0] + lo)


def test_embedding_to_gallery_roundtrip(golden_x):
    """Embeddings -> normalised bf16 gallery -> each image retrieves itself at rank 1."""
    m = product("mobilenetv3_small")
    with torch.no_grad():
        e = m(golden_x.to(DEV), precision="f32")
    gal = fr.search.normalize_rows_bf16(e)
    s, i = fr.search.Matcher(gal, k=5).search(gal)
    assert i[:, 0].cpu().tolist() == list(range(8))
    y, nrm = fr.search.l2norm(e)
    assert torch.allclose(y.norm(dim=1), torch.ones(8, device=DEV), atol=1e-5)
    assert torch.allclose(nrm, e.norm(dim=1), rtol=1e-5)


def test_gallery_files_json_import_search_and_ids(tmp_path, golden_x):
    """SURVEY.md 8(f) rank 3 end to end: reference-format JSON gallery (inference.py:153-160) -> import_json
    (fr_l2norm_bf16 on the GPU) -> shard files -> per-rank load -> search -> ids.  The stored rows equal the
    oracle's normalisation to bf16 rounding, and every enrolled face finds itself first."""
    import json
    from face_recognition_amd import gallery as G
    m = product("mobilenetv3_small")
    with torch.no_grad():
        emb = m(golden_x.to(DEV).repeat(40, 1, 1, 1) + 0.05 * torch.randn(len(golden_x) * 40, 3, 112, 112, device=DEV), precision="f32")
    names = [f"face_{i:03d}.jpg" for i in range(len(emb))]
    j = str(tmp_path / "embeddings.json")
    with open(j, "w") as f:
        json.dump({n: e for n, e in zip(names, emb.cpu().tolist())}, f, indent=2)
    g = G.import_json(j, str(tmp_path / "g"), DEV, shard_rows=100, chunk=64)
    assert len(g) == len(emb) and g.ids == names
    rows = g.load_rows(DEV)
    e64 = emb.cpu().numpy().astype("float64")
    ref = torch.from_numpy(e64 / np.linalg.norm(e64, axis=1, keepdims=True)).to(torch.float32)
    assert torch.equal(rows, fr.search.normalize_rows_bf16(emb))
    assert (rows.cpu().to(torch.float32) - ref).abs().max() <= 2.0 ** -8            # half a bf16 ulp below 1.0, with margin
    matcher = g.matcher(DEV, k=3)
    s, idx = matcher.search(rows)
    assert [r[0] for r in g.lookup(idx)] == names
    two = [g.matcher(DEV, rank=r, world=2, k=3).local.search(rows) for r in range(2)]
    ms, mi = fr.search.merge_topk(torch.stack([t[0] for t in two]), torch.stack([t[1] for t in two]), 3)
    assert torch.equal(mi, idx) and torch.equal(ms, s)
    built = G.build_from_images(m, DEV, ["a", "b"], [(golden_x[i].permute(1, 2, 0) * 127.5 + 127.5).round().to(torch.uint8).numpy() for i in range(2)],
                                str(tmp_path / "b"))
    assert built.rows == 2 and built.ids == ["a", "b"]


def test_alignment_warp_bit_exact_vs_oracle():
    """fr_align_faces (one launch for a batch of faces from images of different sizes) against oracle/align.py's
    restatement of cv2.warpAffine's fixed-point INTER_LINEAR / BORDER_CONSTANT path (utils/face_utils.py:56-71):
    integer arithmetic, so bit for bit.  The oracle itself is parity unpinned (cv2 is not in this image; see
    tests/test_host.py for the properties it is held to).  Ends by embedding the aligned faces (forward_u8)."""
    from oracle import align as oa
    rng = np.random.default_rng(42)
    sizes = [(250, 250), (300, 400), (112, 112), (97, 131), (480, 640), (250, 250)]
    imgs = [rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
    imgs[3] = np.full((97, 131, 3), 255, np.uint8)                                        # saturated image: no overflow
    tmpl = oa.REFERENCE_ALIGNMENT.astype(np.float64)
    Ms = []
    for i, (h, w) in enumerate(sizes):
        s, th = rng.uniform(0.5, 3.0), rng.uniform(-np.pi, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        lm = s * tmpl @ R.T + np.array([w, h]) * rng.uniform(0.0, 0.6, 2) + rng.normal(0, 2.0, (5, 2))
        Ms.append(oa.estimate_norm(lm))
    Ms[2] = np.array([[1, 0, 0], [0, 1, 0.0]])                                            # identity on a 112x112 image
    Ms[5] = np.array([[0.02, 0, 5], [0, 0.02, 7.0]])                                      # 50x minification, mostly outside
    Ms = np.stack(Ms)
    got = fr.align.warp_faces([torch.from_numpy(a).to(DEV) for a in imgs], Ms).cpu().numpy()
    assert got.shape == (6, 112, 112, 3)
    for i in range(6):
        assert np.array_equal(got[i], oa.warp_affine_u8(imgs[i], Ms[i])), i
    assert np.array_equal(got[2], imgs[2])
    # batched-tensor form, another output size, one face, no faces
    batch = torch.from_numpy(np.stack([imgs[0], imgs[5]])).to(DEV)
    g2 = fr.align.warp_faces(batch, Ms[[0, 5]], image_size=224).cpu().numpy()
    assert np.array_equal(g2[1], oa.warp_affine_u8(imgs[5], Ms[5], 224))
    lm0 = tmpl * 1.7 + 20.0
    one = fr.align.face_alignment(torch.from_numpy(imgs[4]).to(DEV), lm0).cpu().numpy()
    assert np.array_equal(one, oa.face_alignment(imgs[4], lm0))
    assert fr.align.warp_faces([], np.zeros((0, 2, 3))).shape == (0, 112, 112, 3)
    with pytest.raises(RuntimeError):
        fr.align.warp_faces([torch.from_numpy(imgs[0])], Ms[:1])
    emb = product("mobilenetv3_small").forward_u8(torch.from_numpy(got).to(DEV), precision="bf16")
    assert emb.shape == (6, 512) and torch.isfinite(emb).all()


# ---- the reference's file-based entry points on the GPU (inference.py:109-177, evaluate.py:179-194) ------------------
def _write_faces(tmp_path, n=5):
    from PIL import Image
    faces = osc.synthetic_faces(n, 1, seed=77).reshape(n, 112, 112, 3)
    names = [f"face_{i}.png" for i in range(n)]
    names[1] = "face_1.PNG"                                  # upper-case extension: globbed too (inference.py:141-144)
    for nme, a in zip(names, faces):
        Image.fromarray(a).save(str(tmp_path / nme), format="PNG")       # PNG: lossless, so the oracle sees the same pixels
    return names, faces


@pytest.mark.parametrize("name", ["mobilenetv3_small", "sphere20"])
def test_file_entry_points_against_oracle(name, tmp_path, capsys):
    """extract_features, extract_deep_features, compare_faces and extract_batch_embeddings called the way the
    reference's __main__ and notebooks call them -- image FILES in, NumPy / JSON out, model from load_model() on
    cuda with the default precision -- against the oracle forward and oracle/scoring.py on the same pixels:
    embedding cosine >= 0.9999, similarity |delta| <= 1e-3, same decision, same JSON keys and values."""
    import json
    from PIL import Image
    names, faces = _write_faces(tmp_path)
    (tmp_path / "broken.jpg").write_bytes(b"not an image")
    (tmp_path / "notes.txt").write_text("ignored")
    onet = ob.seeded(name, 0)
    ck = tmp_path / "w.pth"
    torch.save(onet.state_dict(), ck)
    model = fr.load_model(name, str(ck), DEV)
    assert not model.training and next(model.parameters()).is_cuda
    with torch.no_grad():
        want = onet(torch.stack([osc.to_model_input(a) for a in faces])).numpy()

    f0 = fr.extract_features(model, DEV, str(tmp_path / names[0]))
    assert isinstance(f0, np.ndarray) and f0.shape == (512,) and f0.dtype == np.float32
    assert cos_rows(f0[None], want[:1]).min() >= COS_BAR
    t0 = fr.evaluate.extract_deep_features(model, Image.open(str(tmp_path / names[0])).convert("RGB"), DEV)
    assert isinstance(t0, torch.Tensor) and not t0.is_cuda and t0.shape == (512,)
    assert np.array_equal(t0.numpy(), f0)
    with pytest.raises(FileNotFoundError):
        fr.extract_features(model, DEV, str(tmp_path / "missing.png"))
    with pytest.raises(FileNotFoundError):
        fr.extract_features(model, DEV, str(tmp_path / "broken.jpg"))

    for i, j in ((0, 1), (2, 3), (4, 4)):
        sim, same = fr.compare_faces(model, DEV, str(tmp_path / names[i]), str(tmp_path / names[j]), threshold=0.35)
        ref = osc.compute_similarity(want[i], want[j])
        assert type(sim) is np.float32 and isinstance(same, (bool, np.bool_))
        assert abs(float(sim) - float(ref)) <= 1e-3, (i, j, sim, ref)
        if abs(float(ref) - 0.35) > 1e-3:
            assert bool(same) == bool(ref > 0.35)
    assert fr.compare_faces(model, DEV, str(tmp_path / names[4]), str(tmp_path / names[4]))[1]      # a face matches itself
    sim_host = fr.compute_similarity(want[0], want[1])
    assert type(sim_host) is np.float32 and sim_host == osc.compute_similarity(want[0], want[1])
    with pytest.raises(FileNotFoundError):
        fr.compare_faces(model, DEV, str(tmp_path / names[0]), str(tmp_path / "missing.png"))

    out = tmp_path / "assets" / "embeddings.json"
    d = fr.extract_batch_embeddings(model, DEV, str(tmp_path), str(out))
    printed = capsys.readouterr().out
    assert sorted(d) == sorted(names) and "broken.jpg" not in d
    assert "✗ Error processing broken.jpg" in printed and printed.count("✓ Processed") == len(names)
    assert f"Extracting embeddings from {len(names) + 1} images" in printed
    marks = [ln[len("✓ Processed: "):] if ln.startswith("✓") else ln[len("✗ Error processing "):].split(":")[0]
             for ln in printed.splitlines() if ln.startswith(("✓ Processed: ", "✗ Error processing "))]
    assert [n for n in marks if n != "broken.jpg"] == list(d) and "broken.jpg" in marks
    order = [p_.name for ext in (".jpg", ".jpeg", ".png", ".bmp", ".tiff") for pat in (ext, ext.upper())
             for p_ in __import__("pathlib").Path(tmp_path).glob(f"*{pat}")]
    assert marks == order                                    # ✓ / ✗ lines interleave in file order (inference.py:150-156)
    back = json.load(open(out))
    assert list(back) == list(d) and all(isinstance(v, list) and len(v) == 512 for v in back.values())
    got = np.array([back[n] for n in names], dtype=np.float32)
    assert cos_rows(got, want).min() >= COS_BAR
    assert np.array_equal(got[0], f0)                        # batched u8 path == the per-image path, bit for bit
    assert open(out).read().startswith("{\n  \"")            # indent=2, as json.dump(..., indent=2) in the reference
    # batches smaller than the folder: same dict
    d2 = fr.extract_batch_embeddings(model, DEV, str(tmp_path), str(tmp_path / "a2" / "e.json"), batch_size=2)
    assert d2 == d


def test_eval_lfw_from_jpeg_files_and_state_dict_path(tmp_path, golden_dir):
    """evaluate.eval exactly as train.py / the Eval notebook call it: a directory of .jpg files in LFW layout plus
    lfw_ann.txt, `model_path` pointing at a bare state_dict (evaluate.py:380-382), default image decoding (PIL) --
    against the oracle's pair loop on the SAME decoded pixels.  BN-calibrated weights, so decisions are non-trivial."""
    from PIL import Image
    faces = osc.synthetic_faces(12, 2, seed=4321)
    root = tmp_path / "val"
    images = {}
    for i in range(12):
        (root / f"p{i}").mkdir(parents=True)
        for j in range(2):
            f = root / f"p{i}" / f"p{i}_{j + 1:04d}.jpg"
            Image.fromarray(faces[i, j]).save(str(f), format="JPEG", quality=95)
            images[str(f)] = np.asarray(Image.open(str(f)).convert("RGB"), dtype=np.uint8)      # what any reader decodes
    lines = ["24\n"] + [f"p{i} 1 2\n" for i in range(12)] + [f"p{i} 1 p{(i + 5) % 12} 2\n" for i in range(12)] + ["p0 1 p1 7\n"]
    (root / "lfw_ann.txt").write_text("".join(lines))
    g = np.load(os.path.join(golden_dir, "bncal_mobilenetv3_small.npz"))
    onet = ob.seeded("mobilenetv3_small", 0)
    sd = onet.state_dict()
    for i, k in enumerate(g["keys"]):
        sd[str(k)] = torch.from_numpy(g[f"s{i}"])
    onet.load_state_dict(sd)
    torch.save(sd, tmp_path / "w.pth")
    torch.manual_seed(5)
    model = fr.get_network("mobilenetv3_small")                   # other weights: eval must load model_path over them
    mean, preds, metrics = fr.evaluate.eval(model, model_path=str(tmp_path / "w.pth"), device=DEV, val_root=str(root),
                                            compute_full_metrics=True)
    omean, opreds, _ = osc.eval_pairs(onet, images, lines[1:], str(root))
    assert preds.shape == opreds.shape == (24, 4)                 # the pair with the missing image is skipped
    assert list(preds[:, 0]) == list(opreds[:, 0]) and list(preds[:, 1]) == list(opreds[:, 1]) and list(preds[:, 3]) == list(opreds[:, 3])
    s, o = preds[:, 2].astype(float), opreds[:, 2].astype(float)
    assert np.abs(s - o).max() <= 1e-3 and abs(mean - omean) <= 1e-3
    far = np.abs(o - 0.35) > 1e-3
    assert np.array_equal(osc.decide(s)[far], osc.decide(o)[far]) and osc.decide(o).sum() > 0
    assert metrics["true_positives"] + metrics["false_negatives"] == 12 and 0.0 <= metrics["accuracy"] <= 1.0
    with pytest.warns(RuntimeWarning, match="save_metrics_path"):     # evaluate.py:578-586 plots: not written, and said so
        fr.evaluate.eval(model, device=DEV, val_root=str(root), save_metrics_path=str(tmp_path / "plots"))


def test_file_entry_points_resize_other_sizes(tmp_path):
    """Images that are not 112x112 go through the Resize step (PIL bilinear, as torchvision does for PIL inputs;
    parity unpinned: torchvision is not importable here).  Host logic only -- the oracle side uses the same PIL call."""
    from PIL import Image
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, size=(150, 130, 3), dtype=np.uint8)
    Image.fromarray(a).save(str(tmp_path / "big.png"))
    model = product("mobilenetv3_small")
    f = fr.extract_features(model, DEV, str(tmp_path / "big.png"))
    small = np.asarray(Image.fromarray(a).resize((112, 112), Image.BILINEAR), dtype=np.uint8)
    with torch.no_grad():
        want = ob.seeded("mobilenetv3_small", 0)(osc.to_model_input(small)[None]).numpy()
    assert cos_rows(f[None], want).min() >= COS_BAR


# ---- robustness ------------------------------------------------------------------------------
def test_split_failure_joins_auxiliary_streams(golden_x, monkeypatch):
    """fr_embed cuts a batch >= 128 into two parts on two streams.  When the second part fails half way (test hook
    fr_debug_fail_after: an error after that many layers were enqueued on the auxiliary stream), the call must
    return the error AND leave the caller's stream waiting for what was already enqueued: afterwards the same
    workspace is reused at once by a good call, whose result must be right."""
    m = product("mobilenetv3_small")
    x = golden_x.to(DEV).repeat(32, 1, 1, 1)                 # 256 images: 128 + 128
    with torch.no_grad():
        good = m(x, precision="bf16")
        lib = fr._lib.lib()
        lib.fr_debug_fail_after(9)
        try:
            with pytest.raises(RuntimeError, match="injected failure"):
                m(x, precision="bf16")
        finally:
            lib.fr_debug_fail_after(-1)
        again = m(x, precision="bf16")                       # same plan, same workspace, no synchronisation in between
    torch.cuda.synchronize(DEV)
    assert torch.equal(again, good)


def test_two_streams_get_two_workspaces():
    """Two searches and two forwards enqueued from one host thread on two torch streams may overlap on the GPU: the
    host layer must not hand them one shared workspace (ops._workspace is keyed by stream)."""
    from face_recognition_amd import ops
    G = 400_000
    g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=8, n_gallery_total=G)
    qa = fr.search.synth_rows(11, 0, 3000, DEV)
    qb = fr.search.synth_rows(13, 0, 3000, DEV)
    ma, mb = fr.search.Matcher(g, k=5), fr.search.Matcher(g, k=5)
    ra, rb = ma.search(qa), mb.search(qb)                    # reference results, one after the other
    torch.cuda.synchronize(DEV)
    s1, s2 = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    outs = []
    for _ in range(3):
        with torch.cuda.stream(s1):
            oa = ma.search(qa)
        with torch.cuda.stream(s2):
            ob_ = mb.search(qb)
        outs.append((oa, ob_))
    torch.cuda.synchronize(DEV)
    for oa, ob_ in outs:
        assert torch.equal(oa[1], ra[1]) and torch.equal(oa[0], ra[0])
        assert torch.equal(ob_[1], rb[1]) and torch.equal(ob_[0], rb[0])
    keys = [k for k in ops._workspaces if k[0] == 0 and k[1] == DEV.index]
    assert len({k[2] for k in keys}) >= 3                    # default stream + the two side streams
    # the same for the embedding path: one model, two streams, split batches (each call also forks onto the plan's
    # auxiliary stream) -- bit-identical to the serial result, one workspace per stream
    m = product("mobilenetv3_small")
    rng = torch.Generator().manual_seed(4)
    xa = (torch.rand((160, 3, 112, 112), generator=rng) * 2 - 1).to(DEV)
    xb = (torch.rand((131, 3, 112, 112), generator=rng) * 2 - 1).to(DEV)
    with torch.no_grad():
        wa, wb = m(xa), m(xb)
        torch.cuda.synchronize(DEV)
        outs = []
        for _ in range(3):
            with torch.cuda.stream(s1):
                ya = m(xa)
            with torch.cuda.stream(s2):
                yb = m(xb)
            outs.append((ya, yb))
    torch.cuda.synchronize(DEV)
    for ya, yb in outs:
        assert torch.equal(ya, wa) and torch.equal(yb, wb)
    handle = int(m.plan(DEV).handle.value)
    assert len({k[2] for k in ops._workspaces if k[0] == handle}) == 3
    # bounded: many short-lived streams do not grow the cache past MAX_STREAM_WORKSPACES per plan
    for _ in range(ops.MAX_STREAM_WORKSPACES + 4):
        with torch.cuda.stream(torch.cuda.Stream(DEV)), torch.no_grad():
            assert torch.equal(m(xb[:3]), wb[:3])
    assert len([k for k in ops._workspaces if k[0] == handle]) <= ops.MAX_STREAM_WORKSPACES
    assert ops.drop_stream_workspaces(s1.cuda_stream) >= 1 and not [k for k in ops._workspaces if k[2] == s1.cuda_stream]


def test_search_config4_at_full_size_by_properties():
    """BASELINE.json configs[3] at its full size (10 M gallery rows x 65 536 queries, 512-d bf16, k = 5), far beyond
    what the oracle scores: every one of the 1024 planted queries finds its planted row first, scores are sorted,
    the five rows of a query are distinct and valid, a 4096-query slice searched alone gives the same rows (other
    launch geometry), and a 4096-row window of the gallery agrees with the oracle."""
    G, Q, npl = 10_000_000, 65_536, 1024
    g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=npl, n_gallery_total=G)
    q = fr.search.synth_rows(11, 0, Q, DEV)
    s, i = fr.search.Matcher(g, k=5).search(q)
    planted = torch.tensor([osr.plant_row(t, G, npl) for t in range(npl)], device=DEV)
    assert torch.equal(i[:npl, 0], planted) and (s[:npl, 0] > 0.9).all()
    assert (s[:, 1:] <= s[:, :-1]).all() and (i >= 0).all() and (i < G).all()
    assert (i.sort(dim=1).values.diff(dim=1) > 0).all()
    assert (s[npl:, 0] < 0.5).all()                          # unplanted queries: no accidental near-duplicate
    s2, i2 = fr.search.Matcher(g, k=5).search(q[40_000:44_096])
    assert torch.equal(i2, i[40_000:44_096]) and torch.equal(s2, s[40_000:44_096])
    lo = 7_000_000
    gf = osr.bf16_bits_to_f32(g[lo:lo + 4096].view(torch.int16).cpu().numpy().view(np.uint16))
    qf = osr.bf16_bits_to_f32(q[:32].view(torch.int16).cpu().numpy().view(np.uint16))
    os_, oi, _ = osr.topk_cosine(qf, gf, 5)
    ss, si = fr.search.Matcher(g[lo:lo + 4096].contiguous(), idx_offset=lo, k=5).search(q[:32])
    assert np.abs(ss.cpu().numpy() - os_).max() <= 1e-3 and np.array_equal(si[:, 0].cpu().numpy(), oi[:, 0] + lo)


def test_packed_exchange_merge_equals_plain_merge():
    """fr_topk_search writing straight into packed exchange messages + fr_topk_merge_packed on the concatenated
    messages (what ShardedMatcher does around its one all_gather per chunk) against the plain search + merge:
    3 fake shards on one device, odd Q*k (padding between the score and index blocks), bit for bit."""
    G, Q, k = 30011, 77, 5
    g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=8, n_gallery_total=G)
    q = fr.search.synth_rows(11, 0, Q, DEV)
    s1, i1 = fr.search.Matcher(g, k=k).search(q)
    P = 3
    off, total = fr.search.packed_bytes(Q, k)
    assert off % 8 == 0 and off == 1544 and total == off + Q * k * 8
    recv = torch.empty((P, total), dtype=torch.uint8, device=DEV)
    for r in range(P):
        lo, hi = fr.search.shard_range(G, r, P)
        sv, iv = fr.search.packed_views(recv[r], Q, k)
        fr.search.Matcher(g[lo:hi], idx_offset=lo, k=k).search_into(q, sv, iv)
    sm = fr.search.ShardedMatcher(g[:fr.search.shard_range(G, 0, P)[1]], G, 0, P, k=k)
    out_s = torch.empty((Q, k), dtype=torch.float32, device=DEV)
    out_i = torch.empty((Q, k), dtype=torch.int64, device=DEV)
    sm._merge(recv, Q, out_s, out_i)
    assert torch.equal(out_i, i1) and torch.equal(out_s, s1)


@pytest.mark.gpu
@pytest.mark.parametrize("name,prec", [("mobilenetv3_large", "f32"), ("mobilenetv3_small", "bf16"), ("sphere20", "bf16")])
def test_hipgraph_replay_equals_eager(name, prec, monkeypatch):
    """fr_embed_graph_* (include/fr_hip.h): the captured launch sequence replayed on static buffers gives the bits of
    fr_embed, for new inputs on every replay, for the uint8 form, and through model(x) with FR_GRAPH_MAX set."""
    import face_recognition_amd as fr
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    m = fr.get_network(name).to(dev).eval()
    m.precision = prec
    plan = m.plan(dev)
    for b in (1, 5):
        g = plan.graph(b)
        assert g.nodes >= 20
        for seed in (0, 1):
            x = torch.rand((b, 3, 112, 112), generator=torch.Generator().manual_seed(seed)) * 2 - 1      # CPU tensor: H2D into the static buffer
            want = m(x.to(dev))
            got = g(x)
            assert got.data_ptr() == g.out.data_ptr()
            assert torch.equal(got, want)
    u8 = torch.randint(0, 256, (3, 112, 112, 3), dtype=torch.uint8, device=dev)
    assert torch.equal(plan.graph(3, u8=True)(u8), m.forward_u8(u8))
    with pytest.raises(ValueError):
        plan.graph(1)(torch.zeros((2, 3, 112, 112), device=dev))
    monkeypatch.setenv("FR_GRAPH_MAX", "8")
    x = torch.rand((4, 3, 112, 112), device=dev) * 2 - 1
    y = m(x)
    monkeypatch.setenv("FR_GRAPH_MAX", "0")
    assert torch.equal(y, m(x))
    assert (4, False, torch.cuda.current_stream(dev).cuda_stream) in plan._graphs
    m.invalidate()                                             # destroys the graphs with the plan


def test_rccl_exchange_in_process_equals_single_matcher():
    """north_star (4) / BASELINE.json configs[4]: the exchange step on the real collective library.  An `nccl` (= RCCL)
    process group of ONE rank is initialised inside this process (tcp://127.0.0.1 rendezvous; utils/general.py:143-164
    is the reference's convention for the backend), and ShardedMatcher.search runs its pipelined form -- local
    fr_topk_search into the packed message, all_gather_into_tensor(async_op=True) on the communicator's stream,
    fr_topk_merge_packed on the receive buffer -- over THREE query chunks (the last one ragged).  Must equal
    Matcher.search bit for bit.  More than one rank cannot run on this pool's one-GPU boxes; that part of the path is
    covered by the world-size-2 gloo test (tests/test_host.py) and by the fake-shard tests above."""
    import socket
    import torch.distributed as dist
    assert not dist.is_initialized()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    G, Q, k = 300_000, 2 * 4096 + 123, 5
    g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=16, n_gallery_total=G)
    q = fr.search.synth_rows(11, 0, Q, DEV)
    want_s, want_i = fr.search.Matcher(g, k=k).search(q)
    torch.cuda.set_device(DEV)
    dist.init_process_group("nccl", rank=0, world_size=1, init_method=f"tcp://127.0.0.1:{port}", device_id=DEV)
    try:
        assert dist.get_backend() == "nccl"
        sm = fr.search.ShardedMatcher(g, G, 0, 1, k=k, chunk=4096)
        s, i = sm.search(q)
        torch.cuda.synchronize(DEV)
        assert torch.equal(i, want_i) and torch.equal(s, want_s)
        s2, i2 = sm.search(q[:100])                            # one short chunk, communicator reused
        assert torch.equal(i2, want_i[:100]) and torch.equal(s2, want_s[:100])
        gs, gi = fr.search.all_gather_topk(want_s, want_i)     # the unpipelined two-collective form
        assert torch.equal(gs[0], want_s) and torch.equal(gi[0], want_i)
    finally:
        dist.destroy_process_group()


def test_search_100M_gallery_on_one_gpu_by_properties():
    """BASELINE.json configs[4]'s gallery -- 100 M rows x 512 bf16 = 102.4 GB, the P = 1 point of the 2/4/8-GPU curve
    -- resident on ONE MI355X, 4096 queries, checked through size-independent properties: the 1024 planted queries
    find their planted rows first, scores sorted, five distinct valid rows per query, 8 contiguous shards (the
    partitioning of ShardedMatcher at P = 8, as views of the same buffer) merged == the unsharded search bit for bit,
    a query sub-range searched alone gives the same rows, and a 4096-row window agrees with the oracle."""
    G, Q, npl = 100_000_000, 4096, 1024
    free, _ = torch.cuda.mem_get_info(DEV)
    if free < 120 * (1 << 30):
        torch.cuda.empty_cache()
        free, _ = torch.cuda.mem_get_info(DEV)
    assert free > 110 * (1 << 30), f"needs 102.4 GB of free HBM, have {free >> 30} GiB"
    g = fr.search.synth_rows(7, 0, G, DEV, plant_seed=11, n_plant=npl, n_gallery_total=G)
    try:
        q = fr.search.synth_rows(11, 0, Q, DEV)
        s, i = fr.search.Matcher(g, k=5).search(q)
        planted = torch.tensor([osr.plant_row(t, G, npl) for t in range(npl)], device=DEV)
        assert torch.equal(i[:npl, 0], planted) and (s[:npl, 0] > 0.9).all()
        assert (s[:, 1:] <= s[:, :-1]).all() and (i >= 0).all() and (i < G).all()
        assert (i.sort(dim=1).values.diff(dim=1) > 0).all()
        assert (s[npl:, 0] < 0.5).all()
        parts = []
        for r in range(8):
            lo, hi = fr.search.shard_range(G, r, 8)
            parts.append(fr.search.Matcher(g[lo:hi], idx_offset=lo, k=5).search(q))
        ms, mi = fr.search.merge_topk(torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), 5)
        assert torch.equal(mi, i) and torch.equal(ms, s)
        s2, i2 = fr.search.Matcher(g, k=5).search(q[3000:3500])
        assert torch.equal(i2, i[3000:3500]) and torch.equal(s2, s[3000:3500])
        lo = 93_000_000
        gf = osr.bf16_bits_to_f32(g[lo:lo + 4096].view(torch.int16).cpu().numpy().view(np.uint16))
        qf = osr.bf16_bits_to_f32(q[:32].view(torch.int16).cpu().numpy().view(np.uint16))
        os_, oi, _ = osr.topk_cosine(qf, gf, 5)
        ss, si = fr.search.Matcher(g[lo:lo + 4096].contiguous(), idx_offset=lo, k=5).search(q[:32])
        assert np.abs(ss.cpu().numpy() - os_).max() <= 1e-3 and np.array_equal(si[:, 0].cpu().numpy(), oi[:, 0] + lo)
    finally:
        del g
        torch.cuda.empty_cache()


def test_pair_verification_at_config0_size_by_properties(tmp_path, golden_dir):
    """BASELINE.json configs[0] at its size -- 12 000 synthetic faces, 6000 LFW-layout pairs (3000 same / 3000
    different), mobilenetv3_small -- through evaluate.eval in the default precision (the reference: evaluate.py:517-555,
    12 000 batch-1 forwards).  BN-calibrated weights so that decisions are not all "different".  Properties: every pair
    is scored, in annotation order; the similarity of a pair scored inside the deduplicated 1024-image batches equals
    the similarity of the same two faces embedded ONE AT A TIME, bit for bit (100 pairs); similarities and decisions
    equal the CPU oracle's pair loop on a 200-pair sample (|delta| <= 1e-3, decisions identical away from the
    threshold); mean_similarity is the mean of the column."""
    import bench
    faces, lines = bench.lfw_layout_faces(6000)
    root = str(tmp_path)
    open(os.path.join(root, "lfw_ann.txt"), "w").write("".join(lines))
    keyed = {os.path.join(root, k): v for k, v in faces.items()}
    sd, _ = _calibrated_state("mobilenetv3_small", golden_dir)
    model = product("mobilenetv3_small", sd)
    mean, preds, metrics = fr.evaluate.eval(model, device=DEV, val_root=root, image_loader=keyed.__getitem__, compute_full_metrics=True)
    assert preds.shape == (6000, 4) and len(keyed) == 12000
    want_rows = [osc.parse_lfw_line(root, ln) for ln in lines[1:]]
    assert [tuple(r) for r in preds[:, [0, 1, 3]].tolist()] == want_rows
    sims = preds[:, 2].astype(float)
    assert abs(mean - sims.mean()) < 1e-12 and list(preds[:3000, 3]) == ["1"] * 3000 and list(preds[3000:, 3]) == ["0"] * 3000
    assert metrics["true_positives"] + metrics["false_negatives"] == 3000
    assert 0 < (sims > 0.35).sum() < 6000                                   # calibrated weights: both decisions occur
    pick = list(range(0, 6000, 60))                                         # 100 pairs, same and different
    with torch.no_grad():
        ea = torch.cat([model.forward_u8(torch.from_numpy(keyed[preds[i, 0]][None]).to(DEV)) for i in pick])
        eb = torch.cat([model.forward_u8(torch.from_numpy(keyed[preds[i, 1]][None]).to(DEV)) for i in pick])
    one_by_one, _ = fr.cosine_pairs(ea, eb, eps=1e-5)
    assert np.array_equal(one_by_one.cpu().numpy().astype(np.float64), sims[pick])
    onet = ob.seeded("mobilenetv3_small", 0)
    onet.load_state_dict(sd)
    sample = list(range(0, 6000, 30))                                       # 200 pairs
    _, opreds, _ = osc.eval_pairs(onet, keyed, [lines[1 + i] for i in sample], root)
    o = opreds[:, 2].astype(float)
    assert np.abs(o - sims[sample]).max() <= 1e-3
    far = np.abs(o - 0.35) > 1e-3
    assert np.array_equal(osc.decide(sims[sample])[far], osc.decide(o)[far])


@pytest.mark.parametrize("ascending", [True, False])
def test_topk_search_parity_on_score_ordered_gallery(ascending):
    """The insertion path that random galleries almost never take: a gallery whose score against
    EVERY query rises (or falls) steadily with the row index -- rows = cos(t) u + sin(t) w with t monotone in the row,
    queries = u plus a little noise -- so that with ascending order every 64-row tile beats the running top-5 of every
    lane and the per-lane lists are rewritten on every tile.  Against the oracle's fp32 top-k on the same bf16 rows:
    scores within 1e-3, indices identical wherever neighbouring scores differ by more than 2e-4, and the three kernel
    forms agree bit for bit.  (The RATE on such a gallery is 0.5x the random one: profiles/r02_search_adversarial_*.json.)"""
    G, Q, k = 40_000, 200, 5
    gen = torch.Generator(device="cpu").manual_seed(3)
    u = torch.randn(512, generator=gen); u = u / u.norm()
    w = torch.randn(512, generator=gen); w = w - (w @ u) * u; w = w / w.norm()
    r = torch.arange(G, dtype=torch.float64) / G
    score = ((r if ascending else 1 - r) * 0.95)
    x = score.float()[:, None] * u[None] + torch.sqrt(1 - score * score).float()[:, None] * w[None]
    x = x + torch.randn((G, 512), generator=gen) * (0.02 / 512 ** 0.5)                 # breaks exact ties between neighbours
    qx = u[None] + torch.randn((Q, 512), generator=gen) * (0.05 / 512 ** 0.5)
    g = fr.search.normalize_rows_bf16(x.to(DEV))
    q = fr.search.normalize_rows_bf16(qx.to(DEV))
    s, i = fr.search.Matcher(g, k=k).search(q)
    gf = osr.bf16_bits_to_f32(g.view(torch.int16).cpu().numpy().view(np.uint16))
    qf = osr.bf16_bits_to_f32(q.view(torch.int16).cpu().numpy().view(np.uint16))
    os_, oi, _ = osr.topk_cosine(qf, gf, k)
    s_, i_ = s.cpu().numpy(), i.cpu().numpy()
    assert np.abs(s_ - os_).max() <= 1e-3
    assert (np.diff(s_, axis=1) <= 0).all() and (i_ >= 0).all() and (i_ < G).all()
    full = qf.astype(np.float64) @ gf.astype(np.float64).T / (np.linalg.norm(qf, axis=1)[:, None] * np.linalg.norm(gf, axis=1)[None])
    srt = -np.sort(-full, axis=1)[:, :k + 1]
    clear = (srt[:, :-1] - srt[:, 1:]) > 2e-4                                          # rank e is separated from rank e + 1 ...
    clear[:, 1:] &= clear[:, :-1]                                                      # ... and from rank e - 1
    assert np.array_equal(i_[clear], oi[clear])
    # where neighbours are closer than that (a steady trend of 2e-5 per row): every returned row is a legitimate member
    # of the top k within the score tolerance, and the rows of a query are distinct
    got_true = np.take_along_axis(full, i_, axis=1)
    assert (got_true >= srt[:, k - 1:k] - 1e-3).all() and (np.abs(got_true - s_) <= 1e-3).all()
    assert (np.diff(np.sort(i_, axis=1), axis=1) > 0).all()
    if ascending:
        assert (i_ > G - 3000).all()                                                   # the winners sit at the end: lists rewritten all the way
    import os as _os
    old = {v: _os.environ.get(v) for v in ("FR_SEARCH_MFMA", "FR_SEARCH_RQ")}
    try:
        _os.environ["FR_SEARCH_MFMA"] = "32"
        s1, i1 = fr.search.Matcher(g, k=k).search(q)
        _os.environ["FR_SEARCH_RQ"] = "0"
        s0, i0 = fr.search.Matcher(g, k=k).search(q)
    finally:
        for v, val in old.items():
            if val is None:
                _os.environ.pop(v, None)
            else:
                _os.environ[v] = val
    assert torch.equal(i1, i) and torch.equal(s1, s) and torch.equal(i0, i) and torch.equal(s0, s)
