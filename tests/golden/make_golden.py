#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Run once, in the development container only (the reference lives at
/root/reference and never travels to the GPU box):

    cd /tmp && python -B /root/repo/tests/golden/make_golden.py

The fixtures are data only: seeded inputs, and the outputs the reference's own
functions produced for them (torch CPU, fp32).  Nothing of the reference's
source is stored.  The reference needs `timm` (absent here) only for six
activation class names (quantization/hijacker.py:7-8) -> stub modules.

Fixture files:
  g1_quantize.npz    quantize_to_fp8_ste_MM   (fp8_quantizer.py:91-133)
  g2_grids.npz       generate_all_values_fp   (fp8_quantizer.py:13-41)
  g3_estimators.npz  Current/All/RunningMinMax (range_estimators.py:56-125) + set_quant_range
  g3b_signed_zero.npz the same estimators on rows that mix -0.0 and +0.0 (pins the signed-zero contract of min / max)
  g4_mse.npz         FP_MSE_Estimator         (range_estimators.py:285-369)
  g4c_mse_f64.npz    the same estimator on float64 data (grid from the float64 maximum, float64 means into the float32 table)
  g5_quant_error.npz compute_quant_error.py (config 1) at 200 k samples + closed-form integrals
  g6_manager.npz     QuantizationManager.forward state machine (quantization_manager.py:114-122)
  g7_tinycnn.npz     quantize_model on a tiny CNN (autoquant_utils.py:292-381), config-3 settings
  g8_resnet18.npz    QuantizedResNet (models/resnet_quantized.py:49-133), BASELINE config 3 at 64x64
  g9_mobilenetv2.npz QuantizedMobileNetV2 (models/mobilenet_v2_quantized.py:29-92) + MSE, config 4 at 64x64
  g1b_bulk.npz       quantize_to_fp8_ste_MM on 4 x 4 M seeded normals: output hash + sparse difference to the C oracle
  g10_autograd.npz   backward of quantize_to_fp8_ste_MM (d/dx, d/dmaxval)
  g11_wrappers.npz   quantize_model on two toy nets built of the wrappers g7-g9 do not reach: QuantConv1d, QuantConvTranspose1d/2d
                     (dims 0/1 swapped around the per-channel quantizer), BNQConv1d, BNQLinear, QuantLayerNorm (autoquant_utils.py:20-174)
  g12_allow_unsigned.npz allow_unsigned over three batches through QuantizationManager (min/max estimators, MSE with a fixed width):
                     outputs, maxval and sign_bits after every batch (fp8_quantizer.py:216-225 is sticky)
  g1c_quantize_f64.npz quantize_to_fp8_ste_MM on FLOAT64 inputs (ATen type promotion: bias float32, the rest float64)
  (g5 also holds LineSearchEstimator.loss_array -- 1001 float64 sums per distribution and format -- and the chosen index)
"""
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_stubs():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m

    mk = lambda n: type(n, (nn.Module,), {})
    stub("timm")
    stub("timm.models")
    stub("timm.models.layers")
    stub("timm.models.layers.activations", Swish=mk("Swish"), HardSwish=mk("HardSwish"),
         HardSigmoid=mk("HardSigmoid"))
    stub("timm.models.layers.activations_me", SwishMe=mk("SwishMe"), HardSwishMe=mk("HardSwishMe"),
         HardSigmoidMe=mk("HardSigmoidMe"))


_install_stubs()
sys.path.insert(0, REF)
from quantization.quantizers.fp8_quantizer import (  # noqa: E402
    quantize_to_fp8_ste_MM, FPQuantizer, generate_all_values_fp, generate_all_float_values_scaled)
from quantization.range_estimators import RangeEstimators  # noqa: E402
from quantization.quantization_manager import QuantizationManager, QMethods  # noqa: E402

torch.set_num_threads(1)


def default_maxval(M, n_bits=8):
    E = n_bits - 1 - M
    return float((2 - 2.0 ** (-M)) * 2.0 ** (2 ** E - 1 - 2 ** (E - 1)))


def edge_inputs(M, maxval, sign_bits, n_bits=8):
    """Hand-picked hard inputs for one (M, maxval): ties, binade edges, clamps, specials."""
    E = n_bits - sign_bits - M
    bias = 2.0 ** E - np.log2(maxval) + np.log2(2 - 2.0 ** (-M)) - 1
    vals = [0.0, -0.0, 1e-30, -1e-30, 1e-42, float(maxval), -float(maxval),
            float(maxval) * 1.5, -float(maxval) * 1.5, float("inf"), float("-inf"), float("nan"),
            float(np.nextafter(np.float32(maxval), np.float32(0))),
            float(np.nextafter(np.float32(maxval), np.float32(np.inf)))]
    pmax = int(2 ** E)
    ps = sorted(set([1, 2, 3, pmax // 2, pmax - 2, pmax - 1, pmax]))
    for p in ps:
        if p < 1:
            continue
        s = 2.0 ** (p - M - bias)
        lo = 2.0 ** (p - bias)  # lower edge of binade p
        for k in (0, 1, 2, 2 ** M - 1, 2 ** M, 2 ** M + 1, 2 ** (M + 1) - 2, 2 ** (M + 1) - 1):
            t = np.float32((k + 0.5) * s)  # rounding tie (exact when bias is an integer)
            vals += [float(t), float(np.nextafter(t, np.float32(0))),
                     float(np.nextafter(t, np.float32(np.inf))), -float(t)]
        b = np.float32(lo)
        vals += [float(b), float(np.nextafter(b, np.float32(0))),
                 float(np.nextafter(b, np.float32(np.inf))), -float(b)]
    return np.array(vals, dtype=np.float32)


def make_g1():
    out = {}
    rng = np.random.RandomState(1234)
    base = rng.randn(2048).astype(np.float32)
    out["base"] = base
    cases = []
    cid = 0
    for M in (1, 2, 3, 4, 5, 6):
        mv_list = [default_maxval(M), 3.0, 1.0, 0.7361, 0.0123]
        for mv in mv_list:
            for sb in (1, 0):
                if sb == 0 and mv not in (default_maxval(M), 0.7361):
                    continue
                # scale the normals so that ~2% of them clip
                x = np.concatenate([base * np.float32(mv / 2.3), edge_inputs(M, mv, sb)])
                xt = torch.from_numpy(x.copy())
                y = quantize_to_fp8_ste_MM(xt, 8, torch.Tensor([mv]), torch.Tensor([float(M)]), sb)
                out[f"c{cid}_x"] = x
                out[f"c{cid}_y"] = y.numpy()
                cases.append((cid, M, mv, sb, 1))
                cid += 1
    # per-channel maxval [8], x [8, 300]
    for M in (2, 3, 5):
        mvs = (np.abs(rng.randn(8)) * 2 + 0.05).astype(np.float32)
        x = (rng.randn(8, 300) * (mvs[:, None] / 2.0)).astype(np.float32)
        y = quantize_to_fp8_ste_MM(torch.from_numpy(x.copy()), 8, torch.from_numpy(mvs.copy()),
                                   torch.Tensor([float(M)]), 1)
        out[f"c{cid}_x"] = x
        out[f"c{cid}_y"] = y.numpy()
        out[f"c{cid}_maxval"] = mvs
        cases.append((cid, M, -1.0, 1, 8))
        cid += 1
    # all-zero channel -> maxval 0 -> NaN channel (fp8_quantizer.py:110,128)
    mvs = np.array([1.0, 0.0, 2.5], dtype=np.float32)
    x = (rng.randn(3, 64)).astype(np.float32)
    x[1] = 0
    y = quantize_to_fp8_ste_MM(torch.from_numpy(x.copy()), 8, torch.from_numpy(mvs.copy()),
                               torch.Tensor([3.0]), 1)
    out[f"c{cid}_x"] = x
    out[f"c{cid}_y"] = y.numpy()
    out[f"c{cid}_maxval"] = mvs
    cases.append((cid, 3, -1.0, 1, 3))
    cid += 1
    # non-integer mantissa bits are rounded half-to-even then clamped (fp8_quantizer.py:105)
    for mb in (2.5, 3.5, 0.2, 9.0):
        x = base[:512] * np.float32(0.5)
        y = quantize_to_fp8_ste_MM(torch.from_numpy(x.copy()), 8, torch.Tensor([1.7]),
                                   torch.Tensor([mb]), 1)
        out[f"c{cid}_x"] = x
        out[f"c{cid}_y"] = y.numpy()
        cases.append((cid, mb, 1.7, 1, 1))
        cid += 1
    out["cases"] = np.array(cases, dtype=np.float64)  # id, mbits, maxval(-1: per-channel), sign, n_maxval
    np.savez_compressed(os.path.join(OUT, "g1_quantize.npz"), **out)
    print("g1:", cid, "cases")


def make_g2():
    out = {}
    for e in (2, 3, 4, 5):
        for b in (2 ** (e - 1), 2 ** (e - 1) + 1, 1):
            out[f"e{e}_b{b}"] = generate_all_values_fp(8, e, b)
        out[f"scaled_e{e}"] = generate_all_float_values_scaled(8, e, 2 ** (e - 1), 3.0)
    np.savez_compressed(os.path.join(OUT, "g2_grids.npz"), **out)
    print("g2 ok")


def make_g3():
    out = {}
    torch.manual_seed(0)
    w = torch.randn(64, 3, 7, 7) * 0.1  # BASELINE config 2 tensor
    out["w"] = w.numpy()
    est = RangeEstimators.current_minmax.cls(per_channel=True)
    mn, mx = est(w)
    out["w_cur_pc_min"], out["w_cur_pc_max"] = mn.numpy(), mx.numpy()
    est = RangeEstimators.current_minmax.cls(per_channel=False)
    mn, mx = est(w)
    out["w_cur_pt_min"], out["w_cur_pt_max"] = mn.numpy(), mx.numpy()
    # quantizer range set from per-channel min/max (fp8_quantizer.py:222-240), then quantize
    q = FPQuantizer(n_bits=8, per_channel=True, mantissa_bits=2, maxval=None, set_maxval=True)
    q.set_quant_range(out_t(out["w_cur_pc_min"]), out_t(out["w_cur_pc_max"]))
    out["w_maxval"] = q.maxval.numpy()
    out["w_q_e5m2"] = q(w).numpy()
    # three sequential activation batches
    acts = [torch.randn(4, 8, 6, 6) * (1 + i) + 0.3 * i for i in range(3)]
    out["acts"] = np.stack([a.numpy() for a in acts])
    for name in ("allminmax", "running_minmax"):
        for pc in (False, True):
            est = RangeEstimators[name].cls(per_channel=pc)
            mins, maxs = [], []
            for a in acts:
                mn, mx = est(a)
                mins.append(mn.numpy().copy().reshape(-1))
                maxs.append(mx.numpy().copy().reshape(-1))
            out[f"{name}_pc{int(pc)}_min"] = np.stack(mins)
            out[f"{name}_pc{int(pc)}_max"] = np.stack(maxs)
    # NaN handling of min/max
    an = acts[0].clone()
    an[1, 2, 3, 4] = float("nan")
    est = RangeEstimators.allminmax.cls(per_channel=False)
    mn, mx = est(an)
    out["nan_min"], out["nan_max"] = mn.numpy(), mx.numpy()
    # allow_unsigned: ReLU output -> sign_bits 0 (fp8_quantizer.py:216-225)
    q = FPQuantizer(n_bits=8, per_channel=False, mantissa_bits=3, maxval=None, set_maxval=True,
                    allow_unsigned=True)
    r = torch.relu(acts[1])
    q.set_quant_range(r.min(), r.max())
    out["relu_x"] = r.numpy()
    out["relu_sign_bits"] = np.array(q.sign_bits)
    out["relu_maxval"] = q.maxval.numpy()
    out["relu_q"] = q(r).numpy()
    np.savez_compressed(os.path.join(OUT, "g3_estimators.npz"), **out)
    print("g3 ok")


def out_t(a):
    return torch.from_numpy(np.array(a))


def make_g4():
    out = {}
    torch.manual_seed(1)
    cfgs = [
        ("w_pc_fixm", (32, 3, 3, 3), True, False, 3),
        ("w_pc_srchm", (32, 3, 3, 3), True, True, 3),
        ("a_pt_fixm", (2, 8, 14, 14), False, False, 3),
        ("a_pt_srchm", (2, 8, 14, 14), False, True, 2),
    ]
    for name, shape, pc, incl, M in cfgs:
        q = FPQuantizer(n_bits=8, per_channel=pc, mantissa_bits=M, maxval=None, set_maxval=True,
                        mse_include_mantissa_bits=incl)
        est = RangeEstimators.MSE.cls(per_channel=pc, quantizer=q)
        xs = [torch.randn(*shape) * 0.2, torch.randn(*shape) * 0.3]
        for b, x in enumerate(xs):
            mn, mx = est(x)
            out[f"{name}_x{b}"] = x.numpy()
            out[f"{name}_mses{b}"] = est.mses.numpy().copy()
            out[f"{name}_min{b}"] = mn.numpy().copy()
            out[f"{name}_max{b}"] = mx.numpy().copy()
            out[f"{name}_mbits{b}"] = np.array(float(q.mantissa_bits))
        out[f"{name}_grid"] = est.search_grid.numpy().copy()
    # allow_unsigned + one-sided data: sign_bits stays 1 inside the search grid sign, range min = 0
    q = FPQuantizer(n_bits=8, per_channel=False, mantissa_bits=3, maxval=None, set_maxval=True,
                    mse_include_mantissa_bits=False, allow_unsigned=True)
    est = RangeEstimators.MSE.cls(per_channel=False, quantizer=q)
    x = torch.relu(torch.randn(2, 8, 14, 14))
    mn, mx = est(x)
    out["relu_x"] = x.numpy()
    out["relu_mses"] = est.mses.numpy().copy()
    out["relu_min"], out["relu_max"] = mn.numpy().copy(), mx.numpy().copy()
    out["relu_sign_bits"] = np.array(q.sign_bits)
    np.savez_compressed(os.path.join(OUT, "g4_mse.npz"), **out)
    print("g4 ok")


def make_g4c():
    """FP_MSE_Estimator on FLOAT64 data (ATen's type promotion: the search grid from a float64 maximum, float64 means
    added into the float32 table), two accumulating batches, per tensor and per channel"""
    out = {}
    torch.manual_seed(4)
    for name, shape, pc, incl in (("pt", (2, 8, 14, 14), False, True), ("pc", (12, 3, 3, 3), True, False)):
        q = FPQuantizer(n_bits=8, per_channel=pc, mantissa_bits=3, maxval=None, set_maxval=True, mse_include_mantissa_bits=incl)
        est = RangeEstimators.MSE.cls(per_channel=pc, quantizer=q)
        xs = [torch.randn(*shape, dtype=torch.float64) * 0.37, torch.randn(*shape, dtype=torch.float64) * 0.21]
        for b, x in enumerate(xs):
            mn, mx = est(x)
            out[f"{name}_x{b}"] = x.numpy()
            out[f"{name}_mses{b}"] = est.mses.numpy().copy()
            out[f"{name}_max{b}"] = mx.numpy().copy()
            out[f"{name}_mbits{b}"] = np.array(float(q.mantissa_bits))
        out[f"{name}_grid"] = est.search_grid.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g4c_mse_f64.npz"), **out)
    print("g4c ok")


def make_g12():
    """allow_unsigned over several batches (fp8_quantizer.py:216-225 inside QuantizationManager.forward in estimate state): the
    sign bit goes once the ranges are one-sided and never comes back.  Four data sequences (s = signed batch, r = ReLU batch)
    x the three min/max estimators x per tensor / per channel, and the MSE estimator with a fixed mantissa width (with the
    mantissa search the reference indexes past its table as soon as a one-sided batch is followed by another batch:
    range_estimators.py:337-347 -- nothing to pin there).  Stored: inputs, every batch's output, maxval and sign_bits."""
    out = {}
    torch.manual_seed(12)
    shapes = {0: (2, 8, 14, 14), 1: (24, 3, 5, 5)}
    raw = {pc: [torch.randn(*shapes[pc]) * (1 + i) for i in range(3)] for pc in (0, 1)}
    for pc in (0, 1):
        out[f"raw_pc{pc}"] = np.stack([b.numpy() for b in raw[pc]])
    seqs = {"signed": "sss", "relu": "rrr", "signed_then_relu": "srr", "relu_then_signed": "rss"}
    ests = {"current_minmax": RangeEstimators.current_minmax.cls, "allminmax": RangeEstimators.allminmax.cls,
            "running_minmax": RangeEstimators.running_minmax.cls, "MSE": RangeEstimators.MSE.cls}
    for ename, ecls in ests.items():
        for pc in (0, 1):
            for sname, kinds in seqs.items():
                qm = QuantizationManager(qmethod=QMethods.fp_quantizer.cls, init=ecls, per_channel=bool(pc),
                                         qparams=dict(n_bits=8, mantissa_bits=3, maxval=None, set_maxval=True,
                                                      mse_include_mantissa_bits=False, allow_unsigned=True))
                ys, mvs, sgn = [], [], []
                for b, k in zip(raw[pc], kinds):
                    x = torch.relu(b) if k == "r" else b
                    ys.append(qm(x).numpy().copy())
                    mvs.append(qm.quantizer.maxval.numpy().copy().reshape(-1))
                    sgn.append(int(qm.quantizer.sign_bits))
                key = f"{ename}_pc{pc}_{sname}"
                out[key + "_y"] = np.stack(ys)
                out[key + "_maxval"] = np.stack(mvs)
                out[key + "_sign"] = np.array(sgn)
    np.savez_compressed(os.path.join(OUT, "g12_allow_unsigned.npz"), **out)
    print("g12 ok")


def make_g6():
    """QuantizationManager state machine: estimate -> fix, ranges frozen afterwards."""
    out = {}
    torch.manual_seed(2)
    xs = [torch.randn(4, 16, 5, 5) * (0.5 + i) for i in range(3)]
    out["xs"] = np.stack([x.numpy() for x in xs])
    qm = QuantizationManager(qmethod=QMethods.fp_quantizer.cls,
                             init=RangeEstimators.allminmax.cls, per_channel=False,
                             qparams=dict(n_bits=8, mantissa_bits=3, maxval=None, set_maxval=True))
    ys = [qm(xs[0]).numpy().copy(), qm(xs[1]).numpy().copy()]
    out["maxval_after2"] = qm.quantizer.maxval.numpy().copy()
    qm.fix_ranges()
    ys.append(qm(xs[2]).numpy().copy())
    out["maxval_after_fix"] = qm.quantizer.maxval.numpy().copy()
    out["ys"] = np.stack(ys)
    # set_maxval=False (CLI default): estimation is a no-op, default maxval is used
    qm = QuantizationManager(qmethod=QMethods.fp_quantizer.cls,
                             init=RangeEstimators.allminmax.cls, per_channel=False,
                             qparams=dict(n_bits=8, mantissa_bits=2, maxval=None, set_maxval=False))
    out["nomaxval_y"] = qm(xs[0] * 1e4).numpy().copy()
    out["nomaxval_maxval"] = qm.quantizer.maxval.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g6_manager.npz"), **out)
    print("g6 ok")


def tiny_cnn():
    """conv-bn-relu, conv(+bias)-relu6, residual-free tail: avgpool, flatten, fc  (<100 KB)."""
    torch.manual_seed(7)
    net = nn.Sequential(nn.Conv2d(3, 16, 3, padding=1, bias=False), nn.BatchNorm2d(16), nn.ReLU(),
                        nn.Conv2d(16, 24, 3, stride=2, padding=1, bias=True), nn.ReLU6(),
                        nn.Conv2d(24, 24, 3, padding=1, groups=24, bias=False), nn.BatchNorm2d(24), nn.ReLU(),
                        nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(24, 10))
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    return net.eval()


def make_g7():
    """Wrapper level: reference quantize_model on a tiny CNN with BASELINE config-3 settings
    (fp_quantizer E5M2, per-channel current_minmax weights, per-tensor allminmax activations,
    1 calibration batch, fix_ranges, then a validation batch)."""
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    out = {}
    net = tiny_cnn()
    for k, v in net.state_dict().items():
        out["sd_" + k] = v.numpy().copy()
    torch.manual_seed(8)
    calib = torch.randn(8, 3, 16, 16)
    val = torch.randn(8, 3, 16, 16) * 1.3
    out["calib"], out["val"] = calib.numpy(), val.numpy()
    for tag, M, act_est in (("e5m2", 2, "allminmax"), ("e4m3", 3, "allminmax"), ("e4m3_run", 3, "running_minmax")):
        qparams = dict(method=QMethods.fp_quantizer.cls, act_method=None,
                       weight_range_method=RangeEstimators.current_minmax.cls,
                       act_range_method=RangeEstimators[act_est].cls, n_bits=8, n_bits_act=8,
                       per_channel_weights=True, percentile=None, quantize_input=False,
                       fp8_kwargs=dict(maxval=None, mantissa_bits=M, set_maxval=True, learn_maxval=False,
                                       learn_mantissa_bits=False, mse_include_mantissa_bits=False,
                                       allow_unsigned=False))
        q = quantize_model(copy_net(net), tie_activation_quantizers=True, **qparams).eval()

        def each(fn):
            for m in q.modules():
                if isinstance(m, QuantizedModule):
                    fn(m)
        with torch.no_grad():
            out[f"{tag}_fp_logits"] = q(val).numpy().copy()          # quantizers off
            each(lambda m: m.quantized())
            out[f"{tag}_calib_logits"] = q(calib).numpy().copy()     # estimate_ranges state
            each(lambda m: m.fix_ranges())
            out[f"{tag}_val_logits"] = q(val).numpy().copy()
        mgrs = [(n, m) for n, m in q.named_modules() if isinstance(m, QuantizationManager)]
        out[f"{tag}_mgr_names"] = np.array([n for n, _ in mgrs])
        for n, m in mgrs:
            out[f"{tag}_maxval_{n}"] = m.quantizer.maxval.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g7_tinycnn.npz"), **out)
    print("g7 ok")


def wrapper_nets():
    """Two toy nets that go through every operator wrapper g7-g9 do not touch (autoquant_utils.py:20-31, 46-87, 94-105,
    120-122, 166-174): net1d = Conv1d+BN1d+ReLU (BNQConv1d), Conv1d+ReLU6 (QuantConv1d), ConvTranspose1d
    (QuantConvTranspose1d: weight dims 0/1 swapped around the per-channel quantizer), Flatten, Linear+BN1d+ReLU (BNQLinear),
    LayerNorm (QuantLayerNorm), Linear; net2d = Conv2d+ReLU, ConvTranspose2d+BN-free+ReLU (QuantConvTranspose), ConvTranspose2d
    with groups, AdaptiveAvgPool2d, Flatten, Linear.  Committed weights < 100 KB."""
    torch.manual_seed(11)
    net1d = nn.Sequential(nn.Conv1d(4, 8, 3, padding=1, bias=False), nn.BatchNorm1d(8), nn.ReLU(),
                          nn.Conv1d(8, 8, 3, padding=1, bias=True), nn.ReLU6(),
                          nn.ConvTranspose1d(8, 6, 4, stride=2, padding=1, bias=True),
                          nn.Flatten(), nn.Linear(6 * 32, 16, bias=False), nn.BatchNorm1d(16), nn.ReLU(),
                          nn.LayerNorm(16), nn.Linear(16, 5))
    net2d = nn.Sequential(nn.Conv2d(3, 6, 3, padding=1), nn.ReLU(),
                          nn.ConvTranspose2d(6, 10, 3, stride=2, padding=1, output_padding=1, bias=False), nn.ReLU(),
                          nn.ConvTranspose2d(10, 4, 2, stride=1, groups=2, bias=True),
                          nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.Linear(4, 7))
    with torch.no_grad():
        for net in (net1d, net2d):
            for m in net.modules():
                if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
                    m.running_mean.normal_(0, 0.2)
                    m.running_var.uniform_(0.5, 1.5)
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.normal_(0, 0.1)
                if isinstance(m, nn.LayerNorm):
                    m.weight.uniform_(0.5, 1.5)
                    m.bias.normal_(0, 0.1)
    return net1d.eval(), net2d.eval()


def make_g11():
    """The operator wrappers no other fixture reaches, through the reference's quantize_model with BASELINE config-3 settings
    (fp_quantizer E5M2 and E4M3, per-channel current_minmax weights, per-tensor allminmax activations) and, for the weights,
    per-TENSOR ranges as well (the transposed convolutions swap dims only when per_channel_weights is set)."""
    from quantization.autoquant_utils import quantize_model
    from quantization.base_quantized_classes import QuantizedModule
    out = {}
    nets = dict(zip(("net1d", "net2d"), wrapper_nets()))
    torch.manual_seed(12)
    data = dict(net1d=(torch.randn(8, 4, 16), torch.randn(8, 4, 16) * 1.3),
                net2d=(torch.randn(8, 3, 8, 8), torch.randn(8, 3, 8, 8) * 1.3))
    for name, net in nets.items():
        for k, v in net.state_dict().items():
            out[f"{name}_sd_{k}"] = v.numpy().copy()
        out[f"{name}_calib"], out[f"{name}_val"] = data[name][0].numpy(), data[name][1].numpy()
        for tag, M, pcw in (("e5m2", 2, True), ("e4m3", 3, True), ("e4m3_pt", 3, False)):
            qparams = dict(method=QMethods.fp_quantizer.cls, act_method=None,
                           weight_range_method=RangeEstimators.current_minmax.cls,
                           act_range_method=RangeEstimators.allminmax.cls, n_bits=8, n_bits_act=8,
                           per_channel_weights=pcw, percentile=None, quantize_input=False,
                           fp8_kwargs=dict(maxval=None, mantissa_bits=M, set_maxval=True, learn_maxval=False,
                                           learn_mantissa_bits=False, mse_include_mantissa_bits=False,
                                           allow_unsigned=False))
            q = quantize_model(copy_net(net), tie_activation_quantizers=True, **qparams).eval()
            out[f"{name}_{tag}_classes"] = np.array([type(m).__name__ for m in q])

            def each(fn):
                for m in q.modules():
                    if isinstance(m, QuantizedModule):
                        fn(m)
            calib, val = data[name]
            with torch.no_grad():
                out[f"{name}_{tag}_fp_logits"] = q(val).numpy().copy()
                each(lambda m: m.quantized())
                out[f"{name}_{tag}_calib_logits"] = q(calib).numpy().copy()
                each(lambda m: m.fix_ranges())
                out[f"{name}_{tag}_val_logits"] = q(val).numpy().copy()
                # every layer's quantized weight (what run_forward receives), in the layer's own layout
                for n, m in q.named_modules():
                    if hasattr(m, "weight_quantizer") and hasattr(m, "get_params"):
                        out[f"{name}_{tag}_wq_{n}"] = m.get_params()[0].numpy().copy()
            mgrs = [(n, m) for n, m in q.named_modules() if isinstance(m, QuantizationManager)]
            out[f"{name}_{tag}_mgr_names"] = np.array([n for n, _ in mgrs])
            for n, m in mgrs:
                out[f"{name}_{tag}_maxval_{n}"] = m.quantizer.maxval.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "g11_wrappers.npz"), **out)
    print("g11 ok:", len(out), "arrays")


def edge_inputs_f64(M, maxval, sign_bits, n_bits=8):
    """float64 twin of edge_inputs: ties and binade borders placed with double precision."""
    E = n_bits - sign_bits - M
    mv = float(np.float32(maxval))
    bias = float(np.float32(np.float32(np.float32(2.0 ** E - np.float32(np.log2(np.float32(mv)))) +
                                       np.float32(np.log2(2 - 2.0 ** (-M)))) - np.float32(1)))
    vals = [0.0, -0.0, 1e-30, -1e-30, 1e-310, -4e-320, 1e-42, mv, -mv, mv * 1.5, -mv * 1.5, float("inf"), float("-inf"),
            float("nan"), float(np.nextafter(mv, 0.0)), float(np.nextafter(mv, np.inf)), 1.0, 0.5, 2.0, -0.25]
    pmax = int(2 ** E)
    for p in sorted(set([1, 2, 3, max(pmax // 2, 1), max(pmax - 2, 1), max(pmax - 1, 1), pmax])):
        s = 2.0 ** (p - M - bias)
        lo = 2.0 ** (p - bias)
        for k in (0, 1, 2, 2 ** M - 1, 2 ** M, 2 ** M + 1, 2 ** (M + 1) - 2, 2 ** (M + 1) - 1):
            t = (k + 0.5) * s
            vals += [t, float(np.nextafter(t, 0.0)), float(np.nextafter(t, np.inf)), -t]
        vals += [lo, float(np.nextafter(lo, 0.0)), float(np.nextafter(lo, np.inf)), -lo, lo * (1 + 2.0 ** -30),
                 lo * (1 - 2.0 ** -30)]
    return np.array(vals, dtype=np.float64)


def make_g1c():
    """quantize_to_fp8_ste_MM on float64 tensors (BASELINE config 1's dtype): maxval / mantissa bits float32 tensors,
    so bias is float32 and everything downstream of x is float64 (fp8_quantizer.py:105-133 under type promotion)."""
    out = {}
    rng = np.random.RandomState(4321)
    base = rng.randn(1024)
    cases = []
    cid = 0
    for M in (1, 2, 3, 4, 5, 6):
        for mv in (default_maxval(M), 2.5, 0.7361, 18.583):
            for sb in (1, 0):
                if sb == 0 and mv != 0.7361:
                    continue
                x = np.concatenate([base * (mv / 2.3), edge_inputs_f64(M, mv, sb)])
                y = quantize_to_fp8_ste_MM(torch.from_numpy(x.copy()), 8, torch.Tensor([mv]), torch.Tensor([float(M)]), sb)
                assert y.dtype == torch.float64
                out[f"c{cid}_x"] = x
                out[f"c{cid}_y"] = y.numpy()
                cases.append((cid, M, float(np.float32(mv)), sb, 1))
                cid += 1
    for M in (2, 3, 5):     # per-channel maxval [8], x [8, 300]
        mvs = (np.abs(rng.randn(8)) * 2 + 0.05).astype(np.float32)
        x = rng.randn(8, 300) * (mvs[:, None].astype(np.float64) / 2.0)
        y = quantize_to_fp8_ste_MM(torch.from_numpy(x.copy()), 8, torch.from_numpy(mvs.copy()), torch.Tensor([float(M)]), 1)
        out[f"c{cid}_x"] = x
        out[f"c{cid}_y"] = y.numpy()
        out[f"c{cid}_maxval"] = mvs
        cases.append((cid, M, -1.0, 1, 8))
        cid += 1
    mvs = np.array([1.0, 0.0, 2.5, np.inf, np.nan, 1e-40], dtype=np.float32)    # degenerate channels
    x = rng.randn(6, 64)
    x[1] = 0
    y = quantize_to_fp8_ste_MM(torch.from_numpy(x.copy()), 8, torch.from_numpy(mvs.copy()), torch.Tensor([3.0]), 1)
    out[f"c{cid}_x"] = x
    out[f"c{cid}_y"] = y.numpy()
    out[f"c{cid}_maxval"] = mvs
    cases.append((cid, 3, -1.0, 1, 6))
    cid += 1
    out["cases"] = np.array(cases, dtype=np.float64)  # id, mbits, maxval(-1: per-channel), sign, n_maxval
    np.savez_compressed(os.path.join(OUT, "g1c_quantize_f64.npz"), **out)
    print("g1c:", cid, "cases")


def copy_net(net):
    import copy
    return copy.deepcopy(net)


def make_g5():
    """BASELINE config 1 (fast variant): compute_quant_error.py with 200 000 samples, seed 10.
    Per distribution and format: line-search range, analytic quantization MSE, analytic dot-product
    MSE -- plus a few raw values of the closed-form interval integrals (utils/distributions.py)."""
    from utils.distributions import ClippedGaussDistr, UniformDistr, ClippedStudentTDistr
    from quantization.quant_error_estimator import compute_expected_quant_mse, compute_expected_dot_prod_mse
    from quantization.range_estimators import LineSearchEstimator
    from quantization.quantizers.uniform_quantizers import SymmetricUniformQuantizer
    from utils import seed_all
    distrs = {"uniform": UniformDistr(range_min=-1.0, range_max=1.0, params_dict={}),
              "gauss": ClippedGaussDistr(params_dict={"mu": 0.0, "sigma": 1.0}, range_min=-10.0, range_max=10.0),
              "student": ClippedStudentTDistr(params_dict={"nu": 8.0}, range_min=-100.0, range_max=100.0)}
    out = {}
    n = 200000
    for name, d in distrs.items():
        seed_all(10)
        sample = torch.tensor(d.sample((n,)))
        out[f"{name}_sample_head"] = sample[:64].numpy().copy()
        out[f"{name}_sample_sum"] = np.array(float(sample.sum()))
        rows = []
        for exp_bits in (5, 4, 3, 2, 0):
            M = 7 - exp_bits
            q = FPQuantizer(n_bits=8, mantissa_bits=M, set_maxval=True) if exp_bits > 0 \
                else SymmetricUniformQuantizer(n_bits=8)
            est = LineSearchEstimator(quantizer=q)          # what estimate_range_line_search builds (:372-379)
            rmin, rmax = est.forward(sample)
            out[f"{name}_loss_{exp_bits}"] = np.asarray(est.loss_array, np.float64).copy()     # [1, 1001], [:, 0] = inf
            out[f"{name}_search_{exp_bits}"] = np.array([est.max_pos_thr, est.max_search_range, est.step_size,
                                                         float(est.one_sided_dist)])
            mse = compute_expected_quant_mse(d, q, rmin, rmax, n)
            dp = compute_expected_dot_prod_mse(d, d, q, q, rmin, rmax, rmin, rmax)
            rows.append((exp_bits, float(rmin), float(rmax), float(mse), float(dp)))
            print(name, rows[-1])
        out[f"{name}_rows"] = np.array(rows)
        # raw integrals on a few intervals
        ab = [(-0.7, -0.2, -0.5), (0.0, 0.3, 0.25), (0.1, 0.9, 1.0), (-1.0, 1.0, 0.0)]
        if name != "uniform":
            ab += [(2.0, 7.5, 3.0), (-9.0, -0.5, -4.0)]
        out[f"{name}_ab"] = np.array(ab)
        out[f"{name}_p_sqr_r"] = np.array([d.integr_interv_p_sqr_r(a, b, u) for a, b, u in ab])
        out[f"{name}_x_p_signed_r"] = np.array([d.integr_interv_x_p_signed_r(a, b, u) for a, b, u in ab])
        out[f"{name}_second_moment"] = np.array(d.eval_non_central_second_moment())
    np.savez_compressed(os.path.join(OUT, "g5_quant_error.npz"), **out)
    print("g5 ok")


def _load_own(name, relpath):
    """import one of THIS repo's fp32 model definitions by path (the reference's `models` package
    owns the name `models` in this process)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(os.path.dirname(OUT), "..",
                                                                     "fp8-quantization_amd", relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def warm_bn(model, seed=2):
    """Give the seeded random-init network sane BN statistics (one train-mode batch, momentum 1):
    without it activations vanish by ~10x per block and deep-layer MSE searches are decided by fp32
    noise.  tests/test_models.py applies the identical procedure."""
    torch.manual_seed(seed)
    x = torch.randn(8, 3, 64, 64)
    bns = [m for m in model.modules() if isinstance(m, nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    model.train()
    with torch.no_grad():
        model(x)
    model.eval()
    for m in bns:
        m.momentum = 0.1
    return model


def _model_golden(tag, qmodel, calib, val, out):
    from quantization.base_quantized_classes import QuantizedModule
    qmodel.eval()
    with torch.no_grad():
        out[f"{tag}_fp_logits"] = qmodel(val).numpy().copy()
        qmodel.set_quant_state(True, True)
        out[f"{tag}_calib_logits"] = qmodel(calib).numpy().copy()
        qmodel.fix_ranges()
        out[f"{tag}_val_logits"] = qmodel(val).numpy().copy()
    seen, names = set(), []
    for n, m in qmodel.named_modules():
        if isinstance(m, QuantizationManager) and id(m) not in seen:
            seen.add(id(m))
            names.append(n)
            out[f"{tag}_maxval_{n}"] = m.quantizer.maxval.numpy().copy()
            out[f"{tag}_mbits_{n}"] = np.array(float(m.quantizer.mantissa_bits))
    out[f"{tag}_mgr_names"] = np.array(names)
    out[f"{tag}_state_keys"] = np.array(list(qmodel.state_dict().keys()))


def _qparams(M, w_est, a_est, incl=False):
    return dict(method=QMethods.fp_quantizer.cls, act_method=None,
                weight_range_method=RangeEstimators[w_est].cls, act_range_method=RangeEstimators[a_est].cls,
                n_bits=8, n_bits_act=8, per_channel_weights=True, percentile=None, quantize_input=False,
                weight_range_options={}, act_range_options={},
                fp8_kwargs=dict(maxval=None, mantissa_bits=M, set_maxval=True, learn_maxval=False,
                                learn_mantissa_bits=False, mse_include_mantissa_bits=incl, allow_unsigned=False))


def make_g8():
    """BASELINE config 3 (reduced): reference QuantizedResNet on a seeded ResNet-18, 64x64 inputs,
    E5M2, per-channel current_minmax weights, per-tensor allminmax activations."""
    own = _load_own("amd_resnet", "models/resnet.py")
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvr = types.ModuleType("torchvision.models.resnet")
    tvr.BasicBlock, tvr.Bottleneck = own.BasicBlock, own.Bottleneck
    tvm.resnet18 = own.resnet18
    tvm.resnet50 = lambda *a, **k: (_ for _ in ()).throw(NotImplementedError())
    tvm.resnet = tvr
    tv.models = tvm
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.resnet": tvr})
    from models.resnet_quantized import QuantizedResNet      # the REFERENCE's class
    torch.manual_seed(0)
    fp = warm_bn(own.resnet18())
    torch.manual_seed(1)
    calib, val = torch.randn(4, 3, 64, 64), torch.randn(4, 3, 64, 64)
    out = {}
    q = QuantizedResNet(fp, input_size=(1, 3, 64, 64), **_qparams(2, "current_minmax", "allminmax"))
    _model_golden("r18", q, calib, val, out)
    np.savez_compressed(os.path.join(OUT, "g8_resnet18.npz"), **out)
    print("g8 ok", len(out["r18_mgr_names"]), "managers")


def make_g9():
    """BASELINE config 4 (reduced): reference QuantizedMobileNetV2, E4M3, MSE range estimator for
    weights and activations (--no-fp8-mse-include-mantissa-bits), 64x64 inputs."""
    own = _load_own("amd_mbv2", "models/mobilenet_v2.py")
    from models.mobilenet_v2 import MobileNetV2 as RefMobileNetV2   # reference fp32 definition
    from models.mobilenet_v2_quantized import QuantizedMobileNetV2
    torch.manual_seed(0)
    mine = warm_bn(own.MobileNetV2(input_size=64))
    ref_fp = RefMobileNetV2(input_size=64).eval()
    ref_fp.load_state_dict(mine.state_dict())     # also proves checkpoint-key compatibility
    torch.manual_seed(1)
    calib, val = torch.randn(4, 3, 64, 64), torch.randn(4, 3, 64, 64)
    out = {}
    q = QuantizedMobileNetV2(ref_fp, input_size=(1, 3, 64, 64), **_qparams(3, "MSE", "MSE"))
    _model_golden("mbv2", q, calib, val, out)
    np.savez_compressed(os.path.join(OUT, "g9_mobilenetv2.npz"), **out)
    print("g9 ok", len(out["mbv2_mgr_names"]), "managers")


def bulk_cases():
    """(name, seed, M, shape, maxval) of the bulk flip-rate fixture g1b; inputs are regenerated from the seed by the
    tests (numpy's legacy RandomState stream is stable across versions): x = standard_normal(shape) as fp32."""
    cases = []
    for M, tag in ((2, "e5m2"), (3, "e4m3")):
        cases.append((f"{tag}_tensor", 4100 + M, M, (1 << 22,), np.array([2.7361], np.float32)))
        rng = np.random.RandomState(4200 + M)
        cases.append((f"{tag}_channel", 4300 + M, M, (4096, 1024),
                      (np.abs(rng.standard_normal(4096)) * 2 + 0.05).astype(np.float32)))
    return cases


def bulk_input(seed, shape):
    return np.random.RandomState(seed).standard_normal(int(np.prod(shape))).astype(np.float32).reshape(shape)


def ulp_key(a):
    """monotone integer image of fp32 values (distance = ULP distance)"""
    i = a.view(np.int32).astype(np.int64)
    return np.where(i < 0, np.int64(-2147483648) - i, i)


def make_g1b():
    """4 M seeded normals x {E5M2, E4M3} x {per-tensor arbitrary maxval, per-channel} through the reference's
    quantize_to_fp8_ste_MM.  Stored: the SHA-256 of the reference's output bytes and the sparse difference between
    that output and the C oracle's (index deltas + ULP deltas), so the tests can rebuild the reference's output bit
    for bit from the oracle's and pin the oracle's flip rate on bulk data -- in a few hundred KB instead of 64 MB."""
    import hashlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    import oracle
    out = {}
    names = []
    for name, seed, M, shape, mv in bulk_cases():
        x = bulk_input(seed, shape)
        with torch.no_grad():
            y_ref = quantize_to_fp8_ste_MM(torch.from_numpy(x.copy()), 8, torch.from_numpy(mv.copy()),
                                           torch.Tensor([float(M)]), 1).numpy()
        y_orc = oracle.c_quantize(x, mv, M, 8, 1)
        assert not np.isnan(y_ref).any() and not np.isnan(y_orc).any()
        d = (ulp_key(y_ref.reshape(-1)) - ulp_key(y_orc.reshape(-1)))
        idx = np.flatnonzero(d)
        small = np.abs(d[idx]) <= 100
        out[f"{name}_sha256"] = np.frombuffer(hashlib.sha256(y_ref.tobytes()).digest(), np.uint8)
        out[f"{name}_idx_delta"] = np.diff(idx[small], prepend=0).astype(np.uint32)
        out[f"{name}_ulp_delta"] = d[idx[small]].astype(np.int8)
        out[f"{name}_big_idx"] = idx[~small].astype(np.int64)            # grid-step flips: stored as values
        out[f"{name}_big_val"] = y_ref.reshape(-1)[idx[~small]]
        names.append(name)
        print(f"g1b {name}: {x.size} elements, {idx.size} differ from the C oracle ({idx.size / x.size:.3%}), "
              f"{int((~small).sum())} by more than 100 ULP")
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "g1b_bulk.npz"), **out)


def make_g10():
    """Autograd of quantize_to_fp8_ste_MM (fp8_quantizer.py:105-133: STE rounding, clamp through torch.min/max, the
    scale's dependence on maxval through the bias): d/dx and d/dmaxval for a fixed upstream gradient."""
    out, cases = {}, []
    rng = np.random.RandomState(77)
    cid = 0
    for M in (2, 3):
        for sb in (1, 0):
            for per_channel in (False, True):
                mv = (np.abs(rng.randn(6)) + 0.3).astype(np.float32) if per_channel else np.array([1.3], np.float32)
                x = (rng.randn(6, 50) * 0.8).astype(np.float32)
                x[:, 0] = mv if per_channel else mv[0]            # exactly at the clamp bounds (tie in torch.min)
                x[:, 1] = -(mv if per_channel else mv[0])
                x[:, 2] = 0.0
                g = rng.randn(6, 50).astype(np.float32)
                xt = torch.from_numpy(x.copy()).requires_grad_(True)
                mt = torch.from_numpy(mv.copy()).requires_grad_(True)
                bt = torch.Tensor([float(M)]).requires_grad_(True)       # learn_mantissa_bits: the width as a Parameter (:105-110)
                y = quantize_to_fp8_ste_MM(xt, 8, mt, bt, sb)
                y.backward(torch.from_numpy(g))
                out[f"c{cid}_x"], out[f"c{cid}_maxval"], out[f"c{cid}_g"] = x, mv, g
                out[f"c{cid}_y"] = y.detach().numpy()
                out[f"c{cid}_gx"], out[f"c{cid}_gmaxval"] = xt.grad.numpy(), mt.grad.numpy()
                out[f"c{cid}_gmbits"] = bt.grad.numpy()
                cases.append((cid, M, sb, int(per_channel)))
                cid += 1
    # non-integer and out-of-range widths: round_ste passes the gradient, clamp(1, n_bits - sign_bits) cuts it off outside
    mb_cases = []
    x = (rng.randn(4, 40) * 0.8).astype(np.float32)
    g = rng.randn(4, 40).astype(np.float32)
    for k, mb in enumerate((2.4, 3.5, 0.2, 0.6, 7.0, 7.4, 9.0)):
        bt = torch.Tensor([mb]).requires_grad_(True)
        y = quantize_to_fp8_ste_MM(torch.from_numpy(x.copy()), 8, torch.Tensor([1.3]), bt, 1)
        y.backward(torch.from_numpy(g))
        mb_cases.append((mb, float(bt.grad[0])))
        out[f"mb{k}_y"] = y.detach().numpy()
    out["mb_x"], out["mb_g"], out["mb_cases"] = x, g, np.array(mb_cases, dtype=np.float64)
    out["cases"] = np.array(cases, dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "g10_autograd.npz"), **out)
    print("g10:", cid, "cases")


def make_g3b():
    """rows that hold both -0.0 and +0.0 (and all-zero rows of either sign): the reference's min / max, the maxval its
    quantizer derives from them (fp8_quantizer.py:236) and the quantized rows.  ATen's min / max of mixed zeros depends on
    the order of the elements; the build's contract is IEEE 754-2019 minimum / maximum (-0 < +0): the fixture pins the
    VALUES everywhere, the maxval bit for bit, and records where the reference's signs agree with that contract."""
    nz, pz = np.float32(-0.0), np.float32(0.0)
    rows = np.array([[nz, pz, nz, pz, nz, pz, nz, pz],
                     [pz, nz, pz, nz, pz, nz, pz, nz],
                     [nz] * 8,
                     [pz] * 8,
                     [pz, nz, 1e-3, nz, pz, 1e-3, nz, pz],
                     [nz, pz, -1e-3, pz, nz, -1e-3, pz, nz],
                     [nz, nz, nz, nz, pz, pz, pz, pz],
                     [pz, pz, pz, pz, nz, nz, nz, nz]], dtype=np.float32)
    x = torch.from_numpy(rows.copy())
    out = {"x": rows}
    for pc in (True, False):
        est = RangeEstimators.current_minmax.cls(per_channel=pc)
        mn, mx = est(x)
        out[f"pc{int(pc)}_min"], out[f"pc{int(pc)}_max"] = mn.numpy().reshape(-1), mx.numpy().reshape(-1)
        est = RangeEstimators.allminmax.cls(per_channel=pc)
        est(x)
        mn, mx = est(torch.flip(x, dims=[1]))         # second batch: the same values in the opposite order
        out[f"all_pc{int(pc)}_min"], out[f"all_pc{int(pc)}_max"] = mn.numpy().reshape(-1), mx.numpy().reshape(-1)
    q = FPQuantizer(n_bits=8, per_channel=True, mantissa_bits=3, maxval=None, set_maxval=True)
    q.set_quant_range(out_t(out["pc1_min"]), out_t(out["pc1_max"]))
    out["maxval"] = q.maxval.numpy()
    with np.errstate(all="ignore"):
        out["q"] = q(x).numpy()                       # all-zero rows: maxval 0 -> NaN rows (a reproduced quirk)
    np.savez_compressed(os.path.join(OUT, "g3b_signed_zero.npz"), **out)
    print("g3b ok")


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:          # e.g.  python -B make_golden.py g1b g10
        for name in only:
            globals()["make_" + name]()
        assert not os.path.exists(os.path.join(REF, "quantization", "__pycache__")), "pycache leaked"
        sys.exit(0)
    make_g1b()
    make_g10()
    make_g1()
    make_g1c()
    make_g2()
    make_g3()
    make_g3b()
    make_g4()
    make_g4c()
    make_g5()
    make_g6()
    make_g7()
    make_g8()
    make_g9()
    make_g11()
    make_g12()
    assert not os.path.exists(os.path.join(REF, "quantization", "__pycache__")), "pycache leaked"
