"""Parity at the sizes BASELINE.json states (the small-shape tests live in test_hip_parity / test_models).

* config 5: per-rank slab [512, 4096, 512] fp32 -- allminmax fold -> range exchange -> E4M3 quantize
  (fp8q.dist.calibrate_quantize_sharded), one rank and two gloo ranks sharing the one GPU of the test box;
* configs 3 / 4: ResNet-18 (E5M2, per-channel current_minmax + per-tensor allminmax) and MobileNetV2 (E4M3, MSE
  estimator) at batch 64 x 224 x 224 with a hook on every fp8q.ops launch of the calibration pass, of fix_ranges()
  and of the validation pass: each launch's output is compared with the CPU oracle evaluated on the very tensor that
  launch read.  MIOpen's convolutions sit between the quantizers and round differently from the CPU's, so logits
  cannot be compared bit for bit with a CPU run -- but every quantizer launch can, on its own input.

Reference flow: quantization_manager.py:114-122 (estimate -> set_quant_range -> quantize),
quantized_folded_bn.py:30-56 (batch norm + activation in front of the activation quantizer),
range_estimators.py:83-100 / :318-369, fp8_quantizer.py:91-133.
"""
import inspect
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _np(t):
    return t.detach().cpu().numpy()


def _bits_equal(got, ref, what):
    got, ref = np.ascontiguousarray(got, np.float32), np.ascontiguousarray(ref, np.float32)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    a, b = got.reshape(-1).view(np.int32), ref.reshape(-1).view(np.int32)
    if np.array_equal(a, b):
        return
    nan_a, nan_b = np.isnan(got.reshape(-1)), np.isnan(ref.reshape(-1))
    assert np.array_equal(nan_a, nan_b), f"{what}: NaN pattern differs"
    bad = np.flatnonzero((a != b) & ~nan_a)
    assert bad.size == 0, f"{what}: {bad.size} of {a.size} elements differ, first at {bad[:5]}: " \
                          f"{got.reshape(-1)[bad[:5]]} vs {ref.reshape(-1)[bad[:5]]}"


# ---------------------------------------------------------------------------------------------------------------------
# config 5
# ---------------------------------------------------------------------------------------------------------------------
SLAB = (512, 4096, 512)
N_WIN, WIN = 64, 1 << 16


def _slab(seed, shape=SLAB):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.empty(*shape, device="cuda")
    for i in range(0, shape[0], 64):
        x[i:i + 64].normal_(generator=g)
    return x


def _check_windows(x, y, maxval, mbits, seed, what):
    """y == oracle(x) bit for bit on N_WIN random windows of WIN elements (plus the first and the last window)."""
    n = x.numel()
    rng = np.random.RandomState(seed)
    starts = [0, n - WIN] + [int(s) for s in rng.randint(0, n - WIN, N_WIN - 2)]
    xf, yf = x.view(-1), y.view(-1)
    idx = torch.tensor(starts, device=x.device).view(-1, 1) + torch.arange(WIN, device=x.device).view(1, -1)
    xs, ys = _np(xf[idx]), _np(yf[idx])
    ref = oracle.c_quantize(xs.reshape(-1), np.asarray(maxval, np.float32).reshape(1), mbits, 8, 1)
    _bits_equal(ys.reshape(-1), ref, what)


def test_config5_slab_fold_exchange_quantize():
    """Two sequential slabs through calibrate_quantize_sharded (world 1): the running range after each batch is the
    oracle's min/max fold of the slabs, the quantized slab is the oracle's E4M3 result with that range on 64 windows
    of 64 Ki elements, and the whole flow equals the one-shot path (min/max over both slabs at once, direct quantize)."""
    from fp8q import dist as fd, ops
    state = None
    ref_mn = ref_mx = None
    slabs = []
    for b in range(2):
        x = _slab(1234 + b)
        y, state = fd.calibrate_quantize_sharded(x, 3, 8, 1, state=state)
        mn, mx = oracle.c_minmax(_np(x).reshape(-1), False)
        if ref_mn is None:
            ref_mn, ref_mx = mn, mx
        else:
            ref_mn, ref_mx = oracle.c_fold(ref_mn, ref_mx, mn, mx, 1)
        _bits_equal(_np(state[0]), ref_mn, f"batch {b}: running min")
        _bits_equal(_np(state[1]), ref_mx, f"batch {b}: running max")
        mv = oracle.c_absmax(ref_mn, ref_mx)
        _check_windows(x, y, mv, 3, 100 + b, f"batch {b}: E4M3 slab")
        # the same slab through the plain quantizer with the same range: identical everywhere, on the device
        y2 = ops.quantize(x, torch.from_numpy(mv).cuda(), 3, 8, 1)
        assert torch.equal(y.view(-1).view(torch.int32), y2.view(-1).view(torch.int32))
        del y, y2
        slabs.append(x)
    both = torch.cat(slabs)                       # 8.6 GB: one-shot estimate over the union of the batches
    del slabs
    mn1, mx1 = ops.minmax(both, False)
    assert torch.equal(mn1, state[0]) and torch.equal(mx1, state[1])


def test_config5_two_gloo_ranks_on_one_gpu():
    """The same flow batch-sharded over two ranks (gloo, both on cuda:0, half a slab each): every rank ends with the
    global range = the oracle's min/max over both half-slabs and quantizes its half with it, bit for bit."""
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(HERE, "c5_rank_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    oks = [ln for ln in r.stdout.splitlines() if ln.startswith("C5_RANK_OK")]
    assert len(oks) == 2, r.stdout[-2000:]
    # both ranks report the same global range
    assert len({ln.split("range=")[1] for ln in oks}) == 1, oks


# ---------------------------------------------------------------------------------------------------------------------
# configs 3 / 4: every launch of a full-size model pass against the oracle
# ---------------------------------------------------------------------------------------------------------------------
MSE_FULL_ORACLE_MAX = 1 << 22     # K4 on activations: the oracle walks 111 candidates per element
MSE_SLICE = 1 << 20


class LaunchChecker:
    """Wraps fp8q.ops.*: runs the real launch, then recomputes it with the oracle from the launch's own inputs."""

    NAMES = ("quantize", "minmax", "minmax_quantize", "affine_act_quantize", "affine_act_minmax", "mse_grid",
             "multi_quantize")

    def __init__(self, ops, monkeypatch):
        self.ops = ops
        self.count = {n: 0 for n in self.NAMES}
        self.elements = 0
        self.choice = dict(quantizers=0, channels=0, channels_with_another_candidate=0, oracle_candidates_evaluated=0)
        self.real = {n: getattr(ops, n) for n in self.NAMES}
        for n in self.NAMES:
            monkeypatch.setattr(ops, n, self._wrap(n))
        self.count["plan_launch"] = 0
        checker = self

        class CheckedPlan(ops.MultiPlan):      # the prepared multi-tensor launch fix_ranges() builds
            def __init__(self, items):
                self._items = [tuple(it) for it in items]
                super().__init__(self._items)

            def launch(self):
                outs = super().launch()
                checker.count["plan_launch"] += 1
                checker._check_multi_quantize({"items": self._items}, outs, None)
                return outs
        monkeypatch.setattr(ops, "MultiPlan", CheckedPlan)
        # the one-call MSE calibration step (fp8q_mse_calibrate_f32): epilogue (optional), abs-max + grid, table, selection and
        # the quantized batch -- each recomputed by the oracle from the step's own input
        self.count["mse_calibrate"] = 0
        real_step = ops.MseCalibration.step

        def step(cal, x, quantize=True, pre=None):
            first = cal.first
            before = None if first else _np(cal.mses).copy()
            y = real_step(cal, x, quantize, pre)
            checker.count["mse_calibrate"] += 1
            checker._check_mse_calibrate(cal, x, pre, first, before, y)
            return y
        monkeypatch.setattr(ops.MseCalibration, "step", step)

    def _wrap(self, name):
        real, sig = self.real[name], inspect.signature(self.real[name])
        check = getattr(self, "_check_" + name)

        def wrapper(*a, **k):
            b = sig.bind(*a, **k)
            b.apply_defaults()
            args = b.arguments
            pre = None
            if name in ("minmax", "affine_act_minmax") and args["cur_min"] is not None:
                pre = (_np(args["cur_min"]).copy(), _np(args["cur_max"]).copy())
            if name == "mse_grid":
                pre = _np(args["mses"]).copy()
            out = real(*a, **k)
            self.count[name] += 1
            check(args, out, pre)
            return out
        return wrapper

    # -- one checker per entry point -----------------------------------------------------------------------------
    def _check_quantize(self, a, out, pre):
        x = _np(a["x"])
        ref = oracle.c_quantize(x, _np(a["maxval"]).reshape(-1), a["mbits"], a["n_bits"], a["sign_bits"])
        _bits_equal(_np(out), ref, f"quantize {tuple(x.shape)}")
        self.elements += x.size

    def _fold(self, mn, mx, pre, a):
        if pre is not None and a["mode"] != 0:
            mn, mx = oracle.c_fold(pre[0].reshape(-1), pre[1].reshape(-1), mn, mx, a["mode"], a["momentum"])
        return mn, mx

    def _check_minmax(self, a, out, pre):
        x = _np(a["x"])
        mn, mx = self._fold(*oracle.c_minmax(x, a["per_channel"]), pre, a)
        _bits_equal(_np(out[0]).reshape(-1), mn, f"minmax {tuple(x.shape)} min")
        _bits_equal(_np(out[1]).reshape(-1), mx, f"minmax {tuple(x.shape)} max")
        if a["want_maxval"]:
            _bits_equal(_np(out[2]).reshape(-1), oracle.c_absmax(mn, mx), f"minmax {tuple(x.shape)} maxval")
        self.elements += x.size

    def _check_minmax_quantize(self, a, out, pre):
        x = _np(a["x"])
        mn, mx = oracle.c_minmax(x, True)
        mv = oracle.c_absmax(mn, mx)
        _bits_equal(_np(out[1]), mn, f"minmax_quantize {tuple(x.shape)} min")
        _bits_equal(_np(out[2]), mx, f"minmax_quantize {tuple(x.shape)} max")
        _bits_equal(_np(out[3]), mv, f"minmax_quantize {tuple(x.shape)} maxval")
        _bits_equal(_np(out[0]), oracle.c_quantize(x, mv, a["mbits"], a["n_bits"], a["sign_bits"]),
                    f"minmax_quantize {tuple(x.shape)}")
        self.elements += x.size

    @staticmethod
    def _pre_activation(a):
        bn = tuple(_np(t) for t in a["bn"]) if a["bn"] is not None else None
        res = _np(a["residual"]) if a["residual"] is not None else None
        return oracle.c_affine_act(_np(a["x"]), bn, res, a["act"])

    def _check_affine_act_quantize(self, a, out, pre):
        t = self._pre_activation(a)
        ref = oracle.c_quantize(t, _np(a["maxval"]).reshape(-1), a["mbits"], a["n_bits"], a["sign_bits"])
        _bits_equal(_np(out), ref, f"affine_act_quantize {tuple(t.shape)} bn={a['bn'] is not None} "
                                   f"res={a['residual'] is not None} act={a['act']}")
        self.elements += t.size

    def _check_affine_act_minmax(self, a, out, pre):
        t = self._pre_activation(a)
        mn, mx = self._fold(*oracle.c_minmax(t, False), pre, a)
        _bits_equal(_np(out[0]).reshape(-1), mn, f"affine_act_minmax {tuple(t.shape)} min")
        _bits_equal(_np(out[1]).reshape(-1), mx, f"affine_act_minmax {tuple(t.shape)} max")
        _bits_equal(_np(out[2]).reshape(-1), oracle.c_absmax(mn, mx), f"affine_act_minmax {tuple(t.shape)} maxval")
        self.elements += t.size

    def _check_multi_quantize(self, a, out, pre):
        for it, y in zip(a["items"], out):
            x, mv, mbits = it[0], it[1], it[2]
            n_bits = it[3] if len(it) > 3 else 8
            sign_bits = it[4] if len(it) > 4 else 1
            _bits_equal(_np(y), oracle.c_quantize(_np(x), _np(mv).reshape(-1), mbits, n_bits, sign_bits),
                        f"multi_quantize {tuple(x.shape)}")
            self.elements += x.numel()

    @staticmethod
    def _choice(table):
        """FP_MSE_Estimator's decision from a [n_m, n_cand, C] table (range_estimators.py:350-369): plurality vote of
        the per-channel best mantissa width (torch.mode: the smallest of the most frequent), then the per-channel
        argmin over the candidates of that width."""
        best_m_per_ch = table.min(1).argmin(0)
        m = int(np.bincount(best_m_per_ch, minlength=table.shape[0]).argmax())
        return m, table[m].argmin(0)

    def _check_choice(self, a, got, ref, what):
        """SURVEY.md 8c for K4: the CHOSEN (mantissa bits, maxval[C]) equals the choice from the oracle's table, or the
        oracle's MSE at the chosen candidate is within 1e-6 relative of the oracle's minimum.  `ref` = the oracle's
        full table (per-channel weights, small activations) or None (long per-tensor rows): then the oracle is
        evaluated, on the whole tensor, at every candidate whose device MSE is within 1e-4 of the device minimum --
        the device table agrees with the oracle's to 1e-5 everywhere, so the oracle's argmin is one of those."""
        st = self.choice
        m_got, arg_got = self._choice(got)
        if ref is not None:
            m_ref, arg_ref = self._choice(ref)
            assert m_got == m_ref, (what, "mantissa vote", m_got, m_ref)
            cols = np.arange(got.shape[2])
            chosen, best = ref[m_ref][arg_got, cols], ref[m_ref][arg_ref, cols]
        else:
            x, grid, mb = a["x"].contiguous(), _np(a["grid"]), list(a["mbits_list"])
            near = np.argwhere(got[:, :, 0] <= got.min() * (1 + 1e-4))       # (m, candidate) pairs, C == 1
            ms, cs = sorted(set(int(v) for v in near[:, 0])), sorted(set(int(v) for v in near[:, 1]))
            sub = oracle.c_mse_grid(_np(x), False, np.ascontiguousarray(grid[cs]), [mb[m] for m in ms],
                                    n_bits=a["n_bits"], sign_bits=a["sign_bits"])
            pairs = {(m, c): sub[ms.index(m), cs.index(c), 0] for m, c in near.tolist()}
            (m_ref, c_ref), best = min(pairs.items(), key=lambda kv: (kv[1], kv[0]))
            assert m_got == m_ref or pairs[(m_got, int(arg_got[0]))] <= best * (1 + 1e-6), (what, pairs, m_got, arg_got)
            chosen, best = np.array([pairs[(m_got, int(arg_got[0]))]]), np.array([best])
            arg_ref = np.array([c_ref if m_ref == m_got else -1])
            st["oracle_candidates_evaluated"] += len(pairs)
        differ = arg_got != arg_ref
        ok = chosen <= best * (1 + 1e-6)
        assert ok.all(), (what, "chosen candidate's oracle MSE not within 1e-6 of the oracle's minimum",
                          np.argwhere(~ok)[:4].tolist(), chosen[~ok][:4], best[~ok][:4])
        st["quantizers"] += 1
        st["channels"] += int(differ.size)
        st["channels_with_another_candidate"] += int(differ.sum())

    def _check_mse_calibrate(self, cal, x, pre, first, before, y):
        assert first and before is None, "one calibration batch per estimator in this test"
        per_channel = cal.C != 1
        if pre is not None:
            ab, res, act = pre
            bn = None
            if ab is not None:      # the folded vector {alpha, beta'}: mean 0, invstd alpha, gamma 1, beta beta' reproduce it exactly
                abn = _np(ab)
                bn = (np.zeros(abn.shape[0], np.float32), abn[:, 0].copy(), np.ones(abn.shape[0], np.float32), abn[:, 1].copy())
            t_np = oracle.c_affine_act(_np(x), bn, _np(res) if res is not None else None, act)
            t = torch.from_numpy(t_np).to(x.device)
        else:
            t, t_np = x.detach(), _np(x)
        what = f"mse_calibrate {tuple(x.shape)} per_channel={per_channel} pre={pre is not None}"
        mn, mx = oracle.c_minmax(t_np, per_channel)
        mv = oracle.c_absmax(mn, mx)
        _bits_equal(_np(cal.absmax), mv, what + " absmax")
        grid = torch.stack([torch.linspace(0.1 * float(v), 1.2 * float(v), cal.n_cand) for v in mv.tolist()], 1)
        _bits_equal(_np(cal.grid), grid.numpy(), what + " grid")
        a = dict(x=t, grid=cal.grid, mbits_list=cal.mbits_list, per_channel=per_channel, n_bits=cal.n_bits, sign_bits=cal.sign_bits,
                 mses=cal.mses)
        self._check_mse_grid(a, cal.mses, np.zeros(1, np.float32))
        self.count["mse_grid"] += 1
        # the selection (range_estimators.py:350-369) from the device's own table, then the batch quantized with it
        got = _np(cal.mses)
        m, arg = self._choice(got)
        assert float(cal.mbits.cpu()) == cal.mbits_list[m] and int(cal.vote.cpu()) == m, what
        sel = _np(cal.grid)[arg, np.arange(got.shape[2])]
        _bits_equal(_np(cal.maxval), sel, what + " maxval")
        if y is not None:
            _bits_equal(_np(y), oracle.c_quantize(t_np, sel, cal.mbits_list[m], cal.n_bits, cal.sign_bits), what + " y")
            self.count["quantize"] += 1
            self.elements += t_np.size

    def _check_mse_grid(self, a, out, pre):
        x, grid = a["x"].contiguous(), a["grid"]
        assert not pre.any(), "one calibration batch: the table starts at zero, so the increment is the table"
        got = _np(out)
        kw = dict(n_bits=a["n_bits"], sign_bits=a["sign_bits"])
        mb = list(a["mbits_list"])
        what = f"mse_grid {tuple(x.shape)} per_channel={a['per_channel']}"
        if a["per_channel"] or x.numel() <= MSE_FULL_ORACLE_MAX:
            ref = oracle.c_mse_grid(_np(x), a["per_channel"], _np(grid), mb, **kw)
            self._check_choice(a, got, ref, what)
            # K4 forms x / s_p as x * 2^frac(bias) * 2^j (fp8q_mse.hip): the quotient can differ from the IEEE division
            # by ~2.4e-7 relative, which moves rint() only for an element that close to the midpoint of two grid
            # points -- where both are (almost) equally far, so that element's squared error changes by <= 2^(M+3) *
            # 2.4e-7 relative.  On a 9-element depthwise filter one such element can be most of the row's mean.
            inner = x.numel() // x.shape[0] if a["per_channel"] else x.numel()
            np.testing.assert_allclose(got, ref, rtol=1e-5 if inner >= 256 else 1e-4, atol=1e-30, err_msg=what)
            close = np.isclose(got, ref, rtol=1e-5, atol=1e-30) | ~np.isfinite(ref)
            assert close.mean() >= 0.999, (what, close.mean())
        else:
            # a long per-tensor row: (i) the launch equals the element-weighted mean of launches over 1 Mi-element
            # slices (its split / accumulate logic at this size), (ii) two of those slices against the oracle
            flat = x.view(-1)
            n = flat.numel()
            acc = np.zeros(got.shape, np.float64)
            rng = np.random.RandomState(n % 9973)
            nsl = -(-n // MSE_SLICE)
            picks = set(int(v) for v in rng.choice(nsl, size=min(2 if len(mb) == 1 else 1, nsl), replace=False))
            for s in range(nsl):
                sl = flat[s * MSE_SLICE:(s + 1) * MSE_SLICE]
                part = torch.zeros_like(a["mses"])
                self.real["mse_grid"](sl, False, grid, mb, a["n_bits"], a["sign_bits"], part)
                p = _np(part)
                acc += p.astype(np.float64) * sl.numel()
                if s in picks:
                    if len(mb) > 1:     # mantissa search: 6 x the oracle's work per element -> a quarter of the slice
                        sl = sl[: MSE_SLICE // 4]
                        part = torch.zeros_like(a["mses"])
                        self.real["mse_grid"](sl, False, grid, mb, a["n_bits"], a["sign_bits"], part)
                        p = _np(part)
                    ref = oracle.c_mse_grid(_np(sl), False, _np(grid), mb, **kw)
                    np.testing.assert_allclose(p, ref, rtol=1e-5, atol=1e-30, err_msg=what + f" slice {s}")
            np.testing.assert_allclose(got, acc / n, rtol=2e-6, atol=1e-30, err_msg=what + " vs its slices")
            self._check_choice(a, got, None, what)
        self.elements += x.numel()


def _qparams(M, w_est, a_est, incl=False):
    from quantization.quantization_manager import QMethods
    from quantization.range_estimators import RangeEstimators
    return dict(method=QMethods.fp_quantizer.cls, weight_range_method=RangeEstimators[w_est].cls,
                act_range_method=RangeEstimators[a_est].cls, n_bits=8, n_bits_act=8, per_channel_weights=True,
                fp8_kwargs=dict(maxval=None, mantissa_bits=M, set_maxval=True, learn_maxval=False,
                                learn_mantissa_bits=False, mse_include_mantissa_bits=incl, allow_unsigned=False))


def _build_full_size(tag):
    from torch import nn
    torch.manual_seed(0)
    if tag == "r18":
        from models.resnet import resnet18
        fp = resnet18()
    else:   # "mbv2", "mbv2m"
        from models.mobilenet_v2 import MobileNetV2
        fp = MobileNetV2(input_size=224)
    # random-init weights (no checkpoints on the box): batch-norm statistics from one synthetic batch, so that
    # activations have the scale of a trained network's
    fp = fp.cuda()
    bns = [m for m in fp.modules() if isinstance(m, nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    fp.train()
    with torch.no_grad():
        fp(torch.randn(16, 3, 224, 224, device="cuda"))
    fp.eval()
    for m in bns:
        m.momentum = 0.1
    if tag == "r18":
        from models.resnet_quantized import QuantizedResNet
        q = QuantizedResNet(fp, input_size=(1, 3, 224, 224), **_qparams(2, "current_minmax", "allminmax"))
    else:
        from models.mobilenet_v2_quantized import QuantizedMobileNetV2
        q = QuantizedMobileNetV2(fp, input_size=(1, 3, 224, 224), **_qparams(3, "MSE", "MSE", incl=tag == "mbv2m"))
    return q.cuda().eval()


@pytest.mark.parametrize("tag,n_weight,n_act_min", [("r18", 21, 29), ("mbv2", 53, 60), ("mbv2m", 53, 60)])
def test_every_launch_of_a_full_size_model_pass(tag, n_weight, n_act_min, monkeypatch):
    """BASELINE configs 3 / 4 at batch 64 x 224 x 224: calibration pass, fix_ranges(), validation pass, every
    fp8q.ops launch checked against the oracle on its own input.  mbv2m = config 4 with the mantissa search of the
    reference's CLI default (--fp8-mse-include-mantissa-bits: 6 widths x 111 ranges per quantizer)."""
    from fp8q import ops
    q = _build_full_size(tag)
    torch.manual_seed(1)
    calib = torch.randn(64, 3, 224, 224, device="cuda")
    val = torch.randn(64, 3, 224, 224, device="cuda")
    chk = LaunchChecker(ops, monkeypatch)
    with torch.no_grad():
        q.set_quant_state(True, True)
        q(calib)
        cal = dict(chk.count)
        q.fix_ranges()
        fixed = dict(chk.count)
        out = q(val)
    assert torch.isfinite(out).all() and out.shape == (64, 1000)
    c = chk.count
    print(f"\n{tag} @ 64x3x224x224: launches checked {c}; {chk.elements / 1e6:.0f} M elements through the oracle")
    if tag != "r18":
        # K4's decision at full size: every quantizer's chosen (mantissa bits, maxval) against the oracle
        ch = chk.choice
        print(f"{tag}: MSE decisions checked for {ch['quantizers']} quantizers / {ch['channels']} channels; "
              f"{ch['channels_with_another_candidate']} channels chose another candidate than the oracle's table "
              f"(each within 1e-6 relative of the oracle's minimum); {ch['oracle_candidates_evaluated']} near-minimum "
              f"candidates of long per-tensor rows evaluated by the oracle on the whole tensor")
        assert ch["quantizers"] == c["mse_grid"] >= n_weight + n_act_min
    if tag == "r18":
        # calibration: one fused min/max+quantize launch per weight tensor, range + quantize per activation
        assert cal["minmax_quantize"] == n_weight, cal
        assert cal["affine_act_minmax"] + cal["minmax"] >= n_act_min, cal
    else:
        # MSE estimator: grid maximum + one grid-search launch per quantizer (weights and activations)
        assert cal["mse_grid"] >= n_weight + n_act_min, cal
    assert cal["affine_act_quantize"] + cal["quantize"] >= n_act_min, cal
    # fix_ranges: all weights in the multi-tensor launch; validation: activations only (weights come from the cache)
    assert fixed["plan_launch"] - cal["plan_launch"] >= 1, fixed
    n_val = (c["affine_act_quantize"] + c["quantize"]) - (fixed["affine_act_quantize"] + fixed["quantize"])
    assert n_val >= n_act_min, (c, fixed)
    assert c["minmax"] == fixed["minmax"] and c["mse_grid"] == fixed["mse_grid"] \
        and c["minmax_quantize"] == fixed["minmax_quantize"], (c, fixed)


# ---------------------------------------------------------------------------------------------------------------------
# "top-1 unchanged" (BASELINE metric, second half) without ImageNet on the box: the whole validation pass with the HIP
# quantizers against the same pass with every quantizer replaced by the CPU oracle
# ---------------------------------------------------------------------------------------------------------------------
from oracle.in_the_loop import OracleInTheLoop, validation_parity  # noqa: E402  (the checker bench.py's cpu_baseline leg uses too)


@pytest.mark.parametrize("tag", ["r18", "mbv2"])
def test_validation_logits_identical_with_oracle_quantizers(tag, monkeypatch):
    """BASELINE configs 3 / 4 at batch 64 x 224 x 224, ranges calibrated once (HIP) and fixed: the validation pass with
    the HIP quantizers and the validation pass with every quantizer computed by the CPU oracle (same convolutions on the
    GPU in both) give bit-identical logits -- so top-1 / top-5 of the engine ARE those of the reference arithmetic the
    oracle restates: the top-1 delta attributable to the kernels is exactly zero."""
    from fp8q import ops
    monkeypatch.setattr(torch.backends.cudnn, "deterministic", True)
    monkeypatch.setattr(torch.backends.cudnn, "benchmark", False)
    q = _build_full_size(tag)
    torch.manual_seed(1)
    calib = torch.randn(64, 3, 224, 224, device="cuda")
    val = torch.randn(64, 3, 224, 224, device="cuda")
    with torch.no_grad():
        q.set_quant_state(True, True)
        q(calib)
        q.fix_ranges()
    r = validation_parity(q, val, ops)
    assert r["gpu_pass_reproducible"], "the GPU pass itself must be reproducible for this comparison to mean anything"
    # the layers' cached quantized weights came from the HIP multi-tensor launch: the oracle gives the very same tensors
    assert r["weights_checked"] >= 21 and r["weights_bit_identical"], r
    assert r["logits_bit_identical"], f"logits differ: max |d| = {r['max_abs_diff']:.3e}"
    assert r["argmax_equal"] and r["top5_equal"]
