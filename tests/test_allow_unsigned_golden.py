"""allow_unsigned over several batches against the REFERENCE (tests/golden/g12_allow_unsigned.npz, written by
tests/golden/make_golden.py:make_g12 from /root/reference's QuantizationManager): after every batch the output, maxval and
sign_bits of fp8_quantizer.py:216-225's sticky decision -- the three min/max estimators and the MSE estimator (fixed mantissa
width), per tensor and per channel, four data sequences.  On the CPU the oracle stands in for the HIP ops (host decision); on
the GPU the kernels run and the sign is decided on the device (fp8q_sign_fold_u8, fp8q_quantize_ds_f32), without a host round
trip."""
import os

import numpy as np
import pytest
import torch

from parity import assert_parity, elem_step

ESTS = ["current_minmax", "allminmax", "running_minmax", "MSE"]
SEQS = {"signed": "sss", "relu": "rrr", "signed_then_relu": "srr", "relu_then_signed": "rss"}


def _manager(est_name, pc):
    from quantization.quantization_manager import QuantizationManager, QMethods
    from quantization.range_estimators import RangeEstimators
    return QuantizationManager(qmethod=QMethods.fp_quantizer.cls, init=RangeEstimators[est_name].cls, per_channel=bool(pc),
                               qparams=dict(n_bits=8, mantissa_bits=3, maxval=None, set_maxval=True,
                                            mse_include_mantissa_bits=False, allow_unsigned=True))


def _run(g, est_name, pc, sname, device, sync_free):
    raw = [torch.from_numpy(b).to(device) for b in g[f"raw_pc{pc}"]]
    xs = [torch.relu(b) if k == "r" else b for b, k in zip(raw, SEQS[sname])]
    key = f"{est_name}_pc{pc}_{sname}"
    ref_y, ref_mv, ref_sign = g[key + "_y"], g[key + "_maxval"], g[key + "_sign"]
    qm = _manager(est_name, pc)
    if sync_free:                        # (allocations and code objects first; then the judged pass under the guard)
        for x in xs:
            qm(x)
        qm = _manager(est_name, pc)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
    try:
        ys, mvs, signs = [], [], []
        for x in xs:
            ys.append(qm(x))
            mvs.append(qm.quantizer.maxval.detach().clone().reshape(-1))
            # sign_bits after this batch: the host's value, or a copy of the pending device flag (read after the guard)
            host = qm.quantizer.__dict__.get("_sign_host")
            signs.append(host if host is not None else qm.quantizer._pending_sign_bits().clone())
    finally:
        if sync_free:
            torch.cuda.set_sync_debug_mode("default")
    if sync_free:
        assert any(isinstance(v, torch.Tensor) for v in signs), "the sign was decided on the host"
    for i, x in enumerate(xs):
        sign = int(signs[i].item()) if isinstance(signs[i], torch.Tensor) else int(signs[i])
        assert sign == int(ref_sign[i]), (key, i, sign, ref_sign[i])
        mv = mvs[i].cpu().numpy()
        if est_name == "MSE":
            # the chosen candidate: the reference's, or an equally good one of the same grid (K4's 1e-5 contract)
            np.testing.assert_allclose(mv, ref_mv[i], rtol=0.02)
        else:
            np.testing.assert_array_equal(mv, ref_mv[i])
        if est_name != "MSE" or np.array_equal(mv, ref_mv[i]):
            mvb = ref_mv[i].reshape([-1] + [1] * (x.dim() - 1)) if pc else ref_mv[i]
            assert_parity(ys[i].cpu().numpy(), ref_y[i], elem_step(x.cpu().numpy(), mvb, 3, 8, int(ref_sign[i])), what=f"{key}[{i}]")
    assert int(qm.quantizer.sign_bits) == int(ref_sign[-1])


@pytest.mark.parametrize("sname", list(SEQS))
@pytest.mark.parametrize("pc", [0, 1])
@pytest.mark.parametrize("est_name", ESTS)
def test_allow_unsigned_sequences_on_oracle_backend_cpu(golden_dir, est_name, pc, sname):
    import oracle_ops
    g = np.load(os.path.join(golden_dir, "g12_allow_unsigned.npz"))
    with oracle_ops.patched():
        _run(g, est_name, pc, sname, "cpu", sync_free=False)


@pytest.mark.gpu
@pytest.mark.parametrize("sname", list(SEQS))
@pytest.mark.parametrize("pc", [0, 1])
@pytest.mark.parametrize("est_name", ESTS)
def test_allow_unsigned_sequences_vs_reference_gpu(golden_dir, est_name, pc, sname):
    import fp8q
    g = np.load(os.path.join(golden_dir, "g12_allow_unsigned.npz"))
    fp8q.ops.mse_linspace(torch.ones(1, device="cuda"))       # the once-per-process self-check synchronises
    _run(g, est_name, pc, sname, "cuda", sync_free=True)
