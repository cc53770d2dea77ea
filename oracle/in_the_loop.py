"""TEST INFRASTRUCTURE (like everything under oracle/): the CPU oracle standing in for this library's quantizer launches
inside a model that otherwise runs on the GPU -- the checker behind "top-1 delta attributable to the kernels = 0".

Used by tests/test_baseline_size.py and by bench.py's cpu_baseline leg; never by the product."""
import numpy as np
import torch

import oracle


def _np(t):
    return t.detach().cpu().contiguous().numpy()


class OracleInTheLoop:
    """fp8q.ops look-alike for CUDA tensors whose arithmetic is the CPU oracle (device -> host -> oracle -> device): the
    convolutions / matmuls of the model stay on the GPU (same MIOpen / rocBLAS kernels as in the HIP run), only the
    quantizers change."""

    @staticmethod
    def _dev(a, like):
        return torch.from_numpy(np.ascontiguousarray(a)).to(like.device)

    @classmethod
    def quantize(cls, x, maxval, mbits, n_bits=8, sign_bits=1, out=None):
        y = cls._dev(oracle.c_quantize(_np(x), _np(maxval).reshape(-1), mbits, n_bits, sign_bits), x)
        if out is not None:
            out.copy_(y)
            return out
        return y

    @classmethod
    def affine_act_quantize(cls, x, maxval, mbits, n_bits=8, sign_bits=1, bn=None, residual=None, act=0, out=None, bn_ab=None,
                            prep=None):
        # (bn_ab / prep: the folded BN vector and the prepared quantizer table the layers also hand over -- the oracle
        # forms alpha / beta' from `bn` and the table from `maxval` itself)
        t = oracle.c_affine_act(_np(x), tuple(_np(b) for b in bn) if bn is not None else None,
                                _np(residual) if residual is not None else None, act)
        return cls._dev(oracle.c_quantize(t, _np(maxval).reshape(-1), mbits, n_bits, sign_bits), x)


def validation_parity(q, val, ops):
    """One validation batch through the quantized model `q` (ranges fixed) twice: with the HIP quantizers, and with every
    quantizer launch recomputed by the CPU oracle (same convolutions on the GPU in both).  Returns a dict:
      weights_checked / weights_bit_identical   the layers' cached quantized weights against the oracle
      logits_bit_identical, argmax_equal, top5_equal, max_abs_diff, gpu_pass_reproducible"""
    with torch.no_grad():
        for _ in range(2):          # MIOpen settles on its convolution algorithms during the first passes over a shape
            q(val)
        hip = q(val).clone()
        again = q(val).clone()
        cached = [(m, m._wq_cache) for m in q.modules() if getattr(m, "_wq_cache", None) is not None]
        n_w, w_ok = 0, True
        saved = ops.quantize, ops.affine_act_quantize
        ops.quantize, ops.affine_act_quantize = OracleInTheLoop.quantize, OracleInTheLoop.affine_act_quantize
        try:
            # the layers' cached quantized weights (the HIP multi-tensor launch) against the layer's own quantize_weights()
            # with the oracle underneath (fixed ranges; transposed convolutions swap dims around the quantizer themselves)
            for m, wq in cached:
                ref_w = m.quantize_weights(m.get_weight_bias()[0].detach())
                w_ok = w_ok and bool(torch.equal(ref_w.contiguous().view(torch.int32), wq.contiguous().view(torch.int32)))
                n_w += 1
            ref = q(val)
        finally:
            ops.quantize, ops.affine_act_quantize = saved
    return dict(weights_checked=n_w, weights_bit_identical=w_ok,
                gpu_pass_reproducible=bool(torch.equal(hip, again)),
                logits_bit_identical=bool(torch.equal(hip.view(torch.int32), ref.view(torch.int32))),
                argmax_equal=bool(torch.equal(hip.argmax(1), ref.argmax(1))),
                top5_equal=bool(torch.equal(hip.topk(5).indices, ref.topk(5).indices)),
                max_abs_diff=float((hip - ref).abs().max()))
