"""validate-quantized command: flag names of the reference's README repro line parse, map to the same
quantization kwargs, and (GPU) the procedure runs end to end on synthetic batches."""
import os

import pytest
import torch

README_FLAGS = ("validate-quantized --architecture resnet18_quantized --batch-size 64 --seed 10 --n-bits 8 --cuda "
                "--load-type fp32 --quant-setup all --qmethod fp_quantizer --per-channel --fp8-mantissa-bits=5 "
                "--fp8-set-maxval --no-fp8-mse-include-mantissa-bits --weight-quant-method=current_minmax "
                "--act-quant-method=allminmax --num-est-batches=1").split()


def test_readme_command_parses_to_reference_kwargs():
    import image_net
    from quantization.quantizers.fp8_quantizer import FPQuantizer
    from quantization.range_estimators import CurrentMinMaxEstimator, AllMinMaxEstimator
    a = image_net.build_parser().parse_args(README_FLAGS)
    assert a.command == "validate-quantized" and a.batch_size == 64 and a.seed == 10 and a.cuda
    qp = image_net.quant_params_dict(a)
    assert qp["method"] is FPQuantizer and qp["act_method"] is FPQuantizer
    assert qp["weight_range_method"] is CurrentMinMaxEstimator and qp["act_range_method"] is AllMinMaxEstimator
    assert qp["per_channel_weights"] and qp["n_bits"] == 8 and qp["quant_setup"] == "all" and not qp["quantize_input"]
    assert qp["fp8_kwargs"] == dict(maxval=None, mantissa_bits=5, set_maxval=True, learn_maxval=False,
                                    learn_mantissa_bits=False, mse_include_mantissa_bits=False,
                                    allow_unsigned=False)
    # reference defaults: --load-type quantized, running_minmax activations, set_maxval off, M=4
    d = image_net.build_parser().parse_args(["validate-quantized", "--architecture", "mobilenet_v2_quantized",
                                             "--qmethod", "fp_quantizer"])
    assert d.load_type == "quantized" and d.act_quant_method == "running_minmax" and d.batch_size == 128
    assert d.fp8_mantissa_bits == 4 and not d.fp8_set_maxval and d.fp8_mse_include_mantissa_bits
    assert d.reestimate_bn_stats and d.num_est_batches == 1
    with pytest.raises(SystemExit):      # INT qmethods: the reference crashes (UnboundLocalError) here
        image_net.quant_params_dict(image_net.build_parser().parse_args(
            ["validate-quantized", "--architecture", "resnet18_quantized"]))


@pytest.mark.gpu
def test_validate_quantized_synthetic_resnet18():
    import image_net
    argv = [f for f in README_FLAGS if not f.startswith("--batch-size") and f != "64"]
    argv += ["--batch-size", "8", "--synthetic-batches", "2", "--image-size", "64", "--fp8-mantissa-bits=3"]
    m = image_net.main(argv)
    assert m["images"] == 16 and 0.0 <= m["top_1_accuracy"] <= 1.0 and m["loss"] == m["loss"]
    assert 0.0 <= m["argmax_agreement_with_fp32"] <= 1.0


@pytest.mark.gpu
def test_validate_quantized_hip_graph_same_metrics():
    """--hip-graph replays the validation forward from a captured HIP graph: same metrics as the eager loop."""
    import image_net
    argv = [f for f in README_FLAGS if not f.startswith("--batch-size") and f != "64"]
    argv += ["--batch-size", "8", "--synthetic-batches", "3", "--image-size", "64", "--fp8-mantissa-bits=3"]
    eager = image_net.main(argv)
    graphed = image_net.main(argv + ["--hip-graph"])
    # two independent runs (model build + calibration each): MIOpen may pick other convolution algorithms, and a
    # last-ulp difference before a quantizer moves activations by a grid step, so only coarse agreement is asserted
    # here; the replay itself is bit-identical to eager (test_quantized_forward_in_a_hip_graph)
    assert eager["images"] == graphed["images"] == 24
    for k in ("top_1_accuracy", "top_5_accuracy", "argmax_agreement_with_fp32"):
        assert 0.0 <= graphed[k] <= 1.0
    assert abs(eager["loss"] - graphed["loss"]) <= 0.05 * abs(eager["loss"])


@pytest.mark.gpu
def test_validate_quantized_two_ranks_equal_one_process(tmp_path):
    """validate-quantized under torch.distributed.run with two ranks (gloo, both on the one GPU of the test box): every
    rank takes its images of each batch, calibration all-reduces the ranges per layer, metrics are summed -- the
    result must be the single-process one (same images overall, same ranges)."""
    import json
    import subprocess
    import sys
    import image_net
    argv = [f for f in README_FLAGS if not f.startswith("--batch-size") and f != "64"]
    argv += ["--batch-size", "8", "--synthetic-batches", "3", "--image-size", "64", "--fp8-mantissa-bits=3"]
    single = image_net.main(argv)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(here, "fp8-quantization_amd", "image_net.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29541", script] + argv + ["--dist-backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{'top_1_accuracy'")][-1]
    multi = eval(line)   # noqa: S307  (the script prints a python dict literal)
    assert multi["images"] == single["images"] == 24
    # same images, ranges identical by construction; MIOpen may pick other algorithms for the smaller per-rank batch
    assert abs(multi["loss"] - single["loss"]) <= 0.05 * abs(single["loss"])
    for k in ("top_1_accuracy", "top_5_accuracy"):
        assert abs(multi[k] - single[k]) <= 2.0 / 24 + 1e-12


def test_rank_shard_pads_ragged_batches():
    """A batch with fewer images than ranks: the rank without an image gets an IGNORE-labelled duplicate (so it still
    joins the per-quantizer collectives) and evaluate() counts it nowhere."""
    import image_net
    x = torch.arange(3 * 2 * 2 * 2, dtype=torch.float32).view(3, 2, 2, 2)
    y = torch.tensor([5, 6, 7])
    loader = [(x, y), (x[:1], y[:1]), (x[:0], y[:0])]
    got = [[(bx.shape[0], by.tolist()) for bx, by in image_net.RankShard(loader, r, 2)] for r in range(2)]
    assert got[0] == [(2, [5, 7]), (1, [5])]
    assert got[1] == [(1, [6]), (1, [image_net.RankShard.IGNORE])]
    bx, _ = list(image_net.RankShard(loader, 1, 2))[1]
    assert torch.equal(bx, x[:1])

    class Net(torch.nn.Module):
        def forward(self, t):
            out = torch.zeros(t.shape[0], 10)
            out[:, 5] = 1.0
            return out
    res = image_net.evaluate(Net(), list(image_net.RankShard(loader, 1, 2)), torch.device("cpu"))
    assert res["images"] == 1 and res["top_1_accuracy"] == 0.0          # the duplicate (a "5") is not counted
