#!/usr/bin/env python
"""`validate-quantized` on the MI355X FP8 engine -- same sub-command and flag names as the
reference's image_net.py (README.md:63-68 of the reference), e.g.

  python image_net.py validate-quantized --architecture resnet18_quantized --batch-size 64 --seed 10 \
      --n-bits 8 --cuda --load-type fp32 --quant-setup all --qmethod fp_quantizer --per-channel \
      --fp8-mantissa-bits=5 --fp8-set-maxval --no-fp8-mse-include-mantissa-bits \
      --weight-quant-method=current_minmax --act-quant-method=allminmax --num-est-batches=1 \
      --images-dir /path/to/imagenet            # or: --synthetic-batches 8  (no dataset needed)

Procedure (reference image_net.py:48-96): build the quantized model, pass `--num-est-batches`
training batches in estimate_ranges state, fix the ranges, optionally re-estimate the BN
statistics on 2 % of the training data, then run validation and report top-1 / top-5 / loss.
ignite, click, torchvision and timm are not needed.  Without an ImageNet folder the command runs
on synthetic batches and additionally reports the arg-max agreement with the fp32 network (the
PTQ top-1 *delta* needs real labels; agreement is its label-free proxy).
"""
import argparse
import logging
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from models import QuantArchitectures  # noqa: E402
from quantization.quantization_manager import QMethods  # noqa: E402
from quantization.quantized_folded_bn import BNFusedHijacker  # noqa: E402
from quantization.range_estimators import RangeEstimators  # noqa: E402
from quantization.utils import pass_data_for_range_estimation  # noqa: E402

_BOOL = argparse.BooleanOptionalAction


def build_parser():
    ap = argparse.ArgumentParser(prog="image_net.py")
    sub = ap.add_subparsers(dest="command", required=True)
    p = sub.add_parser("validate-quantized", help="PTQ validation of a pre-trained model")
    # base options (reference utils/click_options.py:23-103)
    p.add_argument("--images-dir", default=None)
    p.add_argument("--synthetic-batches", type=int, default=0,
                   help="use this many random batches instead of an ImageNet folder")
    p.add_argument("--image-size", type=int, default=224)
    p.add_argument("--cuda", action=_BOOL, default=True)
    p.add_argument("--batch-size", type=int, default=128)
    p.add_argument("--num-workers", type=int, default=16)
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--deterministic", action="store_true", default=False)
    p.add_argument("--nondeterministic", dest="deterministic", action="store_false")
    p.add_argument("--architecture", required=True, choices=QuantArchitectures.list_names())
    p.add_argument("--model-dir", default=None)
    p.add_argument("--pretrained", action=_BOOL, default=True)
    p.add_argument("--progress-bar", action=_BOOL, default=False)
    p.add_argument("--load-type", choices=["fp32", "quantized"], default="quantized")
    # quantization options (:320-440)
    p.add_argument("--weight-quant", action=_BOOL, default=True)
    p.add_argument("--act-quant", action=_BOOL, default=True)
    p.add_argument("--qmethod", default="symmetric_uniform", choices=QMethods.list_names())
    p.add_argument("--qmethod-act", default=None, choices=QMethods.list_names())
    p.add_argument("--weight-quant-method", default="current_minmax", choices=RangeEstimators.list_names())
    p.add_argument("--act-quant-method", default="running_minmax", choices=RangeEstimators.list_names())
    p.add_argument("--num-candidates", type=int, default=None)
    p.add_argument("--act-num-candidates", type=int, default=None)
    p.add_argument("--act-momentum", type=float, default=None)
    p.add_argument("--n-bits", type=int, default=8)
    p.add_argument("--n-bits-act", type=int, default=None)
    p.add_argument("--per-channel", action=_BOOL, default=False)
    p.add_argument("--num-est-batches", type=int, default=1)
    p.add_argument("--quant-setup", default="all",
                   choices=["all", "LSQ", "FP_logits", "fc4", "fc4_dw8", "LSQ_paper"])
    # fp8 options (:443-474)
    p.add_argument("--fp8-maxval", type=float, default=None)
    p.add_argument("--fp8-mantissa-bits", type=int, default=4)
    p.add_argument("--fp8-set-maxval", action=_BOOL, default=False)
    p.add_argument("--fp8-learn-maxval", action=_BOOL, default=False)
    p.add_argument("--fp8-learn-mantissa-bits", action=_BOOL, default=False)
    p.add_argument("--fp8-mse-include-mantissa-bits", action=_BOOL, default=True)
    p.add_argument("--fp8-allow-unsigned", action=_BOOL, default=False)
    # qat option that validate-quantized reads (:184-213)
    p.add_argument("--reestimate-bn-stats", action=_BOOL, default=True)
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="under torch.distributed.run (WORLD_SIZE > 1): every rank takes its images of each batch; "
                        "calibration all-reduces the activation ranges per layer (exactly the single-process "
                        "result), the metrics are summed over the ranks.  nccl = RCCL; gloo for smoke tests")
    p.add_argument("--hip-graph", action=_BOOL, default=False,
                   help="replay the quantized validation forward from a HIP graph (fixed ranges; same results, "
                        "faster for small batches)")
    return ap


def quant_params_dict(a):
    """args -> the kwargs every QuantizedModule receives (reference click_options.py:477-510)."""
    if not a.qmethod.startswith("fp_quantizer"):
        # the reference raises UnboundLocalError here (fp8_kwargs is only bound for fp_quantizer)
        raise SystemExit("validate-quantized supports --qmethod fp_quantizer only")
    w_opts, a_opts = {}, {}
    if a.num_candidates is not None:
        w_opts["num_candidates"] = a.num_candidates
    if a.act_num_candidates is not None:
        a_opts["num_candidates"] = a.num_candidates          # sic: the reference reads num_candidates
    if a.act_momentum is not None:
        a_opts["momentum"] = a.act_momentum
    fp8 = dict(maxval=a.fp8_maxval, mantissa_bits=a.fp8_mantissa_bits, set_maxval=a.fp8_set_maxval,
               learn_maxval=a.fp8_learn_maxval, learn_mantissa_bits=a.fp8_learn_mantissa_bits,
               mse_include_mantissa_bits=a.fp8_mse_include_mantissa_bits, allow_unsigned=a.fp8_allow_unsigned)
    return dict(method=QMethods[a.qmethod].cls, n_bits=a.n_bits, n_bits_act=a.n_bits_act,
                act_method=QMethods[a.qmethod_act or a.qmethod].cls, per_channel_weights=a.per_channel,
                quant_setup=a.quant_setup, weight_range_method=RangeEstimators[a.weight_quant_method].cls,
                weight_range_options=w_opts, act_range_method=RangeEstimators[a.act_quant_method].cls,
                act_range_options=a_opts, quantize_input=a.quant_setup == "LSQ_paper", fp8_kwargs=fp8)


def seed_all(seed, deterministic=False):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    if deterministic:
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False


class SyntheticLoader:
    """`n` batches of N(0,1) images (ImageNet-normalised statistics) with random labels."""

    def __init__(self, n, batch_size, size, seed):
        self.n, self.bs, self.size, self.seed = n, batch_size, size, seed

    def __len__(self):
        return self.n

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed)
        for _ in range(self.n):
            yield (torch.randn(self.bs, 3, self.size, self.size, generator=g),
                   torch.randint(0, 1000, (self.bs,), generator=g))


class RankShard:
    """Every rank sees images rank::world of each batch of `loader` (all ranks iterate the same batches).

    A ragged last batch with fewer images than ranks would leave some ranks without data while the others wait for
    them in the per-quantizer range all-reduce (a hang, not an error).  Such a rank gets a DUPLICATE of one of the
    batch's images instead, labelled IGNORE: min/max estimates are unchanged by duplicates, and evaluate() leaves
    IGNORE-labelled images out of every metric."""
    IGNORE = -100   # F.cross_entropy's ignore_index

    def __init__(self, loader, rank, world):
        self.loader, self.rank, self.world = loader, rank, world

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for x, y in self.loader:
            n = x.shape[0]
            if n == 0:
                continue                       # nothing for anybody: every rank skips it alike
            if self.rank < n:
                yield x[self.rank::self.world], y[self.rank::self.world]
            else:
                yield x[self.rank % n: self.rank % n + 1], torch.full_like(y[:1], self.IGNORE)


def imagenet_loaders(images_dir, size, batch_size, workers):
    """train/ and val/ ImageFolder-style trees (needs PIL; no torchvision in this image)."""
    try:
        from PIL import Image
    except Exception as e:  # pragma: no cover
        raise SystemExit(f"reading --images-dir needs Pillow ({e}); use --synthetic-batches N") from e
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)

    class Folder(torch.utils.data.Dataset):
        def __init__(self, root):
            classes = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d)))
            self.items = [(os.path.join(root, c, f), i) for i, c in enumerate(classes)
                          for f in sorted(os.listdir(os.path.join(root, c)))]

        def __len__(self):
            return len(self.items)

        def __getitem__(self, i):
            path, label = self.items[i]
            im = Image.open(path).convert("RGB")
            w, h = im.size
            s = int(round(size / 0.875)) / min(w, h)                     # resize 256, centre crop 224
            im = im.resize((max(size, round(w * s)), max(size, round(h * s))), Image.BILINEAR)
            w, h = im.size
            l, t = (w - size) // 2, (h - size) // 2
            x = torch.from_numpy(np.asarray(im.crop((l, t, l + size, t + size)), dtype=np.float32) / 255.0)
            return (x.permute(2, 0, 1) - mean) / std, label

    mk = lambda split, shuffle: torch.utils.data.DataLoader(
        Folder(os.path.join(images_dir, split)), batch_size=batch_size, shuffle=shuffle,
        num_workers=workers, pin_memory=True)
    return mk("train", True), mk("val", False)


def reestimate_bn_stats(model, loader, num_batches):
    """Average of per-batch BN statistics over `num_batches` batches, with quantization active
    (reference utils/qat_utils.py:46-90: momentum 1, BN-fused modules in train mode)."""
    print("-- Reestimate current BN statistics --")
    model.eval()
    mods = [m for m in model.modules() if isinstance(m, BNFusedHijacker)]
    saved = [m.momentum for m in mods]
    sums = [(torch.zeros_like(m.running_mean), torch.zeros_like(m.running_var)) for m in mods]
    for m in mods:
        m.momentum, m.training = 1.0, True
    device, count = next(model.parameters()).device, 0
    with torch.no_grad():
        for x, _ in loader:
            model(x.to(device))
            for m, (sm, sv) in zip(mods, sums):
                sm += m.running_mean
                sv += m.running_var
            count += 1
            if count == num_batches:
                break
    for m, mom, (sm, sv) in zip(mods, saved, sums):
        m.running_mean, m.running_var, m.momentum = sm / count, sv / count, mom
    model.eval()


def evaluate(model, loader, device, fp_model=None, hip_graph=False):
    model.eval()
    forward = model
    n = top1 = top5 = agree = 0
    loss_sum = 0.0
    ce = torch.nn.CrossEntropyLoss(reduction="sum")
    t0 = time.time()
    with torch.no_grad():
        for x, y in loader:
            x, y = x.to(device), y.to(device)
            if hip_graph and forward is model and x.is_cuda:
                from quantization.base_quantized_model import GraphedForward
                forward = GraphedForward(model, x)
            out = forward(x)
            keep = y != RankShard.IGNORE      # padding duplicates of a ragged sharded batch count nowhere
            n += int(keep.sum())
            top = out.topk(5, dim=1).indices
            top1 += int(((top[:, 0] == y) & keep).sum())
            top5 += int(((top == y[:, None]).any(1) & keep).sum())
            if bool(keep.any()):
                loss_sum += float(ce(out, y))
            if fp_model is not None:
                agree += int(((fp_model(x).argmax(1) == top[:, 0]) & keep).sum())
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        t = torch.tensor([n, top1, top5, agree, loss_sum], dtype=torch.float64,
                         device=device if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(t)
        n, top1, top5, agree, loss_sum = int(t[0]), int(t[1]), int(t[2]), int(t[3]), float(t[4])
    res = {"top_1_accuracy": top1 / n, "top_5_accuracy": top5 / n, "loss": loss_sum / n,
           "images": n, "seconds": round(time.time() - t0, 3)}
    if fp_model is not None:
        res["argmax_agreement_with_fp32"] = agree / n
    return res


def validate_quantized(a):
    print("Setting up network and data loaders")
    if a.seed is not None:
        seed_all(a.seed, a.deterministic)
    elif a.deterministic:
        raise ValueError("Enforcing determinism without providing a seed is not supported")
    qparams = quant_params_dict(a)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if a.cuda:
            torch.cuda.set_device(local % torch.cuda.device_count())
        if not dist.is_initialized():
            dist.init_process_group(a.dist_backend)
    device = torch.device("cuda", torch.cuda.current_device()) if a.cuda else torch.device("cpu")
    synthetic = a.images_dir is None
    if synthetic:
        n = a.synthetic_batches or 4
        train_loader = SyntheticLoader(max(a.num_est_batches, 2), a.batch_size, a.image_size, 1234)
        val_loader = SyntheticLoader(n, a.batch_size, a.image_size, 4321)
    else:
        train_loader, val_loader = imagenet_loaders(a.images_dir, a.image_size, a.batch_size, a.num_workers)
    bn_loader = train_loader          # BN statistics: every rank runs the SAME full batches (replicated, no exchange)
    if world > 1:
        train_loader, val_loader = RankShard(train_loader, rank, world), RankShard(val_loader, rank, world)
    pretrained = a.pretrained and a.model_dir is not None
    model = QuantArchitectures[a.architecture](pretrained=pretrained, load_type=a.load_type,
                                               model_dir=a.model_dir, **qparams).to(device)
    fp_model = None
    if synthetic and not pretrained:
        # random-init weights with the default BN statistics (mean 0, var 1) give degenerate logits (all classes
        # equal), which makes "argmax agreement with fp32" noise: take the BN statistics from synthetic batches
        # first, in full precision -- only in this mode, which the reference does not have
        model.full_precision()
        reestimate_bn_stats(model, bn_loader, 2)
    if synthetic:
        import copy
        fp_model = copy.deepcopy(model).eval()
        fp_model.full_precision()
    if world > 1:
        from fp8q.dist import enable_distributed_calibration
        enable_distributed_calibration(model)
    if a.load_type == "fp32":
        pass_data_for_range_estimation(loader=train_loader, model=model, act_quant=a.act_quant,
                                       weight_quant=a.weight_quant, max_num_batches=a.num_est_batches,
                                       hip_graph=a.hip_graph and world == 1)      # (collectives stay out of the capture)
        model.set_quant_state(a.weight_quant, a.act_quant)
    model.fix_ranges()
    print("Model with the ranges estimated:\n{}".format(model))
    if a.reestimate_bn_stats:
        reestimate_bn_stats(model, bn_loader, max(1, int(0.02 * len(bn_loader))))
    print("Start quantized validation")
    metrics = evaluate(model, val_loader, device, fp_model, hip_graph=a.hip_graph)
    if rank == 0:
        print(metrics)
    return metrics


def main(argv=None):
    logging.basicConfig(level=os.environ.get("LOGLEVEL", "INFO"))
    a = build_parser().parse_args(argv)
    if a.command == "validate-quantized":
        return validate_quantized(a)


if __name__ == "__main__":
    main()
