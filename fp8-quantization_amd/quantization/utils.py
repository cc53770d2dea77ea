"""Calibration driver (reference quantization/utils.py:74-115)."""
import torch


def pass_data_for_range_estimation(loader, model, act_quant, weight_quant, max_num_batches=20,
                                   cross_entropy_layer=None, inp_idx=0, hip_graph=False):
    """Run up to `max_num_batches` batches through the model with every manager in
    estimate_ranges state.  Unlike the reference (:103) the inputs are not copied back to the
    host: that copy is unused there and costs a device sync per batch."""
    print("\nEstimate quantization ranges on training data")
    model.set_quant_state(weight_quant, act_quant)
    model.eval()   # BN statistics must not move
    if cross_entropy_layer is not None:
        raise NotImplementedError("cross-entropy range estimation is not part of this build")
    device = next(model.parameters()).device
    forward = model
    if hip_graph and device.type == "cuda":       # batches >= 3 of a shape replay a captured calibration forward
        from .model import GraphedCalibration
        forward = GraphedCalibration(model)
    with torch.no_grad():
        for i, data in enumerate(loader):
            if isinstance(data, (tuple, list)):
                forward(data[inp_idx].to(device=device))
            else:
                model(**{k: v.to(device=device) for k, v in data.items()})
            print(f"proccesed step={i}")
            if i >= max_num_batches - 1 or not act_quant:
                break
