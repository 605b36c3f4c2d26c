"""Drop-in for the reference's detect_tools/upn/ops/functions/ms_deform_attn_func.py: the same
`MSDeformAttnFunction.apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step)`
-> [N, Lq, M*D], backed by the hand-written gfx950 kernel (vlm_fo1_amd/csrc/msda.hip, fo1_ms_deform_attn_forward) instead of the
`MultiScaleDeformableAttention` CUDA extension (reference :18, :23-28).

Forward only: the reference path is inference (UPNWrapper.inference runs under torch.no_grad, inference_wrapper.py:108-140); asking
for a gradient raises.  The reference's `ms_deform_attn_core_pytorch` ("for debug and test only", :41-61) has no counterpart in
the product — a CPU restatement lives in oracle/ as the checker; CPU tensors raise here."""
import torch
from torch.autograd import Function

from vlm_fo1_amd import ops


class MSDeformAttnFunction(Function):
    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        # im2col_step only sizes the reference's per-launch batches (ms_deform_attn_cuda.cu:52-76): no effect on the result
        return ops.ms_deform_attn(value.contiguous(), value_spatial_shapes.contiguous(), value_level_start_index.contiguous(),
                                  sampling_locations.contiguous(), attention_weights.contiguous())

    @staticmethod
    def backward(ctx, grad_output):
        raise NotImplementedError("MSDeformAttnFunction: backward is not built (inference engine; the reference trains with its CUDA extension)")
