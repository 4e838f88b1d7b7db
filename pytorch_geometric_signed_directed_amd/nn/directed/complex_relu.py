"""complex_relu_layer -- drop-in for nn/directed/complex_relu.py of the reference (one fused HIP
pass over both parts instead of four element-wise torch ops)."""
import torch

from ... import _cabi
from ..._cabi import check, ptr, stream_ptr


class _ComplexRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, real, imag):
        _cabi.require_gpu(real, imag)
        if real.dtype != torch.float32 or imag.dtype != torch.float32:
            raise TypeError("complex_relu: the HIP path computes in float32")
        if real.shape != imag.shape:
            raise ValueError("complex_relu: real and imag must have the same shape")
        real, imag = real.contiguous(), imag.contiguous()
        o_r, o_i = torch.empty_like(real), torch.empty_like(imag)
        with _cabi.on_device(real.device):
            check(_cabi.lib().pygsd_complex_relu_f32(ptr(real), ptr(imag), ptr(o_r), ptr(o_i), real.numel(),
                                                     stream_ptr()), "pygsd_complex_relu_f32")
        ctx.save_for_backward(real)
        return o_r, o_i

    @staticmethod
    def backward(ctx, g_r, g_i):
        (real,) = ctx.saved_tensors
        g_r, g_i = g_r.contiguous(), g_i.contiguous()
        o_r, o_i = torch.empty_like(g_r), torch.empty_like(g_i)
        with _cabi.on_device(real.device):
            check(_cabi.lib().pygsd_complex_relu_bwd_f32(ptr(real), ptr(g_r), ptr(g_i), ptr(o_r), ptr(o_i),
                                                         real.numel(), stream_ptr()),
                  "pygsd_complex_relu_bwd_f32")
        return o_r, o_i


class complex_relu_layer(torch.nn.Module):
    """The complex ReLU of MagNet: mask = (real >= 0) applied to both parts
    (reference complex_relu.py:21-22)."""

    def __init__(self):
        super().__init__()

    def complex_relu(self, real: torch.FloatTensor, img: torch.FloatTensor):
        return _ComplexRelu.apply(real, img)

    def forward(self, real: torch.FloatTensor, img: torch.FloatTensor):
        return self.complex_relu(real, img)
