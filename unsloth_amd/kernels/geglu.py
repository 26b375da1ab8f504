"""GeGLU (exact erf / tanh approximation) forward and fused in-place backward; mirror of
unsloth/kernels/geglu.py:56-71,126-139,170-185,247-260."""
from .swiglu import _glu_bwd, _glu_fwd


def geglu_exact_forward_kernel(gate, up):
    return _glu_fwd("uamd_geglu_exact_forward", gate, up)


def geglu_exact_backward_kernel(DW, e, g):
    return _glu_bwd("uamd_geglu_exact_backward", DW, e, g)


def geglu_approx_forward_kernel(gate, up):
    return _glu_fwd("uamd_geglu_approx_forward", gate, up)


def geglu_approx_backward_kernel(DW, e, g):
    return _glu_bwd("uamd_geglu_approx_backward", DW, e, g)
