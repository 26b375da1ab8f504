"""Kernel API of the MI355X path: the same names unsloth/kernels/__init__.py:15-62 exports."""
from .cross_entropy_loss import (
    fast_cross_entropy_loss,
    post_patch_loss_function,
    patch_loss_functions,
    unsloth_fused_ce_loss,
    Fast_CrossEntropyLoss,
)
from .rms_layernorm import (
    fast_rms_layernorm,
    patch_rms_layernorm,
    unpatch_rms_layernorm,
    Fast_RMS_Layernorm,
)
from .layernorm import fast_layernorm, patch_layernorm, unpatch_layernorm, Fast_Layernorm
from .rope_embedding import (
    fast_rope_embedding,
    inplace_rope_embedding,
    fast_mrope_embedding,
    Fast_MRoPE_Embedding_QK,
    Fast_RoPE_Embedding,
    Fast_RoPE_Embedding_QK,
)
from .swiglu import swiglu_fg_kernel, swiglu_DWf_DW_dfg_kernel
from .geglu import (
    geglu_exact_forward_kernel,
    geglu_exact_backward_kernel,
    geglu_approx_forward_kernel,
    geglu_approx_backward_kernel,
)
from .fast_lora import (
    get_lora_parameters,
    apply_lora_mlp_swiglu,
    apply_lora_mlp_geglu_exact,
    apply_lora_mlp_geglu_approx,
    apply_lora_qkv,
    apply_lora_o,
    fast_lora_forward,
    LoRA_MLP,
    LoRA_QKV,
    LoRA_W,
)
from .utils import (
    fast_dequantize,
    fast_gemv,
    QUANT_STATE,
    fast_linear_forward,
    matmul_lora,
    get_lora_parameters_bias,
    calculate_settings,
    lora_linear_forward,
    lora_linear_dx,
)
