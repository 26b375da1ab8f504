"""get_lora_parameters / get_lora_parameters_bias against an INDEPENDENT mimic of PEFT's layer structure.

`peft` cannot be installed here (no network), and the product's own `unsloth_amd.lora` is by construction what these two
accessors read -- so this file restates, from PEFT's published source (peft/tuners/tuners_utils.py BaseTunerLayer and
peft/tuners/lora/layer.py LoraLayer / Linear, 0.12-0.17), the attribute protocol itself: ModuleDicts keyed by adapter name,
`scaling` dict, `merged` / `disable_adapters` / `active_adapter` / `active_adapters` as PROPERTIES over `merged_adapters`,
`_disable_adapters`, `_active_adapter`; `get_base_layer()`; nested base layers. The accessors (unsloth/kernels/utils.py:335-440)
must return what the reference would return for every state a PEFT layer can be in."""
import pytest
import torch
from torch import nn

from unsloth_amd.kernels.utils import get_lora_parameters, get_lora_parameters_bias


class BaseTunerLayerMimic:
    """tuners_utils.BaseTunerLayer: the state lives in private fields, the public names are properties."""
    adapter_layer_names = ("lora_A", "lora_B")
    _disable_adapters = False
    _active_adapter = "default"
    merged_adapters = []

    def get_base_layer(self):
        base = self
        while hasattr(base, "base_layer"):
            base = base.base_layer
        return base

    @property
    def weight(self):
        return self.get_base_layer().weight

    @property
    def bias(self):
        return self.get_base_layer().bias

    @property
    def merged(self):
        return bool(self.merged_adapters)

    @property
    def disable_adapters(self):
        return self._disable_adapters

    @property
    def active_adapter(self):
        return self._active_adapter

    @property
    def active_adapters(self):
        return [self._active_adapter] if isinstance(self._active_adapter, str) else self._active_adapter

    def set_adapter(self, names):
        self._active_adapter = [names] if isinstance(names, str) else list(names)

    def enable_adapters(self, enabled):
        self._disable_adapters = not enabled


class LoraLinearMimic(nn.Module, BaseTunerLayerMimic):
    """lora/layer.py: LoraLayer.__init__ + Linear.update_layer, nothing of the forward."""

    def __init__(self, base_layer, adapter_name="default", r=8, lora_alpha=16, use_rslora=False):
        super().__init__()
        self.base_layer = base_layer
        self.r, self.lora_alpha, self.scaling = {}, {}, {}
        self.lora_dropout = nn.ModuleDict({})
        self.lora_A, self.lora_B = nn.ModuleDict({}), nn.ModuleDict({})
        self.merged_adapters = []
        self._disable_adapters = False
        self.in_features, self.out_features = base_layer.in_features, base_layer.out_features
        self.update_layer(adapter_name, r, lora_alpha, use_rslora)
        self._active_adapter = adapter_name

    def update_layer(self, name, r, lora_alpha, use_rslora=False):
        self.r[name], self.lora_alpha[name] = r, lora_alpha
        self.lora_dropout[name] = nn.Identity()
        self.lora_A[name] = nn.Linear(self.in_features, r, bias=False)
        self.lora_B[name] = nn.Linear(r, self.out_features, bias=False)
        self.scaling[name] = lora_alpha / (r ** 0.5) if use_rslora else lora_alpha / r


def _layer(bias=False, **kw):
    torch.manual_seed(0)
    return LoraLinearMimic(nn.Linear(32, 48, bias=bias), **kw)


def test_enabled_adapter_returns_weight_factors_and_scaling():
    m = _layer(r=8, lora_alpha=16)
    W, q, A, B, s = get_lora_parameters(m)
    assert W is m.base_layer.weight and q is None
    assert A is m.lora_A["default"].weight and B is m.lora_B["default"].weight and s == 2.0
    assert A.shape == (8, 32) and B.shape == (48, 8)
    out = get_lora_parameters_bias(m)
    assert len(out) == 6 and out[5] is None and out[2] is A and out[3] is B


def test_rslora_scaling_and_bias():
    m = _layer(bias=True, r=16, lora_alpha=16, use_rslora=True)
    *_, s = get_lora_parameters(m)
    assert s == pytest.approx(4.0)
    assert get_lora_parameters_bias(m)[5] is m.base_layer.bias


def test_disabled_or_merged_adapters_return_the_bare_weight():
    m = _layer()
    m.enable_adapters(False)                              # `with model.disable_adapter():` (DPO's reference pass)
    assert get_lora_parameters(m)[2:] == (None, None, None)
    assert get_lora_parameters_bias(m)[2:5] == (None, None, None)
    m.enable_adapters(True)
    m.merged_adapters.append("default")                   # after merge(): the weight already holds W + s B A
    assert get_lora_parameters(m)[2:] == (None, None, None)
    m.merged_adapters.clear()
    assert get_lora_parameters(m)[2] is m.lora_A["default"].weight


def test_second_adapter_becomes_active():
    m = _layer(r=8, lora_alpha=8)
    m.update_layer("policy", r=4, lora_alpha=16)
    m.set_adapter("policy")
    W, q, A, B, s = get_lora_parameters(m)
    assert A is m.lora_A["policy"].weight and B.shape == (48, 4) and s == 4.0
    m.set_adapter(["default", "policy"])                  # several active: the reference reads the FIRST (utils.py:372-375)
    assert get_lora_parameters(m)[2] is m.lora_A["default"].weight


def test_plain_linear_without_any_peft_attribute_is_a_frozen_projection():
    lin = nn.Linear(32, 48, bias=True)
    lin.merged = False                                    # the reference reads `.merged` unconditionally (utils.py:369)
    W, q, A, B, s, b = get_lora_parameters_bias(lin)
    assert W is lin.weight and (q, A, B, s) == (None, None, None, None) and b is lin.bias


def test_quant_state_travels_on_the_weight():
    m = _layer()
    state = object()
    m.base_layer.weight.quant_state = state               # bitsandbytes Params4bit carries it exactly there
    assert get_lora_parameters(m)[1] is state and get_lora_parameters_bias(m)[1] is state


def test_qat_fake_quantizers_are_applied_to_what_is_returned():
    """utils.py:343-349, :380-392: `weight_fake_quantizer` on the base layer and on the factors' Linear modules."""
    m = _layer()
    m.base_layer.weight_fake_quantizer = lambda w: w * 0 + 1
    m.lora_A["default"].weight_fake_quantizer = lambda w: w * 0 + 2
    m.lora_B["default"].weight_fake_quantizer = lambda w: w * 0 + 3
    W, _, A, B, _ = get_lora_parameters(m)
    assert float(W.detach().mean()) == 1.0 and float(A.detach().mean()) == 2.0 and float(B.detach().mean()) == 3.0
