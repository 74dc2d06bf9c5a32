"""HIP-backed mirror of osu_fusion/modules/lora_layers.py and of the peft (0.12) pieces trainer_peft.py:236-244 wires up:
LoRA / DoRA adapters on `attn.to_q`, `attn.to_kv` (peft `lora.Linear`) and `block1.proj`, `block2.proj` (`LoraConv1d`).

Same module tree and state_dict keys as a peft-wrapped reference model (`<target>.base_layer.{weight,bias}`,
`<target>.lora_A.<adapter>.weight`, `<target>.lora_B.<adapter>.weight`, `<target>.lora_magnitude_vector.<adapter>.weight`), so
adapter checkpoints written by `peft_model.save_pretrained` map one to one (`get_adapter_state_dict` / `load_adapter_state_dict`
add / strip peft's `base_model.model.` prefix and the adapter name exactly as peft does).

peft itself is not installed in this image and the reference module cannot be imported without it: the arithmetic follows the
explicit formulas in lora_layers.py (file:line cited per method) -- **parity unpinned** against peft's own Linear path.

Compute: the wrappers are parameter containers plus an `Fn.Adapter` descriptor.  Forward and input-gradient run the layer's
ordinary GEMM on the effective weight g*(W + s*BA) (rebuilt once per optimizer step); adapter gradients use rank-r GEMMs
(functional.adapter_grads).  See functional.Adapter for why this equals the reference's three-convolution formula.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Iterable, List, Optional, Sequence

import torch
import torch.nn as nn

from .. import functional as Fn
from .. import runtime as rt


def _norm_rows(weight: torch.Tensor) -> torch.Tensor:
    return weight.reshape(weight.shape[0], -1).norm(p=2, dim=1)


class DoraLinearLayer(nn.Module):
    """peft 0.12 `DoraLinearLayer`: holds the magnitude vector `weight` (out_features,)."""

    def __init__(self, fan_in_fan_out: bool = False) -> None:
        super().__init__()
        self.fan_in_fan_out = fan_in_fan_out

    def get_weight_norm(self, weight: torch.Tensor, lora_weight: torch.Tensor, scaling: float) -> torch.Tensor:
        return _norm_rows(weight + scaling * lora_weight).to(weight.dtype)

    def update_layer(self, *, base_layer: nn.Module, lora_A: torch.Tensor, lora_B: torch.Tensor, scaling: float) -> None:
        weight = base_layer.weight.detach().float()
        lora_weight = (lora_B.detach().float().flatten(1) @ lora_A.detach().float().flatten(1)).reshape(weight.shape)
        self.weight = nn.Parameter(self.get_weight_norm(weight, lora_weight, scaling).clone(), requires_grad=True)


class DoraConv1dLayer(DoraLinearLayer):
    """lora_layers.py:15-58: magnitude of shape (1, out, 1) = per-out-channel L2 norm over (in, k)."""

    def get_weight_norm(self, weight: torch.Tensor, lora_weight: torch.Tensor, scaling: float) -> torch.Tensor:
        weight = weight + scaling * lora_weight                                     # lora_layers.py:23
        return weight.norm(p=2, dim=(1, 2), keepdim=True).transpose(1, 0)           # lora_layers.py:25


class _LoraBase(nn.Module):
    """State shared by the two wrappers (the subset of peft's LoraLayer this path uses: one active adapter, dropout 0)."""

    def __init__(self, base_layer: nn.Module) -> None:
        super().__init__()
        self.base_layer = base_layer
        self.r: Dict[str, int] = {}
        self.lora_alpha: Dict[str, int] = {}
        self.scaling: Dict[str, float] = {}
        self.use_dora: Dict[str, bool] = {}
        self.lora_dropout = nn.ModuleDict()
        self.lora_A = nn.ModuleDict()
        self.lora_B = nn.ModuleDict()
        self.lora_magnitude_vector = nn.ModuleDict()
        self.merged_adapters: List[str] = []
        self._active_adapter = "default"
        self.disable_adapters = False
        self._norm_cache: Dict[str, torch.Tensor] = {}
        self._adapter_obj: Dict[str, Fn.Adapter] = {}

    # -- peft surface ----------------------------------------------------------------------------------
    def get_base_layer(self) -> nn.Module:
        return self.base_layer

    @property
    def weight(self) -> torch.Tensor:
        return self.base_layer.weight

    @property
    def bias(self) -> Optional[torch.Tensor]:
        return self.base_layer.bias

    @property
    def merged(self) -> bool:
        return bool(self.merged_adapters)

    @property
    def active_adapters(self) -> List[str]:
        return [self._active_adapter]

    def _make_ab(self, r: int):  # pragma: no cover - overridden
        raise NotImplementedError

    def update_layer(self, adapter_name: str, r: int, lora_alpha: int, lora_dropout: float = 0.0, init_lora_weights=True,
                     use_rslora: bool = False, use_dora: bool = False) -> None:
        """lora_layers.py:126-181."""
        if r <= 0:
            raise ValueError(f"`r` should be a positive integer value but the value passed is {r}")
        if r % 8:
            raise NotImplementedError(f"the rank-r GEMMs need r % 8 == 0 (16-byte bf16 rows); got r={r}")
        if lora_dropout > 0.0:
            raise NotImplementedError("lora_dropout > 0 has no HIP path (trainer_peft.py uses the default 0.0)")
        self.r[adapter_name], self.lora_alpha[adapter_name] = r, lora_alpha
        self.lora_dropout[adapter_name] = nn.Identity()
        a, b = self._make_ab(r)
        self.lora_A[adapter_name], self.lora_B[adapter_name] = a, b
        self.scaling[adapter_name] = lora_alpha / math.sqrt(r) if use_rslora else lora_alpha / r
        if init_lora_weights:
            nn.init.kaiming_uniform_(a.weight, a=math.sqrt(5))                     # peft LoraLayer.reset_lora_parameters
            nn.init.zeros_(b.weight)
        dev = self.base_layer.weight.device
        a.to(dev); b.to(dev)
        self.use_dora[adapter_name] = bool(use_dora)
        if use_dora:
            self.dora_init(adapter_name)
        self._active_adapter = adapter_name

    def _dora_layer(self) -> DoraLinearLayer:  # pragma: no cover - overridden
        raise NotImplementedError

    def dora_init(self, adapter_name: str) -> None:
        """lora_layers.py:183-197."""
        layer = self._dora_layer()
        layer.update_layer(base_layer=self.base_layer, lora_A=self.lora_A[adapter_name].weight, lora_B=self.lora_B[adapter_name].weight,
                           scaling=self.scaling[adapter_name])
        self.lora_magnitude_vector[adapter_name] = layer

    def get_delta_weight(self, adapter: str) -> torch.Tensor:
        """scaling * (B A) shaped like the base weight (lora_layers.py:284-310)."""
        a, b = self.lora_A[adapter].weight, self.lora_B[adapter].weight
        return (b.flatten(1) @ a.flatten(1)).reshape(self.base_layer.weight.shape) * self.scaling[adapter]

    def merge(self, safe_merge: bool = False, adapter_names: Optional[Sequence[str]] = None) -> None:
        """Fold the active adapter into base_layer.weight (lora_layers.py:199-246)."""
        for name in (adapter_names or self.active_adapters):
            if name not in self.lora_A or name in self.merged_adapters:
                continue
            w = self.base_layer.weight.data
            delta = self.get_delta_weight(name).detach()
            if not self.use_dora[name]:
                new = w + delta
            else:
                mag = self.lora_magnitude_vector[name]
                weight_norm = mag.get_weight_norm(w, delta, scaling=1).detach()
                self._norm_cache[name] = weight_norm
                factor = (mag.weight.detach() / weight_norm).reshape(-1, *([1] * (w.dim() - 1)))
                new = factor * (w + delta)
            if safe_merge and not torch.isfinite(new).all():
                raise ValueError(f"NaNs detected in the merged weights. The adapter {name} seems to be broken")
            self.base_layer.weight.data = new
            self.merged_adapters.append(name)
        Fn.bump_weight_epoch()

    def unmerge(self) -> None:
        """lora_layers.py:248-264."""
        while self.merged_adapters:
            name = self.merged_adapters.pop()
            w = self.base_layer.weight
            delta = self.get_delta_weight(name).detach()
            if not self.use_dora[name]:
                w.data -= delta
            else:
                weight_norm = self._norm_cache.pop(name)
                factor = (self.lora_magnitude_vector[name].weight.detach() / weight_norm).reshape(-1, *([1] * (w.dim() - 1)))
                w.data = w.data / factor - delta
        Fn.bump_weight_epoch()

    # -- kernel side -----------------------------------------------------------------------------------
    def adapter(self) -> Optional[Fn.Adapter]:
        """Descriptor handed to the autograd Functions; None when the layer must behave like its base (merged / disabled)."""
        name = self._active_adapter
        if self.disable_adapters or self.merged or name not in self.lora_A:
            return None
        ad = self._adapter_obj.get(name)
        mag = self.lora_magnitude_vector[name].weight if self.use_dora[name] else None
        if ad is None or ad.w is not self.base_layer.weight or ad.m is not mag:
            ad = Fn.Adapter(self.base_layer.weight, self.lora_A[name].weight, self.lora_B[name].weight, mag, self.scaling[name])
            self._adapter_obj[name] = ad
        return ad

    def adapter_inputs(self):
        """(adapter, lora_A.weight, lora_B.weight, magnitude) -- the trailing arguments of BlockFn / AttentionFn."""
        ad = self.adapter()
        return (None, None, None, None) if ad is None else (ad, ad.a, ad.b, ad.m)


class LoraConv1d(_LoraBase):
    """lora_layers.py:101-332.  lora_A = Conv1d(in, r, k, stride, padding, bias=False), lora_B = Conv1d(r, out, 1, bias=False)."""

    def __init__(self, base_layer: nn.Conv1d, adapter_name: str = "default", r: int = 0, lora_alpha: int = 1, lora_dropout: float = 0.0,
                 init_lora_weights=True, use_rslora: bool = False, use_dora: bool = False, **kwargs) -> None:
        super().__init__(base_layer)
        self.in_features, self.out_features = base_layer.in_channels, base_layer.out_channels
        self.update_layer(adapter_name, r, lora_alpha=lora_alpha, lora_dropout=lora_dropout, init_lora_weights=init_lora_weights,
                          use_rslora=use_rslora, use_dora=use_dora)
        self._cache = Fn.PackCache()

    def _make_ab(self, r: int):
        b = self.base_layer
        return (nn.Conv1d(self.in_features, r, b.kernel_size[0], stride=b.stride[0], padding=b.padding[0], bias=False),
                nn.Conv1d(r, self.out_features, 1, stride=1, bias=False))

    def _dora_layer(self) -> DoraLinearLayer:
        return DoraConv1dLayer(fan_in_fan_out=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(B, C_in, L) -> (B, C_out, L): the adapted convolution on its own (lora_layers.py:312-328); inside the UNet the owning
        Block fuses it with GroupNorm instead."""
        b = self.base_layer
        if b.kernel_size[0] % 2 == 0 or b.stride[0] != 1 or b.padding[0] != b.kernel_size[0] // 2:
            raise NotImplementedError("stand-alone LoraConv1d.forward covers the 'same' convolutions the reference adapts")
        rows = rt.to_rows(x, rt.compute_dtype(b.weight.dtype))
        ad = self.adapter()
        if ad is None:
            return rt.to_logical(Fn.ConvFn.apply(rows, b.weight, b.bias, self._cache, "same"))
        return rt.to_logical(Fn.AdaptedConvFn.apply(rows, b.bias, self._cache, "same", ad, ad.a, ad.b, ad.m))


class LoraLinear(_LoraBase):
    """peft `lora.Linear` as configured at trainer_peft.py:237-244 (attn.to_q / attn.to_kv): lora_A = Linear(in, r, bias=False),
    lora_B = Linear(r, out, bias=False), DoRA magnitude (out,)."""

    def __init__(self, base_layer: nn.Linear, adapter_name: str = "default", r: int = 0, lora_alpha: int = 1, lora_dropout: float = 0.0,
                 init_lora_weights=True, use_rslora: bool = False, use_dora: bool = False, **kwargs) -> None:
        super().__init__(base_layer)
        self.in_features, self.out_features = base_layer.in_features, base_layer.out_features
        self.update_layer(adapter_name, r, lora_alpha=lora_alpha, lora_dropout=lora_dropout, init_lora_weights=init_lora_weights,
                          use_rslora=use_rslora, use_dora=use_dora)
        self._cache = Fn.PackCache()

    def _make_ab(self, r: int):
        return nn.Linear(self.in_features, r, bias=False), nn.Linear(r, self.out_features, bias=False)

    def _dora_layer(self) -> DoraLinearLayer:
        return DoraLinearLayer(fan_in_fan_out=False)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """(..., in) -> (..., out) on its own; inside the UNet the owning Attention fuses q|kv into one GEMM instead."""
        b = self.base_layer
        lead = x.shape[:-1]
        rows = rt.cast_rows(x.reshape(1, -1, x.shape[-1]).contiguous(), rt.compute_dtype(b.weight.dtype))
        ad = self.adapter()
        if ad is None:
            y = Fn.ConvFn.apply(rows, b.weight, b.bias, self._cache, "same")
        else:
            y = Fn.AdaptedConvFn.apply(rows, b.bias, self._cache, "same", ad, ad.a, ad.b, ad.m)
        return rt.cast_rows(y, x.dtype if x.dtype.is_floating_point else torch.float32).reshape(*lead, -1)


# -- get_peft_model for this path --------------------------------------------------------------------------------
@dataclass
class LoraConfig:
    """The fields of peft.LoraConfig that trainer_peft.py:237-242 sets (same names, same defaults where it sets none)."""
    r: int = 8
    lora_alpha: int = 8
    lora_dropout: float = 0.0
    use_dora: bool = False
    use_rslora: bool = False
    init_lora_weights: bool = True
    target_modules: List[str] = field(default_factory=lambda: ["attn.to_q", "attn.to_kv", "attn.linear", "block1.proj", "block2.proj"])


def _matches(name: str, targets: Iterable[str]) -> bool:
    return any(name == t or name.endswith("." + t) for t in targets)        # peft's suffix rule for list-valued target_modules


def get_peft_model(model: nn.Module, config: LoraConfig, adapter_name: str = "default") -> nn.Module:
    """Freeze `model`, wrap every target Conv1d / Linear in place (trainer_peft.py:236-244) and return it.  Unlike peft there is no
    PeftModel shell: the returned object is the same model, so `.unet`, `.sample`, `.forward` keep working unchanged."""
    for p in model.parameters():
        p.requires_grad_(False)
    hits = [(n, m) for n, m in model.named_modules() if _matches(n, config.target_modules) and isinstance(m, (nn.Conv1d, nn.Linear))]
    if not hits:
        raise ValueError(f"Target modules {config.target_modules} not found in the base model.")
    for name, mod in hits:
        parent = model.get_submodule(name.rsplit(".", 1)[0]) if "." in name else model
        cls = LoraConv1d if isinstance(mod, nn.Conv1d) else LoraLinear
        wrapped = cls(mod, adapter_name, r=config.r, lora_alpha=config.lora_alpha, lora_dropout=config.lora_dropout,
                      init_lora_weights=config.init_lora_weights, use_rslora=config.use_rslora, use_dora=config.use_dora)
        setattr(parent, name.rsplit(".", 1)[-1], wrapped)
    Fn.bump_weight_epoch()
    return model


def lora_modules(model: nn.Module) -> List[_LoraBase]:
    return [m for m in model.modules() if isinstance(m, _LoraBase)]


def trainable_parameter_counts(model: nn.Module):
    """(trainable, all) as PeftModel.print_trainable_parameters reports them."""
    t = sum(p.numel() for p in model.parameters() if p.requires_grad)
    return t, sum(p.numel() for p in model.parameters())


_ADAPTER_MARKERS = ("lora_A", "lora_B", "lora_magnitude_vector")


def get_adapter_state_dict(model: nn.Module, adapter_name: str = "default") -> Dict[str, torch.Tensor]:
    """Adapter tensors under the key names `peft_model.save_pretrained` (peft 0.12 `get_peft_model_state_dict`) writes:
    `base_model.model.` prefix, adapter name removed, and the DoRA magnitude stored as `...lora_magnitude_vector` (no `.weight`)."""
    out = {}
    for k, v in model.state_dict().items():
        if not any(f".{m}.{adapter_name}" in k for m in _ADAPTER_MARKERS):
            continue
        if k.endswith(f"lora_magnitude_vector.{adapter_name}.weight"):
            k = k[: -len(".weight")]
        out["base_model.model." + k.replace(f".{adapter_name}", "")] = v
    return out


def load_adapter_state_dict(model: nn.Module, state: Dict[str, torch.Tensor], adapter_name: str = "default") -> None:
    own = model.state_dict()
    mapped = {}
    for k, v in state.items():
        k = k[len("base_model.model."):] if k.startswith("base_model.model.") else k
        if k.endswith(".lora_magnitude_vector"):
            k = f"{k}.{adapter_name}.weight"
        else:
            for marker in _ADAPTER_MARKERS:
                if f".{marker}." in k and f".{marker}.{adapter_name}." not in k:
                    k = k.replace(f".{marker}.", f".{marker}.{adapter_name}.", 1)
                    break
        if k not in own:
            raise KeyError(f"unexpected adapter key {k}")
        mapped[k] = v
    model.load_state_dict(mapped, strict=False)
    Fn.bump_weight_epoch()


def merge_and_unload(model: nn.Module) -> nn.Module:
    """PeftModel.merge_and_unload (trainer_peft.py:161-164): fold every adapter into its base layer and restore the plain module
    tree, so `state_dict()` is the reference's 1,239-key layout again."""
    for name, mod in list(model.named_modules()):
        if isinstance(mod, _LoraBase):
            mod.merge()
            parent = model.get_submodule(name.rsplit(".", 1)[0]) if "." in name else model
            setattr(parent, name.rsplit(".", 1)[-1], mod.base_layer)
    Fn.bump_weight_epoch()
    return model
