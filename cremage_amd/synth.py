"""Name-keyed deterministic synthetic weights and inputs.

No checkpoint exists offline, so parity and the benchmark run on
synthetic parameters.  The fill is keyed on the *parameter name* (crc32(name) ^ seed
seeds a legacy numpy RandomState, whose stream is frozen across numpy versions), so

  * the oracle's golden fixtures (made from the reference modules) and the HIP
    modules get bit-identical fp32 parameters without shipping any tensor, and
  * a parameter-name mismatch with the reference's checkpoint contract
    (image_generator.py:345, `load_state_dict(strict=False)`) shows up as a parity
    failure instead of being silently dropped.

Scales: matrices/conv kernels ~ N(0, 1/fan_in); biases ~ 0.05*N(0,1); 1-D "weight"
(norm gains) ~ 1 + 0.1*N(0,1).  Zero-initialised tensors of the reference
(`zero_module`, openaimodel.py:233-235,755; attention.py:1002) are re-randomised like
any other, otherwise every block would be an identity and parity vacuous.
"""
import zlib

import numpy as np
import torch


def _rs(name: str, seed: int) -> np.random.RandomState:
    return np.random.RandomState((zlib.crc32(name.encode("utf-8")) ^ (seed * 2654435761)) & 0xFFFFFFFF)


def synth_tensor(name: str, shape, seed: int, kind: str = "auto") -> torch.Tensor:
    """fp32 tensor for parameter `name` of `shape` (see module docstring for scales)."""
    shape = tuple(int(s) for s in shape)
    rs = _rs(name, seed)
    x = rs.standard_normal(shape).astype(np.float32)
    if kind == "auto":
        if len(shape) >= 2:
            kind = "matrix"
        elif name.endswith("bias"):
            kind = "bias"
        else:
            kind = "gain"
    if kind == "matrix":
        fan_in = int(np.prod(shape[1:]))
        x *= np.float32(1.0 / np.sqrt(fan_in))
    elif kind == "bias":
        x *= np.float32(0.05)
    elif kind == "gain":
        x = np.float32(1.0) + np.float32(0.1) * x
    elif kind == "normal":
        pass
    else:
        raise ValueError(kind)
    return torch.from_numpy(x)


@torch.no_grad()
def synth_fill_(module: torch.nn.Module, seed: int, prefix: str = "") -> torch.nn.Module:
    """Overwrite every >=1-D parameter of `module` in place (0-D LoRA alphas are kept)."""
    for name, p in module.named_parameters():
        if p.ndim == 0 or "_lora_" in name:
            continue
        p.copy_(synth_tensor(prefix + name, p.shape, seed).to(p.dtype))
    return module


def synth_input(name: str, shape, seed: int, scale: float = 1.0) -> torch.Tensor:
    """N(0, scale^2) input tensor keyed on (name, seed)."""
    return synth_tensor("input/" + name, shape, seed, kind="normal") * scale
