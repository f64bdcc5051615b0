"""HIP-graph replay of the UNet call ("whole UNetModel.forward as one HIP-graph replay per step").

One denoising step issues ~390 kernel launches from Python; the kernels are 5-400 us each, so host launch cost and
inter-kernel gaps were ~12 % of the step (rocprofv3: kernel time vs wall).  `GraphedModule` captures the module's
forward once per input signature into a hipGraph (through torch.cuda.CUDAGraph, which owns the capture stream and
the graph-private allocator pool) and replays it: inputs are copied into static buffers, the output is read from a
static buffer.  All crg_* launches go to `torch.cuda.current_stream()`, i.e. into the capture.

Correctness rules enforced here:
  * the graph is keyed on the shapes/dtypes of the tensor arguments AND on the identity + version of the tensors the
    forward treats as constants across steps (the conditioning: its cross-attention K / V^T projections are cached
    inside the modules and therefore NOT part of the captured work); a new prompt -> a new capture;
  * two eager warm-up calls run before capture so that every weight pack, cache fill, hipFuncSetAttribute and scratch
    growth happens outside the capture; the context scratch is additionally reserved up front;
  * a graph OWNS every step-invariant tensor it was captured against that lives outside its allocator pool: the
    conditioning, the per-module cross-attention K / V^T caches and the cast copy of the context (all written by the
    eager warm-up).  They are snapshotted into the graph record after warm-up, so a later capture for another prompt,
    which overwrites the modules' single-slot caches, cannot free memory an older graph still reads
    (ctx A -> ctx B -> ctx A replays graph A against its own, still-live K/V);
  * a graph is keyed on a WEIGHT STAMP as well: the (data_ptr, _version) of every parameter of the module (about 0.1 ms of
    host time per call, overlapped with the previous replay; the parameter list is re-read when `TensorKeyedCache.epoch`
    moves, i.e. when a Parameter is registered on any module - LoRA `setattr`, image_generator.py:408-453 - or a derived
    image is rebuilt) plus `TensorKeyedCache.generation`, which moves whenever a derived-weight cache DROPS its values
    (`ops.clear_weight_cache()`, the overflow clear): the packed / stacked / merged images a capture baked in are freed there
    although no parameter changed.  load_state_dict (in-place copy -> version), `.to()` / `.half()` (new storage ->
    data_ptr), LoRA swaps and cache clears therefore re-capture instead of replaying against stale or freed weight images;
    `invalidate()` remains for writes through `.data`, which no counter sees;
  * a failed capture RAISES (`strict=True`, the default): the caller asked for replay, and a silent switch to eager
    launches would make a timing unattributable.  `strict=False` restores the warn-and-go-eager behaviour.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import torch

from . import _lib as L
from .ops import TensorKeyedCache


def _on_parameter_registered(module, name, param):
    TensorKeyedCache.epoch += 1
    return None


torch.nn.modules.module.register_module_parameter_registration_hook(_on_parameter_registered)


class GraphedModule:
    def __init__(self, module: torch.nn.Module, const_args: Tuple[str, ...] = ("context",), scratch_bytes: int = 1 << 30,
                 max_graphs: int = 4, strict: bool = True):
        self.module = module
        self.const_args = const_args
        self.scratch_bytes = scratch_bytes
        self.max_graphs = max_graphs
        self.strict = strict
        self._graphs: Dict[tuple, tuple] = {}
        self.broken = False
        self.replays = 0     # launches of an ALREADY captured graph (the call that captures is not counted: bench.py checks that its timed steps were replays)
        self.captures = 0
        self._params = None  # (epoch, [parameters]) - the list is rebuilt when the epoch moves

    def _weight_stamp(self):
        if self._params is None or self._params[0] != TensorKeyedCache.epoch:
            self._params = (TensorKeyedCache.epoch, list(self.module.parameters()))
        return hash((TensorKeyedCache.generation,) + tuple((q.data_ptr(), q._version) for q in self._params[1]))

    @property
    def active(self) -> bool:
        return not self.broken

    def _held_state(self):
        """Every step-invariant tensor the module tree caches OUTSIDE the graph pool (see the module docstring)."""
        held = []
        for m in self.module.modules():
            kv = getattr(m, "_kv", None)
            if kv is not None:
                held.append(kv)
            cc = getattr(m, "_ctx_cast", None)
            if cc is not None:
                held.append(cc)
        return held

    def invalidate(self):
        self._graphs.clear()

    @staticmethod
    def _sig(t):
        return (tuple(t.shape), t.dtype, t.device) if torch.is_tensor(t) else t

    def __call__(self, x, **kw):
        consts = {k: kw[k] for k in self.const_args if kw.get(k) is not None}
        dyn = {k: v for k, v in kw.items() if k not in consts and torch.is_tensor(v)}
        other = {k: v for k, v in kw.items() if k not in consts and not torch.is_tensor(v)}
        dup = bool(getattr(x, "_crg_cfg_dup", False))  # ops.mark_cfg_dup: a different kernel sequence, hence a different graph
        # ops.attach_time_rows: the rows riding on a dynamic argument are a dynamic input of their own (a graph captured without
        # them computes the embedding itself - another kernel sequence)
        rows = {k: v._crg_time_rows for k, v in dyn.items() if getattr(v, "_crg_time_rows", None) is not None}
        key = (self._sig(x), dup, tuple((k, self._sig(v), id(rows[k][0]) if k in rows else None, self._sig(rows[k][1]) if k in rows else None)
                                        for k, v in sorted(dyn.items())),
               tuple((k, id(v), v._version, self._sig(v)) for k, v in sorted(consts.items())), tuple(sorted(other.items())))
        if self.broken:
            return self.module(x, **kw)
        g = self._graphs.get(key)
        if g is not None and g[5] != self._weight_stamp():  # weights changed since the capture: its packed images are stale
            del self._graphs[key]
            g = None
        fresh = g is None
        if g is None:
            try:
                g = self._capture(key, x, dyn, consts, other, rows)
            except Exception as e:
                if self.strict:
                    raise L.CrgError(f"hipGraph capture failed ({type(e).__name__}: {e})") from e
                import warnings
                warnings.warn(f"hipGraph capture failed ({type(e).__name__}: {e}); continuing with eager launches")
                self.broken = True
                torch.cuda.synchronize(x.device)
                return self.module(x, **kw)
        graph, sx, sdyn, sout = g[:4]
        sx.copy_(x)
        for k, v in dyn.items():
            sdyn[k].copy_(v)
            if k in rows:
                sdyn[k]._crg_time_rows[1].copy_(rows[k][1])
        graph.replay()
        if not fresh:
            self.replays += 1
        return sout.clone()

    def _capture(self, key, x, dyn, consts, other, rows):
        if len(self._graphs) >= self.max_graphs:
            self._graphs.clear()
        dev = x.device
        h = L.ctx(dev.index if dev.index is not None else torch.cuda.current_device())
        L.check(L.load().crg_ctx_reserve(h, C.c_size_t(self.scratch_bytes)), h, "crg_ctx_reserve")
        sx = x.clone()
        if getattr(x, "_crg_cfg_dup", False):
            sx._crg_cfg_dup = True
        sdyn = {k: v.clone() for k, v in dyn.items()}
        for k, (owner, r) in rows.items():
            sdyn[k]._crg_time_rows = (owner, r.clone())
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s), torch.no_grad():
            for _ in range(2):  # warm-up: packs weights, fills K/V and embedding caches, sets kernel attributes
                self.module(sx, **sdyn, **consts, **other)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            sout = self.module(sx, **sdyn, **consts, **other)
        # held: the conditioning and the modules' K / V^T / cast-context caches as they are RIGHT NOW (filled by the warm-up,
        # read by the captured kernels); stamp: the weights this capture baked in
        g = (graph, sx, sdyn, sout, (consts, self._held_state()), self._weight_stamp())
        self._graphs[key] = g
        self.captures += 1
        return g
