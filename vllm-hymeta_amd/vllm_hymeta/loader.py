"""Checkpoint side of the hot path: where a tensor of the published checkpoints goes, and how the
bf16 / GPTQ / W8A8 tensors are cut for one tensor-parallel (and expert-parallel) rank.

Name routing restates ``HymetaForCausalLM.load_weights``
(``vllm_hymeta/models/modeling_hymeta.py:1487-1691``) and ``GLAswaForCausalLM.load_weights``
(``vllm_hymeta/model_for_7B/modeling_gla_swa.py:601-724``): the same classification order, the
same renames (``q_proj|k_proj|v_proj -> qkv_proj`` with shard ids q/k/v, ``gate_proj|up_proj ->
gate_up_proj`` with shard ids 0/1, ``gk_proj.0/1 -> gk_proj.linear0/1`` in the 7B model,
``experts.<e>.gate_proj|up_proj|down_proj -> experts.w13_/w2_`` with vLLM's shard ids w1/w3/w2,
the ``mode.`` typo of the 70B checkpoint), layers past ``num_hidden_layers`` and
``rotary_emb.inv_freq`` dropped.

Sharding restates what vLLM's parallel linears do with those tensors (third party, vllm==0.10.0:
``QKVParallelLinear`` / ``MergedColumnParallelLinear`` / ``RowParallelLinear`` weight loaders and
their GPTQ / compressed-tensors parameter classes), expressed on the tensor formats the HIP
kernels read directly, so that a shard is again a plain ``HymetaLinear``:

* output features (rows of a bf16 / int8 weight ``[out, in]``, columns of a GPTQ ``qweight
  [in / pack, out]``) are split for column-parallel layers;
* input features are split for row-parallel layers: for GPTQ that is rows of ``qweight`` and rows
  of ``qzeros`` / ``scales``, which must fall on group boundaries;
* q/k/v are cut by HEAD with the kv-head replication rule of ``parallel.shard_heads``.
"""

from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Dict, Optional, Sequence, Tuple

import torch

from .parallel import HeadShard, shard_heads

_QKV = (("q_proj", "q"), ("k_proj", "k"), ("v_proj", "v"))
_EXPERT_RE = re.compile(r"model\.layers\.(\d+)\.block_sparse_moe\.experts.(\d+)\.")
# vLLM FusedMoE.make_expert_params_mapping: checkpoint projection -> (fused parameter, shard id)
_EXPERT_PROJ = {"gate_proj": ("w13_", "w1"), "up_proj": ("w13_", "w3"), "down_proj": ("w2_", "w2")}


@dataclass(frozen=True)
class Route:
    """Destination of one checkpoint tensor: parameter name in the serving model, and which part of
    a fused parameter it fills (``shard_id``: q/k/v, 0/1 for gate/up, w1/w2/w3 inside an expert)."""
    param: str
    shard_id: Optional[object] = None
    expert_id: Optional[int] = None


def _layer_of(name: str) -> Optional[int]:
    if "layers" not in name:
        return None
    return int(name.split("layers")[-1].split(".")[1])


def _route_mlp(name: str) -> Route:
    if "gate_proj" in name:
        return Route(name.replace("gate_proj", "gate_up_proj", 1), 0)
    if "up_proj" in name:
        return Route(name.replace("up_proj", "gate_up_proj", 1), 1)
    if "down_proj" in name:
        return Route(name)
    raise AssertionError("MLP weight not in [gate_up_proj, down_proj]")


def route_70b(name: str, num_hidden_layers: int, known_params: Optional[Sequence[str]] = None) -> Optional[Route]:
    """``HymetaForCausalLM.load_weights`` routing.  ``known_params`` (the serving model's parameter
    names) is what the reference's norm test consults (``"norm" in name and name in params_dict``);
    without it every ``norm`` tensor that is not a bias counts as a norm weight."""
    layer = _layer_of(name)
    if layer is not None and layer >= num_hidden_layers:
        return None
    if "mode." in name:   # the checkpoint stores model.norm as "mode.norm"
        name = name.replace("mode.", "model.")
    if "norm" in name and not name.endswith(".bias") and (known_params is None or name in known_params):
        return Route(name)
    if "vanilla_attn" in name or "linear_attn" in name:
        for proj, shard in _QKV:
            if proj in name:
                return Route(name.replace(proj, "qkv_proj"), shard)
        return Route(name)
    if "block_sparse_moe" in name and not name.endswith(".bias"):
        m = _EXPERT_RE.search(name)
        if m:
            expert = int(m.group(2))
            for proj, (fused, shard) in _EXPERT_PROJ.items():
                weight_name = f"experts.{expert}.{proj}."
                if weight_name in name:
                    return Route(name.replace(weight_name, f"experts.{fused}"), shard, expert)
        return Route(name)   # the router gate
    if ("shared_mlp" in name and not name.endswith(".bias")) or ".mlp." in name:
        return _route_mlp(name)
    if "rotary_emb.inv_freq" in name:
        return None
    return Route(name)


def route_7b(name: str, num_hidden_layers: int, known_params: Optional[Sequence[str]] = None) -> Optional[Route]:
    """``GLAswaForCausalLM.load_weights`` routing (MLP first, then norms, then attention)."""
    layer = _layer_of(name)
    if layer is not None and layer >= num_hidden_layers:
        return None
    if ".mlp." in name:
        return _route_mlp(name)
    if "norm" in name and not name.endswith(".bias") and (known_params is None or name in known_params):
        return Route(name)
    if "attn" in name:
        for proj, shard in _QKV:
            if proj in name and "gk_proj" not in name:
                return Route(name.replace(proj, "qkv_proj"), shard)
        if "gk_proj.0." in name:
            name = name.replace("gk_proj.0.", "gk_proj.linear0.")
        elif "gk_proj.1" in name:
            name = name.replace("gk_proj.1.", "gk_proj.linear1.")
        return Route(name)
    return Route(name)


# ------------------------------------------------------------------------------------------------
# Sharding.  A linear is described by a dict of tensors in checkpoint format:
#   bf16 : {"weight" [out, in], "bias" [out]?}
#   gptq : {"qweight" int32 [in/pack, out], "qzeros" int32 [in/group, out/pack], "scales" [in/group, out],
#           "g_idx" int32 [in]?, "bias"?}           (bits, group_size passed alongside)
#   w8a8 : {"weight" int8 [out, in], "weight_scale" fp32 [out] or [out, 1], "bias"?}
# ------------------------------------------------------------------------------------------------
def linear_kind(t: Dict[str, torch.Tensor]) -> str:
    if "qweight" in t:
        return "gptq"
    if t["weight"].dtype == torch.int8:
        return "w8a8"
    return "bf16"


def check_g_idx(g_idx: Optional[torch.Tensor], group_size: int) -> None:
    """The kernels read groups in natural order; an act-order checkpoint is refused, not mis-read."""
    if g_idx is None:
        return
    expect = torch.arange(g_idx.numel(), dtype=g_idx.dtype, device=g_idx.device) // group_size
    if not torch.equal(g_idx, expect):
        raise ValueError("GPTQ act-order (g_idx != k // group_size) is not supported")


def slice_out_features(t: Dict[str, torch.Tensor], lo: int, hi: int, bits: int = 4) -> Dict[str, torch.Tensor]:
    """Output features [lo, hi) of a linear (a column-parallel shard, or one head range)."""
    kind = linear_kind(t)
    out = {}
    if kind == "gptq":
        pack = 32 // bits
        if lo % pack or hi % pack:
            raise ValueError(f"GPTQ output split [{lo}, {hi}) must fall on multiples of {pack} (packed zeros)")
        out["qweight"] = t["qweight"][:, lo:hi]
        out["qzeros"] = t["qzeros"][:, lo // pack:hi // pack]
        out["scales"] = t["scales"][:, lo:hi]
        if "g_idx" in t:
            out["g_idx"] = t["g_idx"]
    else:
        out["weight"] = t["weight"][lo:hi]
        if kind == "w8a8":
            out["weight_scale"] = t["weight_scale"].reshape(-1)[lo:hi]
    if t.get("bias") is not None:
        out["bias"] = t["bias"][lo:hi]
    return out


def slice_in_features(t: Dict[str, torch.Tensor], lo: int, hi: int, bits: int = 4, group_size: int = 128,
                      keep_bias: bool = True) -> Dict[str, torch.Tensor]:
    """Input features [lo, hi) of a linear (a row-parallel shard).  The bias belongs to ONE rank
    (``keep_bias``), the others add nothing, so that the all-reduce sums it once."""
    kind = linear_kind(t)
    out = {}
    if kind == "gptq":
        pack = 32 // bits
        if lo % group_size or hi % group_size:
            raise ValueError(f"GPTQ input split [{lo}, {hi}) must fall on group boundaries ({group_size})")
        check_g_idx(t.get("g_idx"), group_size)
        out["qweight"] = t["qweight"][lo // pack:hi // pack]
        out["qzeros"] = t["qzeros"][lo // group_size:hi // group_size]
        out["scales"] = t["scales"][lo // group_size:hi // group_size]
    else:
        out["weight"] = t["weight"][:, lo:hi]
        if kind == "w8a8":
            out["weight_scale"] = t["weight_scale"].reshape(-1)
    if keep_bias and t.get("bias") is not None:
        out["bias"] = t["bias"]
    return out


# ------------------------------------------------------------------------------------------------
# The same cuts one tensor at a time, for loading as a stream: a checkpoint tensor is cut for the rank
# the moment it arrives and only the rank's piece is kept (on the device), so that loading never holds
# more than the rank's share of the model plus the tensor in flight.
# ------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Cut:
    """How the tensors of one linear are cut for a rank: along the output features (``axis="out"``: the explicit range
    ``[lo, hi)`` -- a head range -- or share ``index`` of ``parts`` equal shares), along the input features
    (``axis="in"``: share ``index`` of ``parts``), or not at all (``axis="none"``)."""
    axis: str = "none"
    lo: Optional[int] = None
    hi: Optional[int] = None
    parts: int = 1
    index: int = 0


def _share(n: int, cut: Cut, what: str) -> Tuple[int, int]:
    if n % cut.parts:
        raise ValueError(f"{what} {n} must be divisible by tensor-parallel size {cut.parts}")
    per = n // cut.parts
    return cut.index * per, (cut.index + 1) * per


def cut_leaf(leaf: str, t: torch.Tensor, cut: Cut, bits: int = 4, group_size: int = 128) -> Optional[torch.Tensor]:
    """One checkpoint tensor of a linear (``leaf``: weight / bias / weight_scale / qweight / qzeros / scales / g_idx) ->
    the rank's piece (a view), or ``None`` when the rank keeps nothing of it (the bias of a row-parallel linear on
    ranks other than 0, which the all-reduce must sum once; ``g_idx``, checked here and not needed afterwards)."""
    pack = 32 // bits
    if leaf == "g_idx":
        check_g_idx(t, group_size)
        return None
    if cut.axis == "none":
        return t.reshape(-1) if leaf == "weight_scale" else t
    if cut.axis == "out":
        if cut.lo is not None:
            lo, hi = cut.lo, cut.hi
        else:
            n = {"weight": t.shape[0], "qweight": t.shape[-1], "scales": t.shape[-1], "qzeros": t.shape[-1] * pack}.get(leaf, t.numel())
            lo, hi = _share(n, cut, "output size")
        if leaf in ("weight", "bias"):
            return t[lo:hi]
        if leaf == "weight_scale":
            return t.reshape(-1)[lo:hi]
        if leaf in ("qweight", "scales"):
            return t[:, lo:hi]
        if leaf == "qzeros":
            if lo % pack or hi % pack:
                raise ValueError(f"GPTQ output split [{lo}, {hi}) must fall on multiples of {pack} (packed zeros)")
            return t[:, lo // pack:hi // pack]
        raise ValueError(f"unknown linear tensor {leaf!r}")
    if cut.axis != "in":
        raise ValueError(f"unknown cut axis {cut.axis!r}")
    if leaf == "bias":
        return t if cut.index == 0 else None
    if leaf == "weight_scale":
        return t.reshape(-1)
    if leaf == "weight":
        lo, hi = _share(t.shape[1], cut, "input size")
        return t[:, lo:hi]
    k_total = t.shape[0] * (pack if leaf == "qweight" else group_size)
    lo, hi = _share(k_total, cut, "input size")
    if lo % group_size or hi % group_size:
        raise ValueError(f"GPTQ input split [{lo}, {hi}) must fall on group boundaries ({group_size})")
    if leaf == "qweight":
        return t[lo // pack:hi // pack]
    if leaf in ("qzeros", "scales"):
        return t[lo // group_size:hi // group_size]
    raise ValueError(f"unknown linear tensor {leaf!r}")


def own(piece: torch.Tensor, device=None) -> torch.Tensor:
    """``piece`` on ``device`` in storage of its own size: a view into a whole checkpoint tensor would keep all of it
    alive."""
    piece = piece.contiguous()
    if device is not None:
        piece = piece.to(device)
    if piece.untyped_storage().nbytes() > piece.numel() * piece.element_size():
        piece = piece.clone()
    return piece


def head_cuts(shard: HeadShard, head_dim: int) -> Dict[str, Cut]:
    """Output-feature cuts of the q / k / v projections for a head shard."""
    kv = Cut("out", shard.kv_head_start * head_dim, (shard.kv_head_start + shard.num_kv_heads) * head_dim)
    return {"q": Cut("out", shard.head_start * head_dim, (shard.head_start + shard.num_heads) * head_dim), "k": kv, "v": kv}


def concat_out_features(parts: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """Fuse linears that share their input along the output axis (q|k|v -> qkv, gate|up -> gate_up)."""
    kind = linear_kind(parts[0])
    out = {}
    if kind == "gptq":
        for key in ("qweight", "qzeros", "scales"):
            out[key] = torch.cat([p[key] for p in parts], dim=1).contiguous()
    else:
        out["weight"] = torch.cat([p["weight"] for p in parts], dim=0).contiguous()
        if kind == "w8a8":
            out["weight_scale"] = torch.cat([p["weight_scale"].reshape(-1) for p in parts]).contiguous()
    biases = [p.get("bias") for p in parts]
    if any(b is not None for b in biases):
        ref = parts[0]["scales"] if kind == "gptq" else None
        filled = []
        for p, b in zip(parts, biases):
            n = p["qweight"].shape[1] if kind == "gptq" else p["weight"].shape[0]
            filled.append(b if b is not None else torch.zeros(n, dtype=torch.bfloat16, device=(ref if ref is not None else p["weight"]).device))
        out["bias"] = torch.cat(filled).contiguous()
    return out


def shard_qkv(q: Dict[str, torch.Tensor], k: Dict[str, torch.Tensor], v: Dict[str, torch.Tensor], total_heads: int,
              total_kv_heads: int, head_dim: int, tp_size: int, tp_rank: int, bits: int = 4
              ) -> Tuple[Dict[str, torch.Tensor], HeadShard]:
    """The fused qkv projection of one rank: its q-heads, then its kv-heads of k, then of v (a
    kv-head is replicated on ``tp / KVH`` ranks when there are fewer kv-heads than ranks)."""
    sh = shard_heads(total_heads, total_kv_heads, tp_size, tp_rank)
    qs = slice_out_features(q, sh.head_start * head_dim, (sh.head_start + sh.num_heads) * head_dim, bits)
    ks = slice_out_features(k, sh.kv_head_start * head_dim, (sh.kv_head_start + sh.num_kv_heads) * head_dim, bits)
    vs = slice_out_features(v, sh.kv_head_start * head_dim, (sh.kv_head_start + sh.num_kv_heads) * head_dim, bits)
    return concat_out_features([qs, ks, vs]), sh


def shard_gate_up(gate: Dict[str, torch.Tensor], up: Dict[str, torch.Tensor], tp_size: int, tp_rank: int, bits: int = 4
                  ) -> Dict[str, torch.Tensor]:
    """``gate_up_proj`` of one rank: its slice of gate rows, then the same slice of up rows."""
    inter = gate["qweight"].shape[1] if "qweight" in gate else gate["weight"].shape[0]
    if inter % tp_size:
        raise ValueError(f"intermediate size {inter} must be divisible by tensor-parallel size {tp_size}")
    per = inter // tp_size
    lo, hi = tp_rank * per, (tp_rank + 1) * per
    return concat_out_features([slice_out_features(gate, lo, hi, bits), slice_out_features(up, lo, hi, bits)])


def shard_row_parallel(t: Dict[str, torch.Tensor], tp_size: int, tp_rank: int, bits: int = 4, group_size: int = 128
                       ) -> Dict[str, torch.Tensor]:
    """``o_proj`` / ``out_proj`` / ``down_proj`` of one rank: its slice of the input features."""
    k_total = t["qweight"].shape[0] * (32 // bits) if "qweight" in t else t["weight"].shape[1]
    if k_total % tp_size:
        raise ValueError(f"input size {k_total} must be divisible by tensor-parallel size {tp_size}")
    per = k_total // tp_size
    return slice_in_features(t, tp_rank * per, (tp_rank + 1) * per, bits, group_size, keep_bias=tp_rank == 0)


def stack_experts(experts: Sequence[Dict[str, torch.Tensor]], tp_size: int = 1, tp_rank: int = 0
                  ) -> Tuple[torch.Tensor, torch.Tensor]:
    """bf16 experts ``[{gate_proj, up_proj, down_proj}]`` (each ``[out, in]``) -> the fused tensors of
    the grouped GEMM: ``w13 [E, 2 I/tp, H]`` (gate rows then up rows) and ``w2 [E, H, I/tp]``.  Expert
    parallelism needs no cut here: a rank passes ``local_experts`` and reads only its slots."""
    w13, w2 = [], []
    for e in experts:
        inter = e["gate_proj"].shape[0]
        if inter % tp_size:
            raise ValueError(f"expert intermediate size {inter} must be divisible by tensor-parallel size {tp_size}")
        per = inter // tp_size
        lo, hi = tp_rank * per, (tp_rank + 1) * per
        w13.append(torch.cat([e["gate_proj"][lo:hi], e["up_proj"][lo:hi]], dim=0))
        w2.append(e["down_proj"][:, lo:hi])
    return torch.stack(w13).contiguous(), torch.stack(w2).contiguous()


def expert_leaf(param: str) -> Tuple[str, str]:
    """``...experts.w13_weight_scale`` -> (``"w13"``, ``"weight_scale"``): which fused expert parameter a routed name fills and
    which tensor of the projection it is (weight / weight_scale / qweight / qzeros / scales / g_idx)."""
    tail = param.rsplit("experts.", 1)[-1]
    fused, _, leaf = tail.partition("_")
    return fused, leaf


def stack_experts_w8a8(experts: Sequence[Dict[str, Dict[str, torch.Tensor]]]
                       ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """W8A8 experts ``[{gate_proj | up_proj | down_proj: {"weight" int8 [out, in], "weight_scale" [out] or [out, 1]}}]`` (one
    rank's pieces) -> ``w13 int8 [E, 2 I, H]`` (gate rows then up rows), ``w13_scale fp32 [E, 2 I]``, ``w2 int8 [E, H, I]``,
    ``w2_scale fp32 [E, H]``: the layouts the grouped int8 GEMM reads."""
    w13 = torch.stack([torch.cat([e["gate_proj"]["weight"], e["up_proj"]["weight"]], dim=0) for e in experts]).contiguous()
    s13 = torch.stack([torch.cat([e["gate_proj"]["weight_scale"].reshape(-1), e["up_proj"]["weight_scale"].reshape(-1)])
                       for e in experts]).float().contiguous()
    w2 = torch.stack([e["down_proj"]["weight"] for e in experts]).contiguous()
    s2 = torch.stack([e["down_proj"]["weight_scale"].reshape(-1) for e in experts]).float().contiguous()
    return w13, s13, w2, s2


def stack_experts_gptq(experts: Sequence[Dict[str, Dict[str, torch.Tensor]]], bits: int = 4, group_size: int = 128,
                       zero_offset: int = 1):
    """GPTQ experts ``[{gate_proj | up_proj | down_proj: {"qweight", "qzeros", "scales"}}]`` (one rank's pieces, cut on packing
    and group boundaries) -> two ``hip_ops.GptqExperts`` stacks standing for ``w13 [E, 2 I, H]`` (gate columns then up
    columns of every packed row) and ``w2 [E, H, I]``."""
    from .hip_ops import GptqExperts

    fused = [concat_out_features([e["gate_proj"], e["up_proj"]]) for e in experts]
    down = [e["down_proj"] for e in experts]

    def stack(parts):
        return GptqExperts(*(torch.stack([p[leaf] for p in parts]).contiguous() for leaf in ("qweight", "qzeros", "scales")),
                           bits, group_size, zero_offset)

    return stack(fused), stack(down)


def make_linear(t: Dict[str, torch.Tensor], bits: int = 4, group_size: int = 128, zero_offset: int = 1, device=None):
    """Checkpoint-format tensors (one rank's) -> ``HymetaLinear`` on ``device``."""
    from .linear import HymetaLinear

    def mv(x):
        return None if x is None else x.contiguous().to(device) if device is not None else x.contiguous()

    def bf16(x):
        # gptqmodel / llmcompressor checkpoints store biases (and the layers they leave un-quantised) in the
        # checkpoint's float type, often fp16; the kernels take bf16 activations, weights and biases
        return None if x is None else x.to(torch.bfloat16) if x.is_floating_point() else x

    kind = linear_kind(t)
    bias = mv(bf16(t.get("bias")))
    if kind == "gptq":
        check_g_idx(t.get("g_idx"), group_size)
        return HymetaLinear.from_gptq(mv(t["qweight"]), mv(t["qzeros"]), mv(t["scales"]), bits, group_size, bias=bias,
                                      zero_offset=zero_offset)
    if kind == "w8a8":
        return HymetaLinear.from_w8a8(mv(t["weight"]), mv(t["weight_scale"].reshape(-1).float()), bias=bias)
    return HymetaLinear.from_weight(mv(bf16(t["weight"])), bias)
