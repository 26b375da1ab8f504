"""Token-index work of the padding-free / sample-packing path; mirror of unsloth/utils/packing.py.

Integer-only host logic (a19 in SURVEY 8): these run as torch integer ops exactly like the
reference (there is no Triton kernel behind them), on whatever device the labels live on, and
must be BIT-EXACT -- pinned by the reference's own vectors (tests/utils/test_packing.py:135-157,
1375-1439, 1488-1524), restated in tests/test_packing.py.

  get_packed_info_from_kwargs        packing.py:586-606   lengths int32, cu_seqlens = [0, cumsum], max_seqlen
  mask_packed_sequence_boundaries    packing.py:710-730   shifted labels: flat[cumsum-1] = -100 (in place)
  mask_packed_boundary_labels        packing.py:733-772   raw labels: flat[cumsum] = -100, out-of-range -> index 0,
                                                          OUT OF PLACE (caller's batch never mutated)
  build_sdpa_packed_attention_mask   packing.py:650-693   block-diagonal causal additive mask
  enable_padding_free_metadata       packing.py:241-284   collator side: position_ids + packed_seq_lengths
"""
from typing import Any, Optional, Tuple

import torch

_PACKED_INFO_CACHE = {}
_SDPA_MASK_CACHE = {}


def get_packed_info_from_kwargs(kwargs: dict, device) -> Optional[Tuple[torch.Tensor, torch.Tensor, int]]:
    seq_lengths = kwargs.get("packed_seq_lengths")
    if seq_lengths is None:
        return None
    entry = _PACKED_INFO_CACHE.get(device)
    if entry is not None and entry["seq_lengths"] is seq_lengths:
        return entry["result"]
    lengths = seq_lengths.to(device=device, dtype=torch.int32, non_blocking=True)
    cu_seqlens = torch.zeros(lengths.numel() + 1, dtype=torch.int32, device=device)
    torch.cumsum(lengths, dim=0, dtype=torch.int32, out=cu_seqlens[1:])
    max_seqlen = int(lengths.max().item())
    result = (lengths, cu_seqlens, max_seqlen)
    _PACKED_INFO_CACHE[device] = {"seq_lengths": seq_lengths, "result": result}
    return result


def build_sdpa_packed_attention_mask(seq_info, *, dtype, device, sliding_window=None):
    seq_lengths, _, _ = seq_info
    params = (dtype, sliding_window)
    entry = _SDPA_MASK_CACHE.get(device)
    if entry is not None and entry["seq_lengths"] is seq_lengths and entry["params"] == params:
        return entry["mask"]
    lengths = seq_lengths.to("cpu", torch.int64)
    total = int(lengths.sum().item())
    # vectorised form of the reference's per-document loop: same document AND causal (AND window)
    doc = torch.repeat_interleave(torch.arange(lengths.numel()), lengths).to(device)
    pos = torch.arange(total, device=device)
    allowed = (doc[:, None] == doc[None, :]) & (pos[:, None] >= pos[None, :])
    if sliding_window is not None and sliding_window > 0:
        allowed &= (pos[:, None] - pos[None, :]) < sliding_window
    mask = torch.full((total, total), float("-inf"), dtype=dtype, device=device)
    mask.masked_fill_(allowed, 0.0)
    result = mask.unsqueeze(0).unsqueeze(0)
    _SDPA_MASK_CACHE[device] = {"seq_lengths": seq_lengths, "params": params, "mask": result}
    return result


def _normalize_packed_lengths(seq_lengths: Any, *, device) -> Optional[torch.Tensor]:
    if seq_lengths is None:
        return None
    if isinstance(seq_lengths, torch.Tensor):
        lengths = seq_lengths.to(device=device, dtype=torch.int64)
    else:
        lengths = torch.tensor(seq_lengths, device=device, dtype=torch.int64)
    if lengths.ndim != 1:
        lengths = lengths.reshape(-1)
    if lengths.numel() == 0:
        return None
    return lengths


def mask_packed_sequence_boundaries(shift_labels, seq_lengths, *, ignore_index: int = -100) -> bool:
    """Mark the final token of every packed sample in ALREADY SHIFTED labels (in place)."""
    lengths = _normalize_packed_lengths(seq_lengths, device=shift_labels.device)
    if lengths is None:
        return False
    flat = shift_labels.reshape(-1)
    total_tokens = flat.shape[0]
    boundary_positions = torch.cumsum(lengths, dim=0) - 1
    valid = boundary_positions < total_tokens
    if not torch.all(valid):
        boundary_positions = boundary_positions[valid]
    if boundary_positions.numel() == 0:
        return False
    flat[boundary_positions] = ignore_index
    return True


def mask_packed_boundary_labels(labels, seq_lengths, *, ignore_index: int = -100):
    """Same guard on RAW labels, out of place, for the fused CE that shifts internally:
    masks labels[cumsum(lengths)]; out-of-range positions are redirected to index 0 (which the
    shift discards)."""
    if labels is None or not isinstance(labels, torch.Tensor):
        return labels
    lengths = _normalize_packed_lengths(seq_lengths, device=labels.device)
    if lengths is None:
        return labels
    total_tokens = labels.numel()
    if total_tokens == 0:
        return labels
    positions = torch.cumsum(lengths, dim=0)
    positions = torch.where(positions < total_tokens, positions, torch.zeros_like(positions))
    flat = labels.reshape(-1).index_fill(0, positions, ignore_index)
    return flat.view(labels.shape)


def packed_position_ids(seq_lengths, device=None) -> torch.Tensor:
    """positions == concat(arange(len)) (tests/utils/test_packing.py:1095-1117), int32 like TRL's."""
    lengths = _normalize_packed_lengths(seq_lengths, device=device or "cpu")
    starts = torch.cumsum(lengths, 0) - lengths
    total = int(lengths.sum())
    return (torch.arange(total, device=lengths.device) - torch.repeat_interleave(starts, lengths)).to(torch.int32)


def enable_padding_free_metadata(batch_input_ids, device=None):
    """Collator-side packing of a list of token-id lists into ONE row (batch dim 1), emitting what the
    model forward consumes (packing.py:241-284): input_ids, labels (first token of every document
    = -100, as TRL's padding-free collator), position_ids (int32, restarting), packed_seq_lengths."""
    lengths = torch.tensor([len(x) for x in batch_input_ids], dtype=torch.int32)
    ids = torch.cat([torch.as_tensor(x, dtype=torch.int64) for x in batch_input_ids]).unsqueeze(0)
    pos = packed_position_ids(lengths).unsqueeze(0)
    labels = ids.clone()
    labels[pos == 0] = -100
    out = dict(input_ids=ids, labels=labels, position_ids=pos, packed_seq_lengths=lengths)
    if device is not None:
        out = {k: v.to(device) for k, v in out.items()}
    return out


def clear_packed_caches():
    _PACKED_INFO_CACHE.clear()
    _SDPA_MASK_CACHE.clear()


__all__ = [
    "get_packed_info_from_kwargs", "build_sdpa_packed_attention_mask", "mask_packed_sequence_boundaries",
    "mask_packed_boundary_labels", "packed_position_ids", "enable_padding_free_metadata", "clear_packed_caches",
]
