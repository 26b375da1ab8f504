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


def _document_ends(seq_lengths, device):
    """Exclusive end offset of every packed document (int64 running sum of the lengths) or None when there are
    no documents. Accepts a tensor or any sequence; any shape is read as a flat list."""
    if seq_lengths is None:
        return None
    ends = torch.as_tensor(seq_lengths, device=device).to(torch.int64).reshape(-1).cumsum(0)
    return ends if ends.numel() else None


def get_packed_info_from_kwargs(kwargs: dict, device) -> Optional[Tuple[torch.Tensor, torch.Tensor, int]]:
    """(lengths int32, cu_seqlens int32 = [0, running sum], max_seqlen) of the batch's `packed_seq_lengths`, or
    None for an unpacked batch. One entry per device is remembered for the tensor object last seen, so the 32
    layers of a forward (and the recompute of a checkpointed backward) share it."""
    src = kwargs.get("packed_seq_lengths")
    if src is None:
        return None
    hit = _PACKED_INFO_CACHE.get(device)
    if hit is not None and hit[0] is src:
        return hit[1]
    lengths = src.to(device=device, dtype=torch.int32, non_blocking=True)
    cu_seqlens = torch.nn.functional.pad(lengths.cumsum(0, dtype=torch.int32), (1, 0))
    info = (lengths, cu_seqlens, int(lengths.max()))
    _PACKED_INFO_CACHE[device] = (src, info)
    return info


def build_sdpa_packed_attention_mask(seq_info, *, dtype, device, sliding_window=None):
    seq_lengths, _, _ = seq_info
    params = (dtype, sliding_window)
    hit = _SDPA_MASK_CACHE.get(device)
    if hit is not None and hit[0] is seq_lengths and hit[1] == params:
        return hit[2]
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
    _SDPA_MASK_CACHE[device] = (seq_lengths, params, result)
    return result


def mask_packed_sequence_boundaries(shift_labels, seq_lengths, *, ignore_index: int = -100) -> bool:
    """ALREADY SHIFTED labels, in place: the last position of every document predicts the first token of the next
    one, so flat[end - 1] = ignore_index for every document end that lies inside the tensor (packing.py:710-730).
    Returns whether anything was written."""
    ends = _document_ends(seq_lengths, shift_labels.device)
    if ends is None:
        return False
    flat = shift_labels.reshape(-1)
    last = ends[ends <= flat.numel()] - 1
    if last.numel() == 0:
        return False
    flat.index_fill_(0, last, ignore_index)
    return True


def mask_packed_boundary_labels(labels, seq_lengths, *, ignore_index: int = -100):
    """RAW labels for the fused cross entropy, which shifts internally: the label at flat[end] (the first token of the
    following document) must not be predicted from the previous one. OUT OF PLACE -- the caller's batch is never
    touched (tests/utils/test_packing.py:1488-1524). An end at or past the tensor's size is redirected to flat
    index 0, which the shift drops anyway (packing.py:733-772)."""
    if not isinstance(labels, torch.Tensor) or labels.numel() == 0:
        return labels
    ends = _document_ends(seq_lengths, labels.device)
    if ends is None:
        return labels
    first_of_next = ends.masked_fill(ends >= labels.numel(), 0)
    return labels.reshape(-1).index_fill(0, first_of_next, ignore_index).view_as(labels)


def packed_position_ids(seq_lengths, device=None) -> torch.Tensor:
    """positions == concat(arange(len)) (tests/utils/test_packing.py:1095-1117), int32 like TRL's."""
    lengths = torch.as_tensor(seq_lengths, device=device or "cpu").to(torch.int64).reshape(-1)
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
