"""CPU, bit-exact: the token-index functions against the reference's own vectors
(/root/reference/tests/utils/test_packing.py:135-157, 1375-1439, 1488-1524, 1575-1582; restated here
because the reference tree does not travel)."""
import torch

from unsloth_amd.utils.packing import (
    build_sdpa_packed_attention_mask,
    enable_padding_free_metadata,
    get_packed_info_from_kwargs,
    mask_packed_boundary_labels,
    mask_packed_sequence_boundaries,
    packed_position_ids,
)
from oracle.ref_ops import shift_labels


def test_mask_packed_sequence_boundaries_marks_single_row():          # ref :135-147
    s = torch.arange(6, dtype=torch.long).view(1, 6)
    assert mask_packed_sequence_boundaries(s, torch.tensor([2, 1, 3], dtype=torch.int32)) is True
    assert s.view(-1).tolist() == [0, -100, -100, 3, 4, -100]


def test_mask_packed_sequence_boundaries_across_multiple_rows():      # ref :149-157
    s = torch.arange(10, dtype=torch.long).view(2, 5)
    assert mask_packed_sequence_boundaries(s, torch.tensor([3, 2, 4, 1], dtype=torch.int32)) is True
    assert [i for i, v in enumerate(s.view(-1).tolist()) if v == -100] == [2, 4, 8, 9]


def test_mask_packed_boundary_labels_vectors():                       # ref :1375-1439
    labels = torch.arange(6, dtype=torch.long).view(1, 6)
    out = mask_packed_boundary_labels(labels, torch.tensor([2, 1, 3], dtype=torch.int32))
    assert out.reshape(-1).tolist() == [-100, 1, -100, -100, 4, 5]
    assert labels.reshape(-1).tolist() == [0, 1, 2, 3, 4, 5]          # out of place
    assert out.shape == labels.shape and out.dtype == labels.dtype
    assert mask_packed_boundary_labels(labels, None) is labels
    assert mask_packed_boundary_labels(labels, torch.tensor([], dtype=torch.int32)) is labels
    assert mask_packed_boundary_labels(None, torch.tensor([2, 4])) is None
    padded = torch.tensor([[10, 11, 12, 13, -100, -100]], dtype=torch.long)
    assert mask_packed_boundary_labels(padded, torch.tensor([2, 2], dtype=torch.int32)).reshape(-1).tolist() == \
        [10, 11, -100, 13, -100, -100]
    whole = torch.arange(4, dtype=torch.long).view(1, 4)
    assert mask_packed_boundary_labels(whole, [2, 2]).reshape(-1).tolist() == [-100, 1, -100, 3]


def test_raw_guard_equals_shifted_guard():                            # ref :1389-1407
    labels = torch.arange(100, 112, dtype=torch.long).view(1, 12)
    lengths = torch.tensor([5, 4, 3], dtype=torch.int32)
    a = shift_labels(labels)
    mask_packed_sequence_boundaries(a, lengths)
    b = shift_labels(mask_packed_boundary_labels(labels, lengths))
    assert torch.equal(a, b)


def test_guard_idempotent_on_trl_labels():                            # ref :1410-1422, :1575-1582
    lengths = torch.tensor([2, 1, 3], dtype=torch.int32)
    labels = torch.arange(6, dtype=torch.long).view(1, 6)
    pos = torch.tensor([[0, 1, 0, 0, 1, 2]])
    trl = labels.clone()
    trl[pos == 0] = -100
    once = mask_packed_boundary_labels(trl, lengths)
    assert torch.equal(once, trl) and torch.equal(mask_packed_boundary_labels(once, lengths), once)


def test_fused_ce_label_vector():                                     # ref :1488-1524
    labels = torch.arange(8, dtype=torch.long).view(1, 8)
    got = mask_packed_boundary_labels(labels, torch.tensor([3, 5], dtype=torch.int32))
    assert got.reshape(-1).tolist() == [-100, 1, 2, -100, 4, 5, 6, 7]
    assert labels.reshape(-1).tolist() == list(range(8))


def test_packed_info_and_positions():                                 # ref :1095-1117, packing.py:586-606
    lengths = torch.tensor([3, 5, 2], dtype=torch.int64)
    info = get_packed_info_from_kwargs({"packed_seq_lengths": lengths}, torch.device("cpu"))
    l32, cu, mx = info
    assert l32.dtype == torch.int32 and cu.dtype == torch.int32
    assert cu.tolist() == [0, 3, 8, 10] and mx == 5
    assert get_packed_info_from_kwargs({}, torch.device("cpu")) is None
    pos = packed_position_ids(lengths)
    assert pos.dtype == torch.int32 and pos.tolist() == [0, 1, 2, 0, 1, 2, 3, 4, 0, 1]


def test_padding_free_collation_and_num_items():                      # ref :1545-1571
    b = enable_padding_free_metadata([[10, 11], [12], [13, 14, 15]])
    assert b["input_ids"].tolist() == [[10, 11, 12, 13, 14, 15]]
    assert b["position_ids"].tolist() == [[0, 1, 0, 0, 1, 2]] and b["position_ids"].dtype == torch.int32
    assert b["packed_seq_lengths"].tolist() == [2, 1, 3] and b["packed_seq_lengths"].dtype == torch.int32
    assert int(b["packed_seq_lengths"].sum()) == b["input_ids"].numel()
    # docs [10,11] [12] [13,14,15] -> 1 + 0 + 2 real CE targets after shift + boundary mask
    tgt = shift_labels(mask_packed_boundary_labels(b["labels"], b["packed_seq_lengths"]))
    assert int((tgt != -100).sum()) == 3


def test_sdpa_packed_mask_is_block_causal():                          # packing.py:650-693
    lengths = torch.tensor([2, 3], dtype=torch.int32)
    info = (lengths, None, 3)
    m = build_sdpa_packed_attention_mask(info, dtype=torch.float32, device=torch.device("cpu"))
    assert m.shape == (1, 1, 5, 5)
    allowed = (m[0, 0] == 0).int().tolist()
    assert allowed == [[1, 0, 0, 0, 0], [1, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 1, 1, 0], [0, 0, 1, 1, 1]]
    w = build_sdpa_packed_attention_mask((torch.tensor([4]), None, 4), dtype=torch.float32,
                                         device=torch.device("cpu"), sliding_window=2)
    assert (w[0, 0] == 0).int().tolist() == [[1, 0, 0, 0], [1, 1, 0, 0], [0, 1, 1, 0], [0, 0, 1, 1]]


def test_attention_band_matches_dense_packed_mask():
    """(lo, hi) band == the finite entries of build_sdpa_packed_attention_mask (reference utils/packing.py:650-693),
    both ways round: lo[q] <= key <= q  <=>  key <= q <= hi[key]. Integer, exact."""
    import torch
    from unsloth_amd.kernels.attention import attention_band
    from unsloth_amd.utils.packing import build_sdpa_packed_attention_mask
    gen = torch.Generator().manual_seed(0)
    for trial in range(20):
        n_docs = int(torch.randint(1, 7, (1,), generator=gen))
        lens = torch.randint(1, 40, (n_docs,), generator=gen).to(torch.int32)
        T = int(lens.sum())
        window = [None, 1, 5, 17, 1000][trial % 5]
        dense = build_sdpa_packed_attention_mask((lens, None, int(lens.max())), dtype=torch.float32, device="cpu",
                                                 sliding_window=window)[0, 0]
        allowed = torch.isfinite(dense) & (dense == 0)
        lo, hi = attention_band(T, seq_lengths=lens, sliding_window=window)
        assert lo.dtype == torch.int32 and lo.shape == (1, T)
        pos = torch.arange(T)
        by_lo = (pos[None, :] >= lo[0][:, None]) & (pos[None, :] <= pos[:, None])
        by_hi = (pos[:, None] <= hi[0][None, :]) & (pos[None, :] <= pos[:, None])
        assert torch.equal(by_lo, allowed) and torch.equal(by_hi, allowed)
        assert bool((lo[0][1:] >= lo[0][:-1]).all()) and bool((hi[0][1:] >= hi[0][:-1]).all())
    # no packing, window only, batch of 3
    lo, hi = attention_band(10, batch=3, sliding_window=4)
    assert lo.shape == (3, 10) and lo[1].tolist() == [0, 0, 0, 0, 1, 2, 3, 4, 5, 6] and hi[2].tolist() == [3, 4, 5, 6, 7, 8, 9, 9, 9, 9]
    # tokens past the packed documents form one more document
    lo, hi = attention_band(8, seq_lengths=[3, 2])
    assert lo[0].tolist() == [0, 0, 0, 3, 3, 5, 5, 5] and hi[0].tolist() == [2, 2, 2, 4, 4, 7, 7, 7]
    # batch of B packed rows: the lengths run over the FLATTENED batch (the reference's cu_seqlens); every row's band
    # equals the band of that row packed alone
    for trial in range(10):
        B, T = int(torch.randint(2, 5, (1,), generator=gen)), 24
        rows, flat = [], []
        for _ in range(B):
            cuts = sorted(set(torch.randint(1, T, (int(torch.randint(0, 4, (1,), generator=gen)),), generator=gen).tolist()))
            lens = [b - a for a, b in zip([0] + cuts, cuts + [T])]
            rows.append(lens)
            flat += lens
        window = [None, 3, 9][trial % 3]
        lo, hi = attention_band(T, batch=B, seq_lengths=flat, sliding_window=window)
        for b, lens in enumerate(rows):
            lo1, hi1 = attention_band(T, seq_lengths=lens, sliding_window=window)
            assert torch.equal(lo[b], lo1[0]) and torch.equal(hi[b], hi1[0])
    # a document that straddles two rows is cut at the row boundary
    lo, hi = attention_band(4, batch=2, seq_lengths=[6, 2])
    assert lo.tolist() == [[0, 0, 0, 0], [0, 0, 2, 2]] and hi.tolist() == [[3, 3, 3, 3], [1, 1, 3, 3]]


def test_adjacent_columns_detection():
    """kernels/utils._adjacent_columns: the q/k/v gradients are merged into one K-concatenated GEMM only when they
    really are consecutive column blocks of one row-major buffer."""
    import torch
    from unsloth_amd.kernels.utils import _adjacent_columns
    buf = torch.arange(6 * 20, dtype=torch.float32).view(6, 20)
    a, b, c = buf[:, :8], buf[:, 8:12], buf[:, 12:20]
    cat = _adjacent_columns([a, b, c])
    assert cat is not None and cat.shape == (6, 20) and torch.equal(cat, buf) and cat.data_ptr() == buf.data_ptr()
    assert _adjacent_columns([a, c]) is None                       # gap
    assert _adjacent_columns([b, a]) is None                       # wrong order
    assert _adjacent_columns([a, b.clone()]) is None               # different storage
    assert _adjacent_columns([a, buf[:5, 8:12]]) is None           # different row count
    assert _adjacent_columns([a.double(), b.double()]) is None     # copies, not views
    part = _adjacent_columns([b, c])                                # a sub-range is fine: starts at column 8
    assert part is not None and part.shape == (6, 12) and torch.equal(part, buf[:, 8:])
    three_d = buf.view(2, 3, 20)[..., :8].reshape(-1, 8)            # the [B, T, H] -> [B*T, H] view autograd hands over
    assert _adjacent_columns([three_d, buf.view(2, 3, 20)[..., 8:12].reshape(-1, 4)]) is not None


def test_padding_mask_as_documents_matches_the_dense_key_mask_on_real_rows():
    """kernels/attention.padding_mask_documents: right / left / two-sided padding become packed documents whose band
    equals causal AND key-mask on every REAL query row; padding rows keep a non-empty key set; holes -> None."""
    import torch
    from unsloth_amd.kernels.attention import attention_band, padding_mask_documents
    T = 12
    mask = torch.tensor([[1] * 12,
                         [1] * 7 + [0] * 5,              # right padding
                         [0] * 4 + [1] * 8,              # left padding
                         [0] * 2 + [1] * 6 + [0] * 4,    # both
                         [0] * 12], dtype=torch.int64)   # an all-padding row
    docs = padding_mask_documents(mask)
    assert docs.dtype == torch.int32 and docs.tolist() == [0, 12, 0, 0, 7, 5, 4, 8, 0, 2, 6, 4, 0, 0, 12]
    lo, hi = attention_band(T, batch=mask.shape[0], seq_lengths=docs)
    pos = torch.arange(T)
    for b in range(mask.shape[0]):
        band_allowed = (pos[None, :] >= lo[b][:, None]) & (pos[None, :] <= pos[:, None])          # [q, key]
        dense = (pos[None, :] <= pos[:, None]) & (mask[b] != 0)[None, :]
        real = mask[b] != 0
        assert torch.equal(band_allowed[real], dense[real]), b
        assert bool(band_allowed.any(1).all())                     # no empty softmax row, padding rows included
        # hi: the last query that sees each key = end of its run
        assert bool((hi[b] >= pos).all())
    holes = torch.tensor([[1, 1, 0, 1, 1, 0]])
    assert padding_mask_documents(holes) is None
