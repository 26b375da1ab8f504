"""Host logic of the decode path that needs no GPU: the tag source of the in-launch hand-offs (kernels/decode.HandOff) and the
engine's split-size rule. (The kernels themselves: tests/test_gpu_decode.py.)"""
import math

import torch


def test_handoff_tags_are_never_zero_and_never_repeat():
    from unsloth_amd import _lib
    from unsloth_amd.kernels.decode import HandOff
    ho = HandOff(torch.device("cpu"))
    assert ho.ws.numel() * 4 >= _lib.GEMV_SYNC_BYTES and int(ho.ws.abs().sum()) == 0
    seen = set()
    for _ in range(1000):                          # eager launches: a host counter, no device half
        tag, dev = ho.tags()
        assert dev is None and tag > 0 and tag not in seen
        seen.add(tag)
    # launches replayed from a hipGraph: a per-site constant below the stride + the owner's device counter
    step = torch.ones(1, dtype=torch.int32)
    hg = HandOff(torch.device("cpu"), step_dev=step)
    for site in (1, 5, _lib.TAG_STRIDE - 1):
        tag, dev = hg.tags(site)
        assert tag == site and dev is step
    for bad in (0, _lib.TAG_STRIDE):
        try:
            hg.tags(bad)
        except AssertionError:
            continue
        raise AssertionError("a site outside 1 .. TAG_STRIDE - 1 must be refused")
    # effective tags of two steps never meet for any pair of sites: step * stride + site with 0 < site < stride
    s = _lib.TAG_STRIDE
    assert {1 * s + a for a in range(1, s)}.isdisjoint({2 * s + a for a in range(1, s)})
    # wrap-around of the host counter clears the workspace and starts over
    ho._host = (1 << 31) - 1
    ho.ws.fill_(7)
    assert ho.next_tag() == 1 and int(ho.ws.abs().sum()) == 0


def test_fused_attention_split_rule_keeps_the_launch_within_256_workgroups():
    """models/decode.DecodeEngine: 128 keys per split, longer ones where that would mean more than 256 workgroups (the granule
    combine needs the whole launch resident)."""
    from unsloth_amd.models.decode import SPLIT_KEYS
    for S, Hk, B, want in ((2048, 8, 1, 128), (4096, 8, 1, 128), (8192, 8, 1, 256), (32768, 8, 1, 1024), (2048, 2, 1, 128),
                           (131072, 8, 1, 4096), (4096, 4, 1, 128)):
        f = max(SPLIT_KEYS, int(math.ceil(S * Hk * B / 256 / 16) * 16))
        assert f == want and f % 16 == 0
        assert ((S + f - 1) // f) * Hk * B <= 256
