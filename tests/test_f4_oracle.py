"""CPU: oracle restatements of the remaining AuroraModel inference features against outputs of
the reference's own code (tests/golden/make_golden_posemb.py): position-table interpolation for non-native input
sizes (aurora.py:909-951) and the slow-fast splice (model/utils.py:297-431)."""
import pytest
import torch

from oracle import aurora_oracle as O
from tests.util import golden


@pytest.fixture(scope="module")
def g11():
    return golden("g11_pos_interp.npz")


def test_interpolate_pos_encoding_matches_reference(g11):
    assert len(g11["cases"]) == 9
    for c in g11["cases"]:
        name, _, patch, h, w = str(c).split(":")
        pos = torch.from_numpy(g11[f"{name}.pos"])
        got = O.interpolate_pos_encoding(pos, int(h), int(w), int(patch))
        assert torch.equal(got, torch.from_numpy(g11[f"{name}.{h}x{w}"])), c
    pos = torch.from_numpy(g11["tiny.pos"])
    assert O.interpolate_pos_encoding(pos, 56, 56, 14) is pos                     # native grid: untouched (:919-924)


def test_slowfast_splice_matches_reference(g11):
    emb, ids = torch.from_numpy(g11["sf.embed"]), torch.from_numpy(g11["sf.ids"])[0]
    feats = [torch.from_numpy(g11[f"sf.feat{i}"]) for i in range(3)]
    got = O.splice_slowfast(ids, emb, feats)
    assert torch.equal(got, torch.from_numpy(g11["sf.inputs_embeds"])[0])
    assert got.shape[0] == 6 + 9 + 4 + 4
    with pytest.raises(IndexError):
        O.splice_slowfast(ids, emb, feats[:2])                                     # a marker without a frame is an error here
