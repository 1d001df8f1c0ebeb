#!/usr/bin/env python3
"""Generate tests/golden/g11_pos_interp.npz by RUNNING the reference's AuroraEncoder.interpolate_pos_encoding
(src/xtuner/xtuner/model/aurora.py:909-951) on seeded position embeddings, and the slow-fast splice
(prepare_inputs_labels_for_multimodal_slowfast, model/utils.py:297-431) on seeded per-frame features.

The method is called unbound on a small stand-in `self` that carries exactly the attributes it touches (pos_emb,
vision_model.config.patch_size, vision_model.embeddings.position_embedding / position_ids); building the whole
AuroraEncoder is not possible under this container's transformers (5.x dropped classes it subclasses, see
make_golden.py).  Fixtures are DATA ONLY.  Re-run: python tests/golden/make_golden_posemb.py  (CPU, seconds).
"""
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference, save  # noqa: E402


def main():
    ref = import_reference()
    aurora, utils = ref["aurora"], ref["utils"]
    out = {}
    cases = []
    torch.manual_seed(0)
    for name, (n_side, dim, patch, shapes) in {
        "tiny": (4, 64, 14, [(56, 56), (56, 84), (42, 56), (70, 70), (28, 98)]),
        "vith": (27, 32, 14, [(378, 378), (224, 224), (378, 252), (336, 448)]),      # 27x27 grid like ViT-H/14-378, thin dim
    }.items():
        pos = torch.randn(1 + n_side * n_side, dim)
        out[f"{name}.pos"] = pos.numpy()
        for (h, w) in shapes:
            emb = SimpleNamespace(position_embedding=SimpleNamespace(weight=None), position_ids=None)
            fake = SimpleNamespace(pos_emb=pos, vision_model=SimpleNamespace(config=SimpleNamespace(patch_size=patch), embeddings=emb))
            aurora.AuroraEncoder.interpolate_pos_encoding(fake, torch.zeros(1, 3, h, w))
            got = emb.position_embedding.weight.detach().float().numpy()
            out[f"{name}.{h}x{w}"] = got
            cases.append(f"{name}:{n_side}:{patch}:{h}:{w}")
            print(name, (h, w), got.shape)
    out["cases"] = np.array(cases)

    # slow-fast splice: frame 0 carries 9 tokens, frames 1..2 carry 4 tokens each
    class Emb(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.e = torch.nn.Embedding(50, 8)

        def get_input_embeddings(self):
            return self.e
    torch.manual_seed(1)
    llm = Emb()
    feats = [torch.randn(1, 9, 8), torch.randn(4, 8), torch.randn(4, 8)]
    ids = torch.tensor([[1, 10, -200, 11, -200, 12, -200, 13, 14]])
    with torch.no_grad():
        res = utils.prepare_inputs_labels_for_multimodal_slowfast(llm=llm, input_ids=ids, pixel_values=feats)
    out["sf.embed"] = llm.e.weight.detach().numpy()
    out["sf.ids"] = ids.numpy()
    for i, f in enumerate(feats):
        out[f"sf.feat{i}"] = f.reshape(-1, 8).numpy()
    out["sf.inputs_embeds"] = res["inputs_embeds"].detach().numpy()
    print("slowfast splice", res["inputs_embeds"].shape)
    save("g11_pos_interp.npz", ref["versions"], **out)


if __name__ == "__main__":
    main()
