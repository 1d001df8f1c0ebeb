#!/usr/bin/env python3
"""G15: projector checkpoints of every shape the reference's `ProjectorModel` can build, written by the reference's own
`ProjectorModel.save_pretrained`, with the outputs the reference module computes from them (`projector/config.json`
must be read - depth, hidden_act, bias - and honoured or rejected, never ignored).

    python tests/golden/make_golden_projector.py         (build container only: needs /root/reference; CPU, seconds)

Writes tests/golden/proj_variants/<name>/{config.json, model.safetensors} and tests/golden/g15_projector_variants.npz
(x [24, 64] and y_<name> [24, 128] per variant, fp32 arithmetic on the fp16-stored weights).
Variants: d3_silu (depth 3, 'silu'), d1 (a single Linear), d2_nobias (bias False), d2_tanh ('gelu_pytorch_tanh'), d2_relu,
d2_quick ('quick_gelu') - all must load and agree - and d2_mish ('mish': an ACT2FN entry the kernels do not implement - must raise).
The reference's two source files that `save_pretrained` copies next to the weights (auto_map) are deleted: source is not fixture data.
"""
import os
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G          # noqa: E402

OUT = os.path.join(HERE, "proj_variants")
VARIANTS = {"d3_silu": dict(depth=3, hidden_act="silu"), "d1": dict(depth=1), "d2_nobias": dict(depth=2, bias=False),
            "d2_tanh": dict(depth=2, hidden_act="gelu_pytorch_tanh"), "d2_relu": dict(depth=2, hidden_act="relu"),
            "d2_quick": dict(depth=2, hidden_act="quick_gelu"), "d2_mish": dict(depth=2, hidden_act="mish")}


def main():
    R = G.import_reference()
    sys.modules.pop("peft", None)
    pmod, pcfg = R["pmod"], R["pcfg"]
    import transformers
    from safetensors.torch import load_file
    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    gen = torch.Generator().manual_seed(1515)
    x = torch.randn(24, 64, generator=gen).half().float()
    arrays = {"x": x.numpy(), "versions": np.array(f"torch {torch.__version__}, transformers {transformers.__version__}")}
    for name, kw in VARIANTS.items():
        d = os.path.join(OUT, name)
        m = pmod.ProjectorModel(pcfg.ProjectorConfig(visual_hidden_size=64, llm_hidden_size=128, **kw))
        with torch.no_grad():
            for n, p in m.named_parameters():
                p.copy_(torch.randn(p.shape, generator=gen) * (0.12 if p.dim() == 2 else 0.05))
        m.half().save_pretrained(d)
        for f in os.listdir(d):
            if f.endswith(".py"):
                os.remove(os.path.join(d, f))
        m2 = pmod.ProjectorModel(pcfg.ProjectorConfig.from_pretrained(d)).float().eval()
        m2.load_state_dict({k: v.float() for k, v in load_file(os.path.join(d, "model.safetensors")).items()}, strict=True)
        with torch.no_grad():
            arrays["y_" + name] = m2(x).numpy()
        print(name, sorted(load_file(os.path.join(d, "model.safetensors"))), float(np.abs(arrays["y_" + name]).max()))
    np.savez_compressed(os.path.join(HERE, "g15_projector_variants.npz"), **arrays)


if __name__ == "__main__":
    main()
