"""CPU: the checkpoint loader against a directory this repo did NOT write (tests/golden/ckpt_tiny, made by
tests/golden/make_golden_checkpoint.py in the build container): HF `LlamaForCausalLM.save_pretrained` shards + index + config,
the reference's `ProjectorModel.save_pretrained`, the CLIP tower in the layout of the reference's pin (`vision_model.*` keys + the
`pos_emb` alias, torch pickle) and in transformers 5's own (flat keys, safetensors).  `load_auroracap` must turn each into the
oracle's weights such that the ORACLE reproduces what the HF / reference-side stack computed from the same directory
(g13_checkpoint_e2e.npz): ViT + ToMe features, spliced embeddings, greedy ids and logits - inference.py:42-57, aurora.py:214-258."""
import json
import os

import numpy as np
import pytest
import torch

from aurora_amd import checkpoint as CK
from oracle import aurora_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "golden", "ckpt_tiny")
G13 = os.path.join(HERE, "golden", "g13_checkpoint_e2e.npz")


@pytest.mark.parametrize("visual", ["visual_encoder", "visual_encoder_hf5"])
def test_loader_and_oracle_reproduce_the_reference_side_stack(visual):
    g = np.load(G13)
    cfg, w = CK.load_auroracap(ROOT, visual_encoder=visual)
    assert cfg["llm"]["rope_factor"] == 4.0 and cfg["llm"]["rope_theta"] == 10000.0 and cfg["llm"]["eos_token_id"] == 2
    assert cfg["llm"]["hidden_size"] == 128 and cfg["llm"]["num_hidden_layers"] == 2 and cfg["llm"]["vocab_size"] == 320
    assert cfg["vit"]["hidden_size"] == 64 and cfg["vit"]["patch_size"] == 14 and cfg["vit"]["image_size"] == 56
    assert cfg["vit"]["hidden_act"] == "quick_gelu"
    f32 = lambda t: {k: ([f32(x) for x in v] if isinstance(v, list) else v.float()) for k, v in t.items()}
    w = {k: f32(v) for k, v in w.items()}                                  # fp16 on disk, fp32 arithmetic (as the generator ran it)
    px = torch.from_numpy(g["pixel_values"]).float()
    ids = g["input_ids"][0].tolist()
    ratio = float(g["ratio"])
    feats = O.vit_features(px, w["vit"], cfg["vit"], ratio, None)
    assert feats.shape[1] == int(g["n_kept"])
    ref = torch.from_numpy(g["vis_feats"])
    assert (feats - ref).norm() / ref.norm() < 1e-4                       # same merges, fp32 both sides
    out, logits = O.caption_ids(px, ids, w, cfg, ratio, len(g["ids"]), eos_id=None, q=None, return_logits=True)
    assert out == g["ids"].tolist()
    rl = torch.from_numpy(g["logits"])
    assert (logits - rl).abs().max() <= 2e-3 * rl.abs().max()


def test_rope_config_spellings():
    """transformers <= 4.x (the reference's pin, vicuna-7b-v1.5-16k) and 5.x spell linear rope scaling differently."""
    base = dict(hidden_size=128, num_attention_heads=4, num_hidden_layers=2, intermediate_size=256, vocab_size=320)

    def parse(extra, tmp):
        with open(os.path.join(tmp, "config.json"), "w") as f:
            json.dump(dict(base, **extra), f)
        return CK.llm_config(tmp)

    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        c = parse({"rope_scaling": {"factor": 4.0, "type": "linear"}, "rms_norm_eps": 1e-5}, tmp)          # 4.x style
        assert c["rope_factor"] == 4.0 and c["rope_theta"] == 10000.0
        c = parse({"rope_scaling": {"factor": 8.0, "rope_type": "linear"}, "rope_theta": 5e5}, tmp)
        assert c["rope_factor"] == 8.0 and c["rope_theta"] == 5e5
        c = parse({"rope_parameters": {"factor": 4.0, "rope_theta": 1e4, "rope_type": "linear", "type": "linear"}}, tmp)     # 5.x style
        assert c["rope_factor"] == 4.0 and c["rope_theta"] == 1e4
        c = parse({"rope_parameters": {"rope_theta": 1e6, "rope_type": "default"}}, tmp)
        assert c["rope_factor"] == 1.0 and c["rope_theta"] == 1e6
        c = parse({"rope_scaling": None}, tmp)
        assert c["rope_factor"] == 1.0
        with pytest.raises(NotImplementedError):
            parse({"rope_scaling": {"type": "dynamic", "factor": 2.0}}, tmp)
        with pytest.raises(NotImplementedError):
            parse({"num_key_value_heads": 2}, tmp)


def test_shard_index_is_authoritative(tmp_path):
    """A stale consolidated file next to the shards must not be read; a shard the index names must exist."""
    import shutil
    d = tmp_path / "llm"
    shutil.copytree(ROOT, d, ignore=shutil.ignore_patterns("visual_encoder*", "projector"))
    good = CK._load_state(str(d))
    from safetensors.torch import save_file
    save_file({"model.norm.weight": torch.zeros(128, dtype=torch.float16)}, str(d / "zzz_stale.safetensors"))
    again = CK._load_state(str(d))
    assert set(again) == set(good) and torch.equal(again["model.norm.weight"], good["model.norm.weight"])
    os.remove(d / "model-00002-of-00003.safetensors")
    with pytest.raises(Exception):
        CK._load_state(str(d))


def test_cli_path_pieces_against_the_reference_side_run():
    """G14: the CLI path of inference.py:58-98 on the fixture directory - a real HF tokenizer (`AutoTokenizer.from_pretrained`), the
    reference's prompt template and `process_text`, `CLIPImageProcessor` pixels, the reference-side stack, EOS-aware greedy decode and
    `batch_decode`.  Here on the CPU: this repo's `build_prompt` / `process_text` give the reference's prompt and ids, the input-stage
    restatement gives the processor's pixels bit for bit, and the oracle gives the same caption ids and text."""
    from PIL import Image
    from transformers import AutoTokenizer
    from aurora_amd.model import build_prompt, process_text
    from oracle import preprocess_ref as PR
    g = np.load(os.path.join(HERE, "golden", "g14_cli_e2e.npz"))
    tok = AutoTokenizer.from_pretrained(ROOT, padding_side="right")
    prompt_text = build_prompt(str(g["prompt"]), 1)
    assert prompt_text == str(g["prompt_text"])
    ids = process_text(prompt_text, tok)
    assert ids.tolist() == g["input_ids"].tolist() and ids[0, 0] == tok.bos_token_id
    img = np.asarray(Image.open(os.path.join(ROOT, "clip.png")).convert("RGB"))
    px = PR.clip_preprocess(img[None], 56)
    assert np.array_equal(np.asarray(px).view(np.uint16), g["pixel_values"].view(np.uint16))          # fp16 bits of the processor's output
    cfg, w = CK.load_auroracap(ROOT)
    f32 = lambda t: {k: ([f32(x) for x in v] if isinstance(v, list) else v.float()) for k, v in t.items()}
    w = {k: f32(v) for k, v in w.items()}
    out = O.caption_ids(torch.from_numpy(g["pixel_values"]).float(), ids[0].tolist(), w, cfg, float(g["ratio"]), int(g["new"]),
                        eos_id=cfg["llm"]["eos_token_id"], q=None)
    assert out == g["ids"].tolist()
    assert tok.batch_decode([out], skip_special_tokens=True)[0] == str(g["text"])


# ---- G15: projector/config.json is read and honoured.  Directories written by the reference's own
# ProjectorModel.save_pretrained for every shape ProjectorModel.__init__ can build (modeling_projector.py:20-33,
# configuration_projector.py:9-22); outputs computed by the reference module (tests/golden/make_golden_projector.py).
PV = os.path.join(HERE, "golden", "proj_variants")
G15 = os.path.join(HERE, "golden", "g15_projector_variants.npz")
PV_OK = {"d3_silu": (3, "silu", True), "d1": (1, "gelu", True), "d2_nobias": (2, "gelu", False), "d2_tanh": (2, "gelu_pytorch_tanh", True),
         "d2_relu": (2, "relu", True), "d2_quick": (2, "quick_gelu", True)}


@pytest.mark.parametrize("name", sorted(PV_OK))
def test_projector_config_is_read_and_the_oracle_reproduces_the_reference_module(name):
    from aurora_amd.engine import projector_settings
    g = np.load(G15)
    pc = CK.projector_config(os.path.join(PV, name))
    assert (pc["depth"], pc["hidden_act"], pc["bias"]) == PV_OK[name] and pc["visual_hidden_size"] == 64 and pc["llm_hidden_size"] == 128
    assert projector_settings(pc, 64, 128) == dict(depth=pc["depth"], hidden_act=pc["hidden_act"], bias=pc["bias"])
    pw = CK.projector_weights(CK._load_state(os.path.join(PV, name)), pc)
    assert len([k for k in pw if k.endswith(".weight")]) == pc["depth"] and any(k.endswith(".bias") for k in pw) == pc["bias"]
    y = O.projector(torch.from_numpy(g["x"]), {k: v.float() for k, v in pw.items()}, None, pc["hidden_act"])
    ref = torch.from_numpy(g["y_" + name])
    assert (y - ref).abs().max() <= 1e-5 * ref.abs().max()


def test_projector_config_the_kernels_cannot_honour_raises_instead_of_loading_wrong():
    from aurora_amd.engine import projector_settings
    pc = CK.projector_config(os.path.join(PV, "d2_mish"))
    assert pc["hidden_act"] == "mish"
    with pytest.raises(NotImplementedError, match="mish"):
        projector_settings(pc, 64, 128)
    with pytest.raises(NotImplementedError, match="depth"):
        projector_settings(dict(depth=9), 64, 128)
    with pytest.raises(ValueError, match="visual_hidden_size"):
        projector_settings(CK.projector_config(os.path.join(PV, "d1")), 1280, 128)
    # weights that do not match the config (a depth-3 config over a depth-2 file, a bias-free config over biased weights)
    sd2 = CK._load_state(os.path.join(PV, "d2_relu"))
    with pytest.raises(KeyError, match="model.4"):
        CK.projector_weights(sd2, dict(depth=3, bias=True))
    with pytest.raises(KeyError, match="unexpected"):
        CK.projector_weights(sd2, dict(depth=2, bias=False))


def test_whole_directory_with_a_mismatched_projector_config_is_rejected(tmp_path):
    """load_auroracap cross-checks projector/config.json against the tower and the language model it sits between"""
    import shutil
    root = tmp_path / "ckpt"
    shutil.copytree(ROOT, root)
    cfg, w = CK.load_auroracap(str(root))
    assert cfg["projector"] == dict(visual_hidden_size=64, llm_hidden_size=128, depth=2, hidden_act="gelu", bias=True)
    pj = json.load(open(root / "projector" / "config.json"))
    pj["visual_hidden_size"] = 1280
    json.dump(pj, open(root / "projector" / "config.json", "w"))
    with pytest.raises(ValueError, match="visual_hidden_size"):
        CK.load_auroracap(str(root))
