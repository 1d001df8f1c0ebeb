#!/usr/bin/env python3
"""G13: a tiny AuroraCap checkpoint directory NOT written by this repo's code, plus the greedy ids the reference-side stack
produces from it (pin aurora_amd/checkpoint.py to on-disk formats it did not define itself).

    python tests/golden/make_golden_checkpoint.py        (build container only: needs /root/reference; CPU, ~20 s)

Writers (the same three calls `to_xtuner_llava` makes, aurora.py:333, 353, 360):
  <root>/                 transformers `LlamaForCausalLM.save_pretrained(max_shard_size=...)`: sharded safetensors +
                          model.safetensors.index.json + config.json with linear rope scaling x4 (vicuna-7b-v1.5-16k's scheme)
  <root>/projector/       the REFERENCE's `ProjectorModel.save_pretrained` (modules/projector/modeling_projector.py) - config.json
                          and key names are the reference's
  <root>/visual_encoder/  the HF CLIP vision tower nested under the attribute names `AuroraEncoder` uses (aurora.py:870-878:
                          `self.vision_model = ...`, `self.pos_emb = self.vision_model.embeddings.position_embedding.weight`),
                          serialised as `pytorch_model.bin` - what pth_to_hf.py writes by default (no --safe-serialization; a
                          safetensors save of two names for one tensor is refused by transformers) - and the config written by
                          `CLIPVisionConfig.save_pretrained`.  transformers 5.x (this image) flattens CLIPVisionModel's own keys,
                          the reference's pin (<= 4.42.4) keeps the `vision_model.` prefix: the nesting above reproduces the
                          pinned layout.  <root>/visual_encoder_hf5/ is the same tower as transformers 5.15 writes it natively
                          (flat keys, safetensors, no `pos_emb`): the loader must read both.
Readers for the expected ids (nothing of aurora_amd is imported here): `AutoModelForCausalLM.from_pretrained(root)`,
`CLIPVisionModel.from_pretrained(visual_encoder_hf5)` for embeddings + pre_layrnorm, the reference's `AuroraCLIPEncoder` with the
tower's encoder weights, `hidden_states[-2][:, 1:]` (aurora.py:253), the reference's `ProjectorModel.from_pretrained`, the reference's
`prepare_inputs_labels_for_multimodal` (model/utils.py:138-295) and HF greedy `generate` (inference.py:89-96), fp32 arithmetic on the
fp16-stored weights.  The full `AuroraEncoder` / `AuroraModel` wrappers cannot run on transformers 5.x; the
composition above is the call sequence of aurora.py:214-258 made of the pieces that do.

G14 (the CLI path, inference.py:58-98, on the same directory): a real HF tokenizer - a small Llama-style BPE vocabulary (metaspace, `<unk>` /
`<s>` / `</s>` = 0 / 1 / 2) trained here on a few sentences with the `tokenizers` library and written by `PreTrainedTokenizerFast.save_pretrained`
(`tokenizer.json`, `tokenizer_config.json`; no Vicuna vocabulary exists offline) - and `clip.png`, a seeded 90 x 70 RGB image.  Expected output:
the reference's own prompt template (`PROMPT_TEMPLATE.vicuna`, xtuner/utils/templates.py) and its own `process_text` (the function's source is
executed from /root/reference/inference.py at generation time, with `.cuda()` made a no-op), HF `CLIPImageProcessor(size=56, crop_size=56)` (the
reference passes its model's 378), then the stack above, greedy `generate` WITH the EOS rule, `tokenizer.batch_decode(..., skip_special_tokens=True)`.
"""
import json
import os
import shutil
import sys

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G          # noqa: E402

OUT = os.path.join(HERE, "ckpt_tiny")
RATIO, FRAMES, NEW = 0.5, 3, 12


def main():
    R = G.import_reference()
    sys.modules.pop("peft", None)               # the import stub has served its purpose; transformers probes for the real package
    # the fixture must let the GPU test compare many positions under the margin rule: take the first seed whose first 8 greedy
    # steps have a top-1 / top-2 margin above 2.5 % of the logit scale (the test's tolerance is 1 %: margins above 2 % are compared)
    for seed in range(1313, 1313 + 400):
        clear = build(R, seed)
        print("seed", seed, "positions with a clear margin:", clear, flush=True)
        if clear >= 8:
            break
    else:
        raise SystemExit("no seed with clear margins found")


def build(R, seed):
    aurora, utils, pmod, pcfg = R["aurora"], R["utils"], R["pmod"], R["pcfg"]
    import transformers
    from transformers import AutoModelForCausalLM, CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM

    if os.path.isdir(OUT):
        shutil.rmtree(OUT)
    os.makedirs(OUT)
    gen = torch.Generator().manual_seed(seed)

    def reinit(model, std):
        with torch.no_grad():
            for n, p in model.named_parameters():
                if p.dim() >= 2:
                    p.copy_(torch.randn(p.shape, generator=gen) * std)
                elif "norm" in n and n.endswith("weight"):
                    p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=gen))
                else:
                    p.copy_(0.05 * torch.randn(p.shape, generator=gen))

    # ---- language model: dims the kernels accept (hidden % 128, head_dim % 32)
    lkw = dict(vocab_size=320, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4,
               max_position_embeddings=16384, rms_norm_eps=1e-5, rope_theta=10000.0, bos_token_id=1, eos_token_id=2, pad_token_id=0,
               tie_word_embeddings=False)
    try:
        lc = LlamaConfig(**lkw, rope_scaling={"type": "linear", "factor": 4.0})
    except Exception:                                               # noqa: BLE001 - key spelling differs between transformers versions
        lc = LlamaConfig(**lkw, rope_scaling={"rope_type": "linear", "factor": 4.0})
    llm = LlamaForCausalLM(lc)
    reinit(llm, 0.15)
    llm.half().save_pretrained(OUT, max_shard_size="300KB")
    # ---- vision tower
    vc = CLIPVisionConfig(hidden_size=64, num_attention_heads=4, num_hidden_layers=4, intermediate_size=128, patch_size=14, image_size=56,
                          hidden_act="quick_gelu", layer_norm_eps=1e-5)
    vis = CLIPVisionModel(vc)
    reinit(vis, 0.06)
    vis.half()
    vis.save_pretrained(os.path.join(OUT, "visual_encoder_hf5"))

    class PinnedLayout(nn.Module):                                   # attribute names of AuroraEncoder, aurora.py:870-878
        def __init__(self, tower):
            super().__init__()
            self.vision_model = tower
            self.pos_emb = tower.embeddings.position_embedding.weight

    inner = vis.vision_model if hasattr(vis, "vision_model") else vis
    vdir = os.path.join(OUT, "visual_encoder")
    os.makedirs(vdir)
    torch.save(PinnedLayout(inner).state_dict(), os.path.join(vdir, "pytorch_model.bin"))
    vc.save_pretrained(vdir)
    # ---- projector: the reference's own class and writer
    proj = pmod.ProjectorModel(pcfg.ProjectorConfig(visual_hidden_size=64, llm_hidden_size=128, depth=2))
    reinit(proj, 0.1)
    proj.half().save_pretrained(os.path.join(OUT, "projector"))
    for f in os.listdir(os.path.join(OUT, "projector")):            # the writer also copies the reference's two source files next
        if f.endswith(".py"):                                       # to the weights (auto_map): source is not fixture data - drop it
            os.remove(os.path.join(OUT, "projector", f))

    # ---- expected ids from the directory, through HF / reference readers only
    llm2 = AutoModelForCausalLM.from_pretrained(OUT, dtype=torch.float32).eval()
    vis2 = CLIPVisionModel.from_pretrained(os.path.join(OUT, "visual_encoder_hf5"), dtype=torch.float32).eval()
    tower = vis2.vision_model if hasattr(vis2, "vision_model") else vis2
    pinned = torch.load(os.path.join(vdir, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    assert torch.equal(pinned["pos_emb"], pinned["vision_model.embeddings.position_embedding.weight"])
    for k, v in tower.state_dict().items():                         # both layouts hold the same tower
        assert torch.equal(pinned["vision_model." + k].float(), v), k
    # (the reference's ProjectorModel predates transformers 5's from_pretrained bookkeeping: build it from the config the
    #  reference's writer left and load the tensors the writer left, strictly)
    from safetensors.torch import load_file
    pdir = os.path.join(OUT, "projector")
    proj2 = pmod.ProjectorModel(pcfg.ProjectorConfig.from_pretrained(pdir)).float().eval()
    proj2.load_state_dict({k: v.float() for k, v in load_file(os.path.join(pdir, "model.safetensors")).items()}, strict=True)
    H = W = vc.image_size
    r = int(W * H / (vc.patch_size ** 2) * (1 - RATIO) / vc.num_hidden_layers)                      # aurora.py:895
    enc = aurora.AuroraCLIPEncoder(vc, r=r).eval()
    enc.load_state_dict(tower.encoder.state_dict())
    px = (torch.randn(FRAMES, 3, H, W, generator=gen) * 0.8).half().float()
    ids = torch.tensor([[1, 17] + [-200, 30] * FRAMES + [40, 41, 42]])
    with torch.no_grad():
        h = tower.pre_layrnorm(tower.embeddings(px))                                              # HF CLIPVisionTransformer.forward
        out = enc(h, output_hidden_states=True, return_dict=True)
        feats = out.hidden_states[-2][:, 1:]                                                        # aurora.py:253
        n_kept = feats.shape[1]
        vo = proj2(feats.reshape(1, FRAMES * n_kept, -1)).reshape(1, FRAMES, n_kept, -1)           # aurora.py:254-256
        data = utils.prepare_inputs_labels_for_multimodal(llm=llm2, input_ids=ids, pixel_values=vo)  # aurora.py:258
        emb = data["inputs_embeds"]
        gen_ids = llm2.generate(inputs_embeds=emb, attention_mask=None, do_sample=False, temperature=0.0, top_p=1.0, num_beams=1,
                                max_new_tokens=NEW, min_new_tokens=NEW)
        tok_emb = llm2.get_input_embeddings()(gen_ids[0, :-1])[None]
        logits = llm2(inputs_embeds=torch.cat([emb, tok_emb], dim=1)).logits[0, emb.shape[1] - 1:]
    if not (logits.argmax(-1) == gen_ids[0]).all():          # a near-tie decided differently by the cached and the full forward
        return 0
    top2 = logits.topk(2, dim=-1).values
    rel = (top2[:, 0] - top2[:, 1]) / logits.abs().max()
    clear = 0
    while clear < NEW and rel[clear] > 0.025:
        clear += 1
    if clear < 8:
        return clear
    print("n_kept", n_kept, "prefix", emb.shape[1], "ids", gen_ids[0].tolist(), "min top-2 margin / max |logit|",
          float((top2[:, 0] - top2[:, 1]).min() / logits.abs().max()))
    G.save("g13_checkpoint_e2e.npz", dict(R["versions"], transformers_pin_of_reference="<=4.42.4"), pixel_values=px.numpy().astype(np.float16),
           input_ids=ids.numpy(), ratio=np.array(RATIO), r=np.array(r), n_kept=np.array(n_kept), ids=gen_ids[0].numpy(), logits=logits.numpy(),
           vis_feats=feats.numpy(), embeds=emb[0].numpy())
    g14(R, llm2, tower, enc, proj2, vc, gen)
    files = sorted(os.path.relpath(os.path.join(d, f), OUT) for d, _, fs in os.walk(OUT) for f in fs)
    print("\n".join(f"{os.path.getsize(os.path.join(OUT, f)):9d}  {f}" for f in files))
    print(json.dumps(json.load(open(os.path.join(OUT, "config.json"))), indent=None)[:600])
    return clear


CORPUS = ["a chat between a curious user and an artificial intelligence assistant.",
          "the assistant gives helpful, detailed, and polite answers to the user's questions.",
          "USER: describe the video in detail. ASSISTANT:", "USER: what is shown in this image? ASSISTANT:",
          "a person walks across the street while cars pass by.", "the quick brown fox jumps over the lazy dog.",
          "two children play with a red ball on the grass near a house.", "the camera moves slowly from left to right over a city at night.",
          "USER: \nDescribe the video in detail. ASSISTANT: The video shows a dog.", "What happens next?\nA man opens the door."]


def g14(R, llm2, tower, enc, proj2, vc, gen):
    import ast
    from PIL import Image
    from tokenizers import Tokenizer, decoders, models, normalizers, processors, trainers
    from transformers import AutoTokenizer, CLIPImageProcessor, PreTrainedTokenizerFast
    aurora, utils, consts, templates = R["aurora"], R["utils"], R["consts"], R["templates"]
    # ---- tokenizer: Llama-style BPE (metaspace), ids 0 / 1 / 2 = <unk> / <s> / </s>, BOS prepended by encode() unless add_special_tokens=False
    tk = Tokenizer(models.BPE(unk_token="<unk>"))
    tk.normalizer = normalizers.Sequence([normalizers.Prepend("\u2581"), normalizers.Replace(" ", "\u2581")])
    tk.decoder = decoders.Sequence([decoders.Replace("\u2581", " "), decoders.Fuse(), decoders.Strip(" ", 1, 0)])
    tk.train_from_iterator(CORPUS * 10, trainers.BpeTrainer(vocab_size=300, special_tokens=["<unk>", "<s>", "</s>"]))
    tk.post_processor = processors.TemplateProcessing(single="<s> $A", pair="<s> $A <s> $B", special_tokens=[("<s>", 1)])
    assert tk.get_vocab_size() <= 320
    PreTrainedTokenizerFast(tokenizer_object=tk, bos_token="<s>", eos_token="</s>", unk_token="<unk>", padding_side="right").save_pretrained(OUT)
    tokenizer = AutoTokenizer.from_pretrained(OUT, padding_side="right")                        # inference.py:64-68
    assert tokenizer.bos_token_id == 1 and tokenizer.eos_token_id == 2 and tokenizer.encode("the")[0] == 1
    # ---- the reference's own process_text, executed from its source file
    src = open(os.path.join(G.REF, "inference.py")).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "process_text")

    class _Torch:                                                                                # torch.tensor(ids).cuda() without a GPU
        @staticmethod
        def tensor(x):
            t = torch.tensor(x)
            t.cuda = lambda: t
            return t
    ns = {"DEFAULT_IMAGE_TOKEN": consts.DEFAULT_IMAGE_TOKEN, "IMAGE_TOKEN_INDEX": consts.IMAGE_TOKEN_INDEX, "torch": _Torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "reference:inference.py", "exec"), ns)
    prompt = "Describe the video in detail."
    text_input = consts.DEFAULT_IMAGE_TOKEN + "\n" + prompt                                      # inference.py:80-85 (image branch)
    prompt_text = templates.PROMPT_TEMPLATE["vicuna"]["INSTRUCTION"].format(input=text_input, round=1)
    ids = ns["process_text"](prompt_text, tokenizer)
    processor = CLIPImageProcessor(size=vc.image_size, crop_size=vc.image_size)                  # inference.py:58-63 with the model's own size
    enc.r = int(vc.image_size * vc.image_size / (vc.patch_size ** 2) * (1 - RATIO) / vc.num_hidden_layers)
    for img_seed in range(1414, 1414 + 400):                                                      # first image whose caption has clear margins throughout
        rng = np.random.default_rng(img_seed)
        img = (rng.integers(0, 256, (70, 90, 3)).astype(np.float32) * 0.5 + np.linspace(0, 127, 90)[None, :, None]).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(OUT, "clip.png"))
        image = Image.open(os.path.join(OUT, "clip.png"))
        px = processor(image, return_tensors="pt")["pixel_values"].to(torch.float16)            # inference.py:79 (.to(dtype=torch.float16))
        with torch.no_grad():
            h = tower.pre_layrnorm(tower.embeddings(px.float()))
            out = enc(h, output_hidden_states=True, return_dict=True)
            feats = out.hidden_states[-2][:, 1:]
            vo = proj2(feats.reshape(1, feats.shape[1], -1)).reshape(1, 1, feats.shape[1], -1)
            data = utils.prepare_inputs_labels_for_multimodal(llm=llm2, input_ids=ids, pixel_values=vo)
            emb = data["inputs_embeds"]
            cont = llm2.generate(inputs_embeds=emb, attention_mask=None, do_sample=False, temperature=0.0, top_p=1.0, num_beams=1, max_new_tokens=NEW)
            tok_emb = llm2.get_input_embeddings()(cont[0, :-1])[None] if cont.shape[1] > 1 else emb[:, :0]
            logits = llm2(inputs_embeds=torch.cat([emb, tok_emb], dim=1)).logits[0, emb.shape[1] - 1:]
        top2 = logits.topk(2, dim=-1).values
        rel = ((top2[:, 0] - top2[:, 1]) / logits.abs().max()).tolist()
        if cont.shape[1] >= 8 and min(rel) > 0.025 and bool((logits.argmax(-1) == cont[0]).all()):
            break
    else:
        raise SystemExit("G14: no image with clear margins found")
    text = tokenizer.batch_decode(cont, skip_special_tokens=True)[0]                            # inference.py:97
    print("G14 prompt ids", ids.shape[1], "generated", cont[0].tolist(), "text", repr(text), "min margin", min(rel))
    G.save("g14_cli_e2e.npz", dict(R["versions"]), prompt=np.array(prompt), prompt_text=np.array(prompt_text), input_ids=ids.numpy(),
           pixel_values=px.numpy(), ids=cont[0].numpy(), logits=logits.numpy(), text=np.array(text), ratio=np.array(RATIO), new=np.array(NEW))


if __name__ == "__main__":
    main()
