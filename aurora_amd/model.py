"""Python operator seam of the reference, re-hosted on the MI355X engine (drop-in for this path only).

The three calls every reference caller makes (inference.py:87-96, gradio_gui.py:83-92,
lmms_eval/models/auroracap.py:500-509):

    model.visual_encoder.reset_tome_r(token_kept_ratio)
    output = model(data, mode="inference")            # data: {"pixel_values", "input_ids"}
    cont = model.llm.generate(**output, do_sample=False, temperature=0.0, top_p=1.0, num_beams=1,
                              max_new_tokens=N)       # LongTensor [1, <= N]: new ids only

keep their names, argument meaning and error behaviour (NotImplementedError for unknown modes,
aurora.py:269-270).  Only greedy decoding exists on this path (`do_sample=False`, inference.py:91).
"""
from __future__ import annotations

from typing import Optional

import torch

from .engine import AuroraCapEngine

DEFAULT_IMAGE_TOKEN = "<image>"          # src/xtuner/xtuner/utils/constants.py:5
IMAGE_TOKEN_INDEX = -200                 # constants.py:4
VICUNA_INSTRUCTION = "USER: {input} ASSISTANT:"   # utils/templates.py:92 (PROMPT_TEMPLATE.vicuna)


def process_text(inputs: str, tokenizer) -> torch.Tensor:
    """inference.py:12-27: split on <image>, BOS only on the first chunk, -200 between chunks -> [1, n] ids."""
    ids = []
    chunks = inputs.split(DEFAULT_IMAGE_TOKEN)
    for idx, chunk in enumerate(chunks):
        enc = tokenizer.encode(chunk) if idx == 0 else tokenizer.encode(chunk, add_special_tokens=False)
        ids.extend(enc)
        if idx != len(chunks) - 1:
            ids.append(IMAGE_TOKEN_INDEX)
    return torch.tensor(ids).unsqueeze(0)


def build_prompt(prompt: str, num_images: int) -> str:
    """inference.py:76-85."""
    return VICUNA_INSTRUCTION.format(input=" ".join([DEFAULT_IMAGE_TOKEN] * num_images) + "\n" + prompt, round=1)


class AuroraEncoder:
    """visual_encoder handle (aurora.py:869-904): only the ratio knob and the r formula live on the host."""

    def __init__(self, engine: AuroraCapEngine, visual_token_merge_ratio: float = 1.0):
        self._e = engine
        self.visual_token_merge_ratio = visual_token_merge_ratio
        self.config = engine.v

    def reset_tome_r(self, visual_token_merge_ratio: float):            # aurora.py:880-881
        self.visual_token_merge_ratio = visual_token_merge_ratio

    def __call__(self, pixel_values: torch.Tensor) -> torch.Tensor:
        r = self._e.tome_r(self.visual_token_merge_ratio, pixel_values.shape[-2], pixel_values.shape[-1])
        return self._e.vit_encode(pixel_values, r)


class _LLM:
    def __init__(self, engine: AuroraCapEngine, eos_token_id: Optional[int]):
        self._e = engine
        self.eos_token_id = eos_token_id
        self.config = engine.l

    def generate(self, inputs_embeds: torch.Tensor = None, do_sample: bool = False, temperature: float = 0.0, top_p: float = 1.0,
                 num_beams: int = 1, max_new_tokens: int = 2048, input_ids=None, position_ids=None, attention_mask=None,
                 past_key_values=None, labels=None, eos_token_id="default", **unused) -> torch.Tensor:
        if inputs_embeds is None:
            raise ValueError("generate() on this path takes inputs_embeds (the output of model(data, mode='inference'))")
        if do_sample or num_beams != 1:
            raise NotImplementedError("only greedy decoding (do_sample=False, num_beams=1) is implemented on the MI355X path")
        if inputs_embeds.dim() != 3 or inputs_embeds.shape[0] != 1:
            raise ValueError(f"inputs_embeds must be [1, L, {self.config['hidden_size']}], got {tuple(inputs_embeds.shape)}")
        eos = self.eos_token_id if eos_token_id == "default" else eos_token_id
        L, d = inputs_embeds.shape[1], inputs_embeds.shape[2]
        pad = torch.zeros((L + 31) // 32 * 32, d, dtype=torch.float16, device=self._e.dev)
        pad[:L] = inputs_embeds[0].to(device=self._e.dev, dtype=torch.float16)
        n = min(max_new_tokens, self._e.max_new_tokens)
        ids = self._e.generate([pad], [L], n, eos_id=eos)[0]
        return torch.tensor([ids], dtype=torch.long)


class AuroraModel:
    def __init__(self, engine: AuroraCapEngine, eos_token_id: Optional[int] = 2, slowfast: bool = False):
        self.engine = engine
        self.slowfast = slowfast                                                  # aurora.py:81 (AuroraModel(slowfast=...))
        self.visual_encoder = AuroraEncoder(engine)
        self.llm = _LLM(engine, eos_token_id)

    @classmethod
    def from_pretrained(cls, path: str, *, max_frames: int = 16, max_ctx: int = 8192, max_new_tokens: int = 2048,
                        max_batch: int = 1, slowfast: bool = False, **kw):
        from .checkpoint import load_auroracap
        cfg, weights = load_auroracap(path)
        eng = AuroraCapEngine(cfg, weights, max_frames=max_frames, max_batch=max_batch, max_ctx=max_ctx,
                              max_new_tokens=max_new_tokens, **kw)
        return cls(eng, cfg["llm"].get("eos_token_id", 2), slowfast=slowfast)

    def caption_batch(self, clips, max_new_tokens: int = 2048, eos_token_id="default"):
        """Several clips through the three calls at once (what a harness gets by looping inference.py:87-96 over its
        requests): clips = [(pixel_values [f, c, h, w], input_ids with -200 markers), ...] -> list of new-token id lists,
        identical per clip to the one-at-a-time calls.  Uses visual_encoder.visual_token_merge_ratio."""
        eos = self.llm.eos_token_id if eos_token_id == "default" else eos_token_id
        out = []
        B = self.engine.max_batch
        n = min(max_new_tokens, self.engine.max_new_tokens)
        for c0 in range(0, len(clips), B):
            group = [(px, ids.tolist() if torch.is_tensor(ids) else list(ids)) for px, ids in clips[c0:c0 + B]]
            out.extend(self.engine.caption_batch(group, self.visual_encoder.visual_token_merge_ratio, n, eos))
        return out

    def caption_stream(self, clips, max_new_tokens: int = 2048, eos_token_id="default", check_every: int = 16, on_error=None,
                       overlap=None):
        """Continuous batching over an iterable of (pixel_values, input_ids): yields (index, new-token ids) as captions
        finish (EOS or max_new_tokens), re-filling freed KV slots with the next clips.  Per-clip results equal the
        one-at-a-time calls.  overlap (default: on for an engine with spare KV sequences): the next clips' front ends run
        ahead on a CU-masked stream while the slots decode (AuroraCapEngine.caption_stream)."""
        eos = self.llm.eos_token_id if eos_token_id == "default" else eos_token_id
        n = min(max_new_tokens, self.engine.max_new_tokens)
        it = ((px, ids.tolist() if torch.is_tensor(ids) else list(ids)) for px, ids in clips)
        yield from self.engine.caption_stream(it, self.visual_encoder.visual_token_merge_ratio, n, eos, check_every=check_every,
                                              on_error=on_error, overlap=overlap)

    def __call__(self, data: dict, data_samples=None, mode: str = "loss"):
        return self.forward(data, data_samples, mode)

    def forward(self, data: dict, data_samples=None, mode: str = "loss"):
        if mode != "inference":
            # aurora.py:261-270: 'loss' / 'predict' / 'tensor' are training-side modes, anything else raises
            raise NotImplementedError(f"mode={mode!r}: only mode='inference' exists on the MI355X path")
        if "pixel_values" not in data:
            raise KeyError("data['pixel_values'] is required")
        px = data["pixel_values"]
        if px.dim() == 4:                                   # single image -> one-frame video (aurora.py:217-218)
            px = px.unsqueeze(1)
        if px.dim() != 5 or px.shape[0] != 1:
            raise ValueError(f"pixel_values must be [1, f, c, h, w] or [1, c, h, w], got {tuple(data['pixel_values'].shape)}")
        ids = data["input_ids"]
        ids = ids[0].tolist() if torch.is_tensor(ids) else list(ids[0])
        if self.slowfast and px.shape[1] != 1:                                    # aurora.py:223-246
            low = self.visual_encoder(px[0, 1:])                                  # frames 1.. at the current ratio
            self.visual_encoder.visual_token_merge_ratio = 1.0                    # :231 - the reference leaves it at 1.0 too
            high = self.visual_encoder(px[0, :1])                                 # frame 0 unmerged
            flat = torch.cat([high.reshape(-1, high.shape[-1]), low.reshape(-1, low.shape[-1])], 0)
            counts = [high.shape[1]] + [low.shape[1]] * low.shape[0]
            plan = self.engine.splice_plan(ids, px.shape[1], counts, strict=True)  # utils.py:297-431: marker i <- frame i
            embeds, L = self.engine.project_splice(flat, plan=plan)
        else:
            vis = self.visual_encoder(px[0])                                      # aurora.py:249-253
            embeds, L = self.engine.project_splice(vis, ids)                     # aurora.py:254-258
        return {"input_ids": None, "position_ids": None, "attention_mask": None, "past_key_values": None,
                "inputs_embeds": embeds[:L].unsqueeze(0), "labels": None}         # utils.py:288-295
