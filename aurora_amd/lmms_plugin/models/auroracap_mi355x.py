"""lmms-eval adaptor for the MI355X AuroraCap path.

Mirrors the reference adaptor `AuroraCap` (src/lmms-eval/lmms_eval/models/auroracap.py:44-65, 344-525): same
constructor arguments and defaults (token_merge_ratio=0.4, max_frames_num=16, conv_template="vicuna_v1"), same
prompt construction (image tokens joined by spaces + "\\n" + context, llava `vicuna_v1` conversation with its system
prompt, llava `tokenizer_image_token`), same generation defaults (max_new_tokens=1024, greedy), same result order.
Differences, all on purpose:
  * the whole chunk of `batch_size` requests is captioned in ONE pass of the HIP engine
    (AuroraModel.caption_batch: batched ViT / prefill groups / 32-wide decode) instead of one clip at a time;
  * resize / crop / normalise run on the GPU (aurora_amd.preprocess, bit-identical to the CLIPImageProcessor call);
  * sampling (temperature > 0), beams and loglikelihood (broken in the reference: :240-290 reference undefined
    names) raise NotImplementedError instead of failing later; slowfast=True runs one clip at a time.

llava (conversation templates, tokenizer_image_token) is a third-party package that is absent from the reference
tree; its two published helpers are restated below.  With lmms_eval importable the class extends `lmms` and
registers as "auroracap_mi355x"; without it the module still imports (duck-typed base) so the host logic is testable.
"""
from __future__ import annotations

import json
import os
import os.path as osp
from typing import Callable, List, Optional, Sequence, Union

import numpy as np
import torch

from ...engine import IMAGE_TOKEN_INDEX

DEFAULT_IMAGE_TOKEN = "<image>"                       # src/xtuner/xtuner/utils/constants.py:5

try:                                                   # pragma: no cover - lmms_eval is not installed in this image
    from lmms_eval.api.model import lmms as _Base
    from lmms_eval.api.registry import register_model as _register
except Exception:                                      # noqa: BLE001
    class _Base:                                       # the attributes lmms.__init__ provides (api/model.py:17-26)
        def __init__(self) -> None:
            self._rank, self._world_size = 0, 1
            self.cache_hook = None

    def _register(*names):
        return lambda cls: cls

# llava.conversation.conv_vicuna_v1 (SeparatorStyle.TWO, sep=" ", sep2="</s>", roles USER / ASSISTANT)
VICUNA_V1_SYSTEM = ("A chat between a curious human and an artificial intelligence assistant. "
                    "The assistant gives helpful, detailed, and polite answers to the human's questions.")


def conv_prompt(question: str, conv_template: str = "vicuna_v1") -> str:
    """conv.append_message(USER, question); conv.append_message(ASSISTANT, None); conv.get_prompt()  (auroracap.py:445-449)."""
    if conv_template != "vicuna_v1":
        raise NotImplementedError(f"conv_template={conv_template!r}: only llava's 'vicuna_v1' (the reference default) is restated")
    return f"{VICUNA_V1_SYSTEM} USER: {question} ASSISTANT:"


def tokenizer_image_token(prompt: str, tokenizer, image_token_index: int = IMAGE_TOKEN_INDEX) -> List[int]:
    """llava.mm_utils.tokenizer_image_token: tokenize the text between <image> markers, keep ONE leading BOS, put
    image_token_index between the chunks (auroracap.py:478)."""
    chunks = [tokenizer(chunk).input_ids for chunk in prompt.split(DEFAULT_IMAGE_TOKEN)]
    ids: List[int] = []
    offset = 0
    if chunks and len(chunks[0]) > 0 and chunks[0][0] == tokenizer.bos_token_id:
        offset = 1
        ids.append(chunks[0][0])
    sep = [image_token_index] * (offset + 1)
    woven = [x for pair in zip(chunks, [sep] * len(chunks)) for x in pair][:-1]
    for x in woven:
        ids.extend(x[offset:])
    return ids


def question_with_image_tokens(context: str, num_images: int) -> str:
    """auroracap.py:418-441: prepend `<image> <image> ...\\n` unless the context already names an image token."""
    if num_images > 0 and DEFAULT_IMAGE_TOKEN not in context:
        return " ".join([DEFAULT_IMAGE_TOKEN] * num_images) + "\n" + context
    return context


def plan_batches(args_list: Sequence[tuple], tok_len: Callable[[str], int], batch_size: int) -> List[List[int]]:
    """utils.Collator(..., grouping=True).get_batched(n=batch_size) as index lists (auroracap.py:347-364): requests
    are grouped by their generation kwargs, sorted inside a group by (-len(tokens), context), chunked."""
    groups: dict = {}
    for i, a in enumerate(args_list):
        groups.setdefault(json.dumps(a[1], sort_keys=True, default=str), []).append(i)
    batches = []
    for idx in groups.values():
        idx = sorted(idx, key=lambda i: (-tok_len(args_list[i][0]), args_list[i][0]))
        batches += [idx[k:k + batch_size] for k in range(0, len(idx), batch_size)]
    return batches


def gen_defaults(gen_kwargs: dict) -> dict:
    """auroracap.py:468-476: max_new_tokens 1024, temperature 0, top_p None, num_beams 1; `until` is accepted and
    dropped exactly like the reference does (it never truncates on it)."""
    g = dict(gen_kwargs)
    until = g.pop("until", None)
    if until is not None and not isinstance(until, (str, list)):
        raise ValueError(f"Expected `gen_kwargs['until']` to be of type Union[str,list] but got {type(until)}")
    g.pop("image_aspect_ratio", None)
    g.setdefault("max_new_tokens", 1024)
    g.setdefault("temperature", 0)
    g.setdefault("top_p", None)
    g.setdefault("num_beams", 1)
    if g["temperature"] and g["temperature"] > 0:
        raise NotImplementedError("sampling (temperature > 0) is not implemented on the MI355X path (greedy only)")
    if g["num_beams"] != 1:
        raise NotImplementedError("beam search is not implemented on the MI355X path (num_beams=1 only)")
    return g


@_register("auroracap_mi355x")
class AuroraCapMI355X(_Base):
    def __init__(self, pretrained_llm: str = "", pretrained_vit: str = "", pretrained: str = "model/PATH", resolution: int = 378,
                 token_merge_ratio: float = 0.4, device: Optional[str] = "cuda", dtype: Optional[Union[str, torch.dtype]] = "auto",
                 batch_size: Optional[Union[int, str]] = 1, conv_template: str = "vicuna_v1", video_decode_backend: str = "pyav",
                 max_frames_num: int = 16, slowfast: bool = False, max_new_tokens: int = 1024, _model=None, _tokenizer=None,
                 _preprocessor=None, **kwargs) -> None:
        super().__init__()
        assert kwargs == {}, f"Unexpected kwargs: {kwargs}"                 # auroracap.py:67
        self.slowfast = bool(slowfast)                                       # first frame unmerged (aurora.py:223-246)
        # one process per GPU (accelerate launch / torchrun).  The harness calls lm.accelerator.gather / wait_for_everyone
        # whenever lm.world_size > 1 (evaluator.py:426-428, 457): DistShim provides exactly those over RCCL.
        if int(os.environ.get("WORLD_SIZE", 1)) > 1:
            from ...parallel import DistShim
            self.accelerator = DistShim()
            self._rank, self._world_size = self.accelerator.process_index, self.accelerator.num_processes
            self._device = torch.device("cuda", self.accelerator.local_process_index) if self.accelerator.backend == "nccl" \
                else torch.device(device)
        else:
            self._rank, self._world_size = 0, 1
            self._device = torch.device(device)
        self.batch_size_per_gpu = int(batch_size)
        self.conv_template = conv_template
        self.token_merge_ratio = float(token_merge_ratio)
        self.video_decode_backend = video_decode_backend
        self.max_frames_num = int(max_frames_num)
        self.resolution = int(resolution)
        self.task_dict = {}
        if _model is not None:                                               # injected parts (tests)
            self._model, self._tokenizer, self._pre = _model, _tokenizer, _preprocessor
        else:                                                                # pragma: no cover - needs a checkpoint + GPU
            from transformers import AutoTokenizer
            from ...model import AuroraModel
            from ...preprocess import FramePreprocessor
            if not osp.isdir(pretrained):
                from huggingface_hub import snapshot_download
                pretrained = snapshot_download(repo_id=pretrained)
            from ...checkpoint import vit_config
            from ...engine import tokens_at_layer, tome_r
            vc = vit_config(osp.join(pretrained, "visual_encoder"))              # patch size / depth from config.json
            per_frame = (self.resolution // vc["patch_size"]) ** 2
            r = tome_r(self.resolution, self.resolution, vc["patch_size"], self.token_merge_ratio, vc["num_hidden_layers"])
            n_kept_max = tokens_at_layer(per_frame + 1, r, vc["num_hidden_layers"] - 1) - 1
            self._engine_max_new = int(max_new_tokens)
            self._model = AuroraModel.from_pretrained(
                pretrained, max_frames=self.max_frames_num + 1, max_batch=min(self.batch_size_per_gpu, 64),
                max_ctx=256 + (self.max_frames_num + 1) * n_kept_max + (per_frame if self.slowfast else 0) + max_new_tokens,
                max_new_tokens=max_new_tokens, slowfast=self.slowfast, device=str(self._device))
            self._tokenizer = AutoTokenizer.from_pretrained(pretrained, trust_remote_code=True, padding_side="right")
            self._pre = FramePreprocessor(image=self.resolution, device=self._device)
        self._config = getattr(self._model, "config", None)
        self._max_length = 2048

    # ---- the small surface the harness reads (auroracap.py:152-198) -------------------------------------------
    config = property(lambda self: self._config)
    tokenizer = property(lambda self: self._tokenizer)
    model = property(lambda self: self._model)
    eot_token_id = property(lambda self: self._tokenizer.eos_token_id)
    max_length = property(lambda self: self._max_length)
    batch_size = property(lambda self: self.batch_size_per_gpu)
    device = property(lambda self: self._device)
    rank = property(lambda self: self._rank)
    world_size = property(lambda self: self._world_size)

    def tok_encode(self, string: str, left_truncate_len=None, add_special_tokens=None) -> List[int]:
        enc = self._tokenizer.encode(string, add_special_tokens=False if add_special_tokens is None else add_special_tokens)
        return enc[-left_truncate_len:] if left_truncate_len else enc

    def tok_decode(self, tokens):
        return self._tokenizer.decode(tokens)

    def flatten(self, input):
        return [j for i in input for j in i]

    def loglikelihood(self, requests):
        raise NotImplementedError("loglikelihood is not functional in the reference adaptor either (auroracap.py:240-290)")

    # ---- visuals -> decoded rgb24 frames [f, H, W, 3] uint8 --------------------------------------------------------
    def load_frames(self, visuals: list) -> np.ndarray:
        """The visual kinds the reference adaptor accepts (auroracap.py:385-409)."""
        from ...preprocess import read_video_pyav
        v0 = visuals[0]
        if isinstance(v0, dict):                                              # {'video_path', 'keyframe'} (VDC keyframes)
            return self.extract_keyframes(v0["video_path"], v0["keyframe"])
        if isinstance(v0, np.ndarray):                                        # already decoded frames
            return v0 if v0.ndim == 4 else np.stack(visuals)
        if isinstance(v0, str):
            if v0.endswith("mp4") or v0.endswith("mkv"):
                if self.video_decode_backend == "decord" and v0.endswith("mp4"):
                    return self.load_video(v0, self.max_frames_num)
                return read_video_pyav(v0, num_frm=self.max_frames_num)
            raise ValueError(f"unsupported visual {v0!r}")
        if hasattr(v0, "convert"):                                            # PIL images (one prompt image token each)
            imgs = [np.asarray(im.convert("RGB")) for im in visuals]
            if len({im.shape for im in imgs}) != 1:
                raise NotImplementedError("images of different sizes in one request")
            return np.stack(imgs)
        raise ValueError(f"unsupported visual type {type(v0)}")

    def load_video(self, video_path, max_frames_num):                        # auroracap.py:306-312 (decord backend)
        from decord import VideoReader, cpu
        vr = VideoReader(video_path, ctx=cpu(0))
        idx = np.linspace(0, len(vr) - 1, max_frames_num, dtype=int).tolist()
        return vr.get_batch(idx).asnumpy()

    def extract_keyframes(self, video_path, keyframes):                      # auroracap.py:314-342
        import av
        container = av.open(video_path)
        stream = container.streams.video[0]
        fps, time_base = stream.average_rate, stream.time_base
        frames = []
        for keyframe in keyframes:
            t = float(keyframe)
            number = int(t * fps)
            container.seek(int(t / time_base))
            got = None
            for packet in container.demux(video=0):
                for frame in packet.decode():
                    if frame.index >= number:
                        got = frame
                        break
                if got is not None:
                    break
            if got is None:                                                   # past the end: take the last frame
                container.seek(-1, any_frame=False)
                for packet in container.demux(video=0):
                    for frame in packet.decode():
                        got = frame
            frames.append(got)
        return np.stack([x.to_ndarray(format="rgb24") for x in frames])

    # ---- generate_until ---------------------------------------------------------------------------------------------
    def _clip(self, args) -> tuple:
        """One request -> (pixel_values [f, 3, R, R] fp16 on the device, input ids with -200 markers)."""
        context, _, doc_to_visual, doc_id, task, split = args
        visuals = self.flatten([doc_to_visual(self.task_dict[task][split][doc_id])])
        if not visuals:
            raise NotImplementedError("text-only requests: the AuroraCap path needs at least one frame")
        frames = self.load_frames(visuals)
        pixel_values = self._pre(torch.from_numpy(np.ascontiguousarray(frames)).to(self._device))
        prompt = conv_prompt(question_with_image_tokens(context, len(frames)), self.conv_template)
        return pixel_values, tokenizer_image_token(prompt, self._tokenizer, IMAGE_TOKEN_INDEX)

    def generate_until(self, requests) -> List[str]:
        import logging
        log = logging.getLogger("lmms-eval")
        args_list = [r.args for r in requests]
        res: List[Optional[str]] = [None] * len(args_list)

        def deliver(i, ids, gen):
            res[i] = self._tokenizer.batch_decode([ids], skip_special_tokens=True)[0]
            if self.cache_hook is not None:
                self.cache_hook.add_partial("generate_until", (args_list[i][0], gen), [res[i]])

        def fail(i, gen, exc):
            """auroracap.py:511-514: a failing request is logged and answered with "" - the evaluation goes on."""
            log.error(f"Error {exc!r} in generating (request {i})")
            res[i] = ""
            if self.cache_hook is not None:
                self.cache_hook.add_partial("generate_until", (args_list[i][0], gen), [""])

        def clip_or_fail(i, gen):
            try:
                return self._clip(args_list[i])
            except Exception as e:                                            # noqa: BLE001 - undecodable video, no frames, ...
                fail(i, gen, e)
                return None

        cap = getattr(self, "_engine_max_new", None) or getattr(getattr(self._model, "engine", None), "max_new_tokens", None)
        stream = hasattr(self._model, "caption_stream") and not self.slowfast
        # with continuous batching a whole generation-kwargs group is one stream (clips are decoded / preprocessed lazily,
        # finished KV slots are re-filled); otherwise the group is cut into chunks of batch_size like the reference does
        for batch in plan_batches(args_list, lambda s: len(self.tok_encode(s)), len(args_list) if stream else self.batch_size):
            gen = gen_defaults(args_list[batch[0]][1])
            if cap is not None and gen["max_new_tokens"] > cap:
                log.warning(f"gen_kwargs max_new_tokens={gen['max_new_tokens']} exceeds the engine capacity {cap} "
                            f"(constructor argument max_new_tokens): captions are cut at {cap} tokens")
            self._model.visual_encoder.reset_tome_r(self.token_merge_ratio)
            if stream:
                order: List[int] = []                                         # stream position -> request index

                def clips():
                    for i in batch:
                        c = clip_or_fail(i, gen)
                        if c is not None:
                            order.append(i)
                            yield c
                for k, ids in self._model.caption_stream(clips(), max_new_tokens=gen["max_new_tokens"],
                                                         on_error=lambda k, e: fail(order[k], gen, e)):
                    deliver(order[k], ids, gen)
            elif self.slowfast:                 # ragged per-frame token counts: one clip at a time through the three calls
                for i in batch:
                    c = clip_or_fail(i, gen)
                    if c is None:
                        continue
                    px, tok_ids = c
                    try:
                        self._model.visual_encoder.reset_tome_r(self.token_merge_ratio)     # the slow-fast forward leaves it at 1.0
                        out = self._model({"pixel_values": px.unsqueeze(0), "input_ids": torch.tensor([tok_ids])}, mode="inference")
                        deliver(i, self._model.llm.generate(**out, do_sample=False, num_beams=1,
                                                            max_new_tokens=gen["max_new_tokens"])[0].tolist(), gen)
                    except Exception as e:                                    # noqa: BLE001
                        fail(i, gen, e)
            else:
                live = [(i, c) for i in batch for c in [clip_or_fail(i, gen)] if c is not None]
                try:
                    ids = self._model.caption_batch([c for _, c in live], max_new_tokens=gen["max_new_tokens"])
                    for (i, _), x in zip(live, ids):
                        deliver(i, x, gen)
                except Exception:                                             # noqa: BLE001 - isolate the failing request
                    for i, c in live:
                        try:
                            deliver(i, self._model.caption_batch([c], max_new_tokens=gen["max_new_tokens"])[0], gen)
                        except Exception as e:                                # noqa: BLE001
                            fail(i, gen, e)
        return res
