"""CPU: host logic of the lmms-eval adaptor with injected fakes for model / tokenizer /
preprocessor - prompt construction, llava tokenizer_image_token semantics, batching order, generation defaults,
plugin-loader contract.  The engine itself is covered by the GPU suites."""
import importlib
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from aurora_amd.lmms_plugin.models import auroracap_mi355x as P


class FakeTok:
    bos_token_id, eos_token_id, pad_token_id = 1, 2, None

    def _ids(self, s):
        return [10 + (ord(c) % 50) for c in s]

    def __call__(self, s):
        return SimpleNamespace(input_ids=[self.bos_token_id] + self._ids(s))

    def encode(self, s, add_special_tokens=False):
        return ([self.bos_token_id] if add_special_tokens else []) + self._ids(s)

    def decode(self, ids):
        return " ".join(map(str, ids))

    def batch_decode(self, ids, skip_special_tokens=True):
        return [" ".join(map(str, x)) for x in ids]


class FakeModel:
    def __init__(self):
        self.visual_encoder = SimpleNamespace(ratio=None, reset_tome_r=lambda r: setattr(self.visual_encoder, "ratio", r))
        self.calls = []

    def caption_batch(self, clips, max_new_tokens=2048):
        self.calls.append((len(clips), max_new_tokens))
        return [[int(px.shape[0]), len(ids), sum(1 for t in ids if t == -200)] for px, ids in clips]


def fake_pre(frames):
    return torch.zeros(frames.shape[0], 3, 4, 4, dtype=torch.float16)


def make(batch_size=4, **kw):
    return P.AuroraCapMI355X(pretrained="unused", device="cpu", batch_size=batch_size, _model=FakeModel(), _tokenizer=FakeTok(),
                             _preprocessor=fake_pre, **kw)


def test_reference_defaults_and_surface():
    m = make()
    assert (m.token_merge_ratio, m.max_frames_num, m.conv_template, m.resolution) == (0.4, 16, "vicuna_v1", 378)   # auroracap.py:56-62
    assert m.batch_size == 4 and m.rank == 0 and m.world_size == 1 and m.eot_token_id == 2
    with pytest.raises(AssertionError):
        make(bogus=1)
    assert make(slowfast=True).slowfast is True and m.slowfast is False
    with pytest.raises(NotImplementedError):
        m.loglikelihood([])


def test_vicuna_v1_prompt_and_image_tokens():
    q = P.question_with_image_tokens("Describe the video in detail.", 3)
    assert q == "<image> <image> <image>\nDescribe the video in detail."
    assert P.question_with_image_tokens("<image>\nalready there", 3) == "<image>\nalready there"
    assert P.question_with_image_tokens("no visuals", 0) == "no visuals"
    p = P.conv_prompt(q)
    assert p == ("A chat between a curious user and an artificial intelligence assistant. The assistant gives helpful, "
                 "detailed, and polite answers to the user's questions. USER: <image> <image> <image>\n"
                 "Describe the video in detail. ASSISTANT:")
    with pytest.raises(NotImplementedError):
        P.conv_prompt(q, "llava_llama_3")


def test_tokenizer_image_token_llava_semantics():
    tok = FakeTok()
    ids = P.tokenizer_image_token("ab<image> <image>cd", tok)
    # one BOS kept, per-chunk BOS stripped, -200 between chunks
    assert ids == [1] + tok._ids("ab") + [-200] + tok._ids(" ") + [-200] + tok._ids("cd")
    assert P.tokenizer_image_token("<image>x", tok) == [1, -200] + tok._ids("x")
    assert P.tokenizer_image_token("plain", tok) == [1] + tok._ids("plain")

    class NoBos(FakeTok):
        def __call__(self, s):
            return SimpleNamespace(input_ids=self._ids(s))
    assert P.tokenizer_image_token("a<image>b", NoBos()) == NoBos()._ids("a") + [-200] + NoBos()._ids("b")


def test_plan_batches_groups_sorts_and_covers():
    args = [("bb", {"max_new_tokens": 8}), ("a", {"max_new_tokens": 8}), ("cccc", {"max_new_tokens": 16}),
            ("ddd", {"max_new_tokens": 8}), ("eeeee", {"max_new_tokens": 8}), ("ab", {"max_new_tokens": 8})]
    batches = P.plan_batches(args, len, 2)
    assert sorted(i for b in batches for i in b) == list(range(6))
    for b in batches:                                   # one generation-kwargs group per batch
        assert len({str(args[i][1]) for i in b}) == 1 and len(b) <= 2
    g8 = [i for b in batches for i in b if args[i][1]["max_new_tokens"] == 8]
    assert g8 == [4, 3, 5, 0, 1]                        # longest first, ties by context string ("ab" < "bb")


def test_gen_defaults():
    g = P.gen_defaults({"until": ["\n"], "image_aspect_ratio": "pad"})
    assert g == {"max_new_tokens": 1024, "temperature": 0, "top_p": None, "num_beams": 1}     # auroracap.py:468-476
    assert P.gen_defaults({"max_new_tokens": 64})["max_new_tokens"] == 64
    with pytest.raises(ValueError):
        P.gen_defaults({"until": 3})
    with pytest.raises(NotImplementedError):
        P.gen_defaults({"temperature": 0.7})
    with pytest.raises(NotImplementedError):
        P.gen_defaults({"num_beams": 4})


def test_generate_until_routes_and_restores_order():
    m = make(batch_size=3)
    docs = {i: np.zeros((2 + i % 3, 8, 8, 3), np.uint8) for i in range(7)}
    m.task_dict = {"vdc": {"test": docs}}
    ctxs = ["x" * (1 + (i * 5) % 7) for i in range(7)]
    reqs = [SimpleNamespace(args=(ctxs[i], {"max_new_tokens": 32} if i % 2 else {}, lambda d: [d], i, "vdc", "test")) for i in range(7)]
    out = m.generate_until(reqs)
    assert len(out) == 7 and m.model.visual_encoder.ratio == 0.4
    tok = FakeTok()
    for i, text in enumerate(out):                     # each result belongs to ITS request, whatever the batch order was
        f = 2 + i % 3
        want_ids = P.tokenizer_image_token(P.conv_prompt(P.question_with_image_tokens(ctxs[i], f)), tok)
        assert text == f"{f} {len(want_ids)} {f}"
    assert sorted(m.model.calls) == sorted([(3, 1024), (1, 1024), (3, 32)])


def test_plugin_loader_contract():
    """lmms_eval/models/__init__.py:62-70: import <plugin>.models, read AVAILABLE_MODELS, import each class."""
    m = importlib.import_module("aurora_amd.lmms_plugin.models")
    for name, cls in m.AVAILABLE_MODELS.items():
        mod = importlib.import_module(f"aurora_amd.lmms_plugin.models.{name}")
        assert hasattr(mod, cls)


def test_a_failing_request_becomes_an_empty_caption_and_the_rest_go_on():
    """auroracap.py:511-514: the reference logs the exception and answers "" for that request only."""
    m = make(batch_size=4)
    docs = {i: np.zeros((2, 8, 8, 3), np.uint8) for i in range(5)}
    docs[2] = "clip.avi"                                   # unsupported visual -> load_frames raises
    m.task_dict = {"vdc": {"test": docs}}
    reqs = [SimpleNamespace(args=("q%d" % i, {}, lambda d: [d], i, "vdc", "test")) for i in range(5)]
    out = m.generate_until(reqs)
    assert out[2] == "" and all(o for i, o in enumerate(out) if i != 2)

    class Flaky(FakeModel):                                # the engine rejects one clip of the batch (e.g. context too long)
        def caption_batch(self, clips, max_new_tokens=2048):
            if any(px.shape[0] == 3 for px, _ in clips):
                raise RuntimeError("seq_len + max_new exceeds max_ctx")
            return super().caption_batch(clips, max_new_tokens)
    m2 = P.AuroraCapMI355X(pretrained="unused", device="cpu", batch_size=4, _model=Flaky(), _tokenizer=FakeTok(), _preprocessor=fake_pre)
    docs2 = {i: np.zeros((3 if i == 1 else 2, 8, 8, 3), np.uint8) for i in range(4)}
    m2.task_dict = {"vdc": {"test": docs2}}
    out2 = m2.generate_until([SimpleNamespace(args=("q", {}, lambda d: [d], i, "vdc", "test")) for i in range(4)])
    assert out2[1] == "" and all(out2[i] for i in (0, 2, 3))


def test_stream_mode_reports_rejected_clips_through_on_error():
    class Streaming(FakeModel):
        def caption_stream(self, clips, max_new_tokens=2048, on_error=None):
            for k, (px, ids) in enumerate(clips):
                if px.shape[0] == 5:
                    on_error(k, ValueError("too many frames"))
                else:
                    yield k, [int(px.shape[0]), k]
    m = P.AuroraCapMI355X(pretrained="unused", device="cpu", batch_size=2, _model=Streaming(), _tokenizer=FakeTok(), _preprocessor=fake_pre)
    docs = {0: np.zeros((2, 8, 8, 3), np.uint8), 1: "bad.avi", 2: np.zeros((5, 8, 8, 3), np.uint8), 3: np.zeros((4, 8, 8, 3), np.uint8)}
    m.task_dict = {"vdc": {"test": docs}}
    out = m.generate_until([SimpleNamespace(args=("same", {}, lambda d: [d], i, "vdc", "test")) for i in range(4)])
    # request 1 never reaches the stream (decode failure), request 2 is rejected by the engine; 0 and 3 keep THEIR results
    assert out == ["2 0", "", "", "4 2"]


def test_constructor_plumbing_reaches_from_pretrained_with_config_derived_capacity(tmp_path, monkeypatch):
    """INTEGRATION.md option C, single process, default device="cuda": the real constructor path with a stub engine
    (ADVICE r01): capacity comes from visual_encoder/config.json, not from hard-coded ViT-H/14 numbers."""
    import json
    import sys
    import types
    (tmp_path / "visual_encoder").mkdir()
    (tmp_path / "visual_encoder" / "config.json").write_text(json.dumps(dict(
        hidden_size=64, num_attention_heads=4, num_hidden_layers=8, intermediate_size=128, patch_size=16, image_size=64)))
    (tmp_path / "config.json").write_text(json.dumps(dict(                      # the language model's config.json: vicuna-7b dims
        hidden_size=4096, num_attention_heads=32, num_hidden_layers=32, intermediate_size=11008, vocab_size=32000)))
    seen = {}

    class StubModel:
        config = None

        @classmethod
        def from_pretrained(cls, path, **kw):
            seen.update(kw, path=path)
            return cls()
    import aurora_amd.model as M
    import aurora_amd.preprocess as PP
    monkeypatch.setattr(M, "AuroraModel", StubModel)
    monkeypatch.setattr(PP, "FramePreprocessor", lambda image, device: ("pre", image, device))
    fake_tf = types.ModuleType("transformers")
    fake_tf.AutoTokenizer = SimpleNamespace(from_pretrained=lambda *a, **k: FakeTok())
    monkeypatch.setitem(sys.modules, "transformers", fake_tf)
    m = P.AuroraCapMI355X(pretrained=str(tmp_path), resolution=64, token_merge_ratio=0.5, batch_size=3, max_frames_num=4, max_new_tokens=100)
    # 64/16 = 4 -> 16 patches + CLS; r = int(16 * 0.5 / 8) = 1 per layer over 7 layers -> 17 - 7 = 10 tokens, 9 without CLS
    assert seen["max_frames"] == 5 and seen["max_batch"] == 3 and seen["max_new_tokens"] == 100 and seen["device"] == "cuda"
    assert seen["max_ctx"] == 256 + 5 * 9 + 100
    assert m.world_size == 1 and not hasattr(m, "accelerator")
    # front ends are prefetched beside the decode (spare KV sequences) only where that many slots make a K / V-bound decode step
    assert seen["spare_slots"] == 0                                              # 3 slots x 200 tokens of context: the weight stream
    P.AuroraCapMI355X(pretrained=str(tmp_path), resolution=64, token_merge_ratio=0.5, batch_size=64, max_frames_num=4, max_new_tokens=2000)
    assert seen["max_batch"] == 64 and seen["spare_slots"] == 4                  # 64 slots x ~1.2 k tokens: K / V-bound


def test_engine_device_normalisation(monkeypatch):
    from aurora_amd.engine import AuroraCapEngine
    from aurora_amd._lib import AuroraHipError
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 3)
    assert AuroraCapEngine.normalize_device("cuda") == torch.device("cuda", 3)        # torch.cuda.set_device needs an index
    assert AuroraCapEngine.normalize_device("cuda:1") == torch.device("cuda", 1)
    with pytest.raises(AuroraHipError):
        AuroraCapEngine.normalize_device("cpu")


def test_g12_prompts_and_ids_of_the_reference_adaptor_helpers():
    """tests/golden/g12_adaptor_prompts.npz: prompt strings from the reference's own vendored
    llava.conversation.conv_templates["vicuna_v1"] and ids from its llava.mm_utils.tokenizer_image_token
    (lmms_eval/models/auroracap.py:445-449, 478), generated by tests/golden/make_golden_adaptor.py in the build container.
    The restated helpers of the MI355X adaptor must reproduce every one of them exactly."""
    import json
    from tests.util import golden
    g = golden("g12_adaptor_prompts.npz")
    assert int(g["image_token_index"]) == P.IMAGE_TOKEN_INDEX and str(g["default_image_token"]) == P.DEFAULT_IMAGE_TOKEN

    class CharTok:                                        # the fixture's tokenizer rule: ids = [1 if bos] + [10 + ord(c) % 50]
        bos_token_id = 1

        def __init__(self, add_bos):
            self.add_bos = add_bos

        def __call__(self, s):
            return SimpleNamespace(input_ids=([1] if self.add_bos else []) + [10 + (ord(c) % 50) for c in s])
    cases = json.loads(str(g["cases"]))
    assert len(cases) == 6
    for i, (context, n_img) in enumerate(cases):
        prompt = P.conv_prompt(P.question_with_image_tokens(context, n_img))
        assert prompt == str(g[f"prompt{i}"]), i
        for tag, tok in (("bos", CharTok(True)), ("nobos", CharTok(False))):
            assert P.tokenizer_image_token(prompt, tok) == g[f"ids{i}_{tag}"].tolist(), (i, tag)
    # round 3: the reference helper driven by a real HF tokenizer (fixture G14's, tests/golden/ckpt_tiny) - same ids from the restatement
    import os
    from transformers import AutoTokenizer
    hf = AutoTokenizer.from_pretrained(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_tiny"), padding_side="right")
    for i, (context, n_img) in enumerate(cases):
        prompt = P.conv_prompt(P.question_with_image_tokens(context, n_img))
        got = P.tokenizer_image_token(prompt, hf)
        assert got == g[f"ids{i}_hf"].tolist(), i
        assert got.count(P.IMAGE_TOKEN_INDEX) == prompt.count(P.DEFAULT_IMAGE_TOKEN) and got[0] == hf.bos_token_id and got.count(hf.bos_token_id) == 1
