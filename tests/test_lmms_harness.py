"""CPU, build container only: the adaptor driven by the REFERENCE's own lmms-eval harness.

`/root/reference/src/lmms-eval` is imported (skipped when absent - e.g. on the GPU box); its third-party imports that this image
lacks (loguru, evaluate, sqlitedict, tenacity, sacrebleu, hf_transfer) are stubbed, nothing of lmms_eval itself is.  Then, as `lmms_eval` does:
  * `LMMS_EVAL_PLUGINS=aurora_amd.lmms_plugin` -> `lmms_eval/models/__init__.py:62-70` imports our module, whose class registers
    itself through the reference's `register_model` (api/registry.py:11-24, which asserts it extends `lmms`);
  * `get_model(name).create_from_arg_string(model_args, {"batch_size": .., "device": ..})` (evaluator.py simple_evaluate);
  * a real `ConfigurableTask` (api/task.py:605) over an in-memory dataset builds the `Instance` requests
    (`task.build_all_requests`), and `evaluator.evaluate` runs them: `getattr(lm, "generate_until")(cloned_reqs)`,
    `req.resps.append`, filters, `process_results`, aggregation (evaluator.py:406-457, 519-546).
The engine behind the adaptor is a fake (host logic under test, no GPU): its "caption" encodes (frames, prompt ids, markers), so the
harness's metric can check that every document got ITS OWN response, in the harness's order, whatever order the adaptor batched in."""
import importlib
import logging
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference/src/lmms-eval"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")


def _import_reference_harness():
    if REF not in sys.path:
        sys.path.insert(0, REF)

    def stub(name, **attrs):
        if name in sys.modules:
            return
        try:
            importlib.import_module(name)
        except ImportError:
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m

    lg = logging.getLogger("lmms-eval")

    class _Logger:
        def __getattr__(self, n):
            return getattr(lg, "info" if n == "success" else n, lg.info)

        def remove(self, *a, **k):
            pass

        def add(self, *a, **k):
            return 0

    class _Stop:
        def __or__(self, other):
            return self

    stub("loguru", logger=_Logger())
    stub("evaluate", load=lambda *a, **k: None)
    stub("sqlitedict", SqliteDict=dict)
    stub("tenacity", retry=lambda *a, **k: (lambda f: f), stop_after_attempt=lambda *a, **k: _Stop(), stop_after_delay=lambda *a, **k: _Stop(),
         wait_fixed=lambda *a, **k: None)
    stub("sacrebleu")
    stub("hf_transfer")
    os.environ["LMMS_EVAL_PLUGINS"] = "aurora_amd.lmms_plugin"
    for m in [k for k in sys.modules if k.startswith("aurora_amd.lmms_plugin")]:      # re-import under the real `lmms` base class
        del sys.modules[m]
    models = importlib.import_module("lmms_eval.models")                              # runs the plugin loop (models/__init__.py:62-70)
    registry = importlib.import_module("lmms_eval.api.registry")
    task = importlib.import_module("lmms_eval.api.task")
    evaluator = importlib.import_module("lmms_eval.evaluator")
    return models, registry, task, evaluator


@pytest.fixture
def clean_process_state():
    """the harness import leaves traces (sys.path entry, stub modules, LMMS_EVAL_PLUGINS, the plugin re-imported under the real base
    class): put the process back so that the other test files see what they would have seen without this one"""
    path, mods, env = list(sys.path), dict(sys.modules), os.environ.get("LMMS_EVAL_PLUGINS")
    yield
    sys.path[:] = path
    for k in [k for k in sys.modules if k not in mods]:
        del sys.modules[k]
    sys.modules.update(mods)
    if env is None:
        os.environ.pop("LMMS_EVAL_PLUGINS", None)
    else:
        os.environ["LMMS_EVAL_PLUGINS"] = env


def test_reference_harness_drives_the_adaptor(clean_process_state):
    import datasets
    from tests.test_lmms_plugin import FakeTok, FakeModel, fake_pre
    models, registry, task_mod, evaluator = _import_reference_harness()
    from lmms_eval.api.model import lmms
    P = importlib.import_module("aurora_amd.lmms_plugin.models.auroracap_mi355x")
    cls = registry.get_model("auroracap_mi355x")                                      # the reference's registry lookup
    assert cls is P.AuroraCapMI355X and issubclass(cls, lmms)
    fake = FakeModel()
    lm = cls.create_from_arg_string("pretrained=unused,token_merge_ratio=0.3,max_frames_num=4",
                                    {"batch_size": 3, "device": "cpu", "_model": fake, "_tokenizer": FakeTok(), "_preprocessor": fake_pre})
    assert lm.token_merge_ratio == 0.3 and lm.max_frames_num == 4 and lm.batch_size == 3
    assert lm.rank == 0 and lm.world_size == 1

    n_docs = 7
    rng = np.random.default_rng(0)
    frames = {i: rng.integers(0, 255, (1 + i % 3, 8, 8, 3), dtype=np.uint8) for i in range(n_docs)}
    questions = ["describe clip %d " % i + "x" * (3 * (i % 4)) for i in range(n_docs)]

    def expected(i):                                                                  # what the fake engine answers for document i
        prompt = P.conv_prompt(P.question_with_image_tokens(questions[i], len(frames[i])))
        ids = P.tokenizer_image_token(prompt, FakeTok())
        return " ".join(map(str, [len(frames[i]), len(ids), len(frames[i])]))

    class InMemoryCaptionTask(task_mod.ConfigurableTask):
        def download(self, dataset_kwargs=None):                                      # the one thing the reference would fetch from the hub
            ds = datasets.Dataset.from_dict({"idx": list(range(n_docs)), "question": questions, "answer": [expected(i) for i in range(n_docs)]})
            self.dataset = datasets.DatasetDict({"test": ds})
            self.dataset_no_image = self.dataset

    seen = []

    def process_results(doc, results):
        seen.append((doc["idx"], results[0]))
        return {"exact": float(results[0] == doc["answer"])}

    cfg = dict(task="tiny_vdc", dataset_path="in-memory", test_split="test", output_type="generate_until",
               doc_to_visual=lambda doc: [frames[doc["idx"]]], doc_to_text=lambda doc, kw=None: doc["question"], doc_to_target="answer",
               generation_kwargs={"max_new_tokens": 8, "temperature": 0, "do_sample": False},
               process_results=process_results,
               metric_list=[{"metric": "exact", "aggregation": lambda xs: sum(xs) / len(xs), "higher_is_better": True}])
    t = InMemoryCaptionTask(config=cfg, model_name="auroracap_mi355x")
    lm.task_dict = {"tiny_vdc": {"test": t.dataset["test"]}}                          # what evaluator.simple_evaluate hands the model (lm.task_dict)
    res = evaluator.evaluate(lm=lm, task_dict={"tiny_vdc": t}, limit=None, bootstrap_iters=0, log_samples=True)
    assert res is not None
    got = dict(seen)
    assert sorted(got) == list(range(n_docs))
    for i in range(n_docs):
        assert got[i] == expected(i), (i, got[i], expected(i))
    metric = [v for k, v in res["results"]["tiny_vdc"].items() if k.startswith("exact")]
    assert metric and metric[0] == 1.0
    assert sum(n for n, _ in fake.calls) == n_docs                                    # every request reached the engine exactly once
    assert fake.visual_encoder.ratio == 0.3
