"""CPU tests of the host side: prompt/text processing, checkpoint mapping, clip sharding + gather (gloo, world 2)."""
import json
import os
import socket

import numpy as np
import pytest
import torch

from aurora_amd import checkpoint, model, parallel
from oracle import aurora_oracle as O
from tests.util import rand_llm_weights, rand_proj_weights, rand_vit_weights


class FakeTok:
    def encode(self, s, add_special_tokens=True):
        return ([1] if add_special_tokens else []) + [100 + len(s)]


def test_prompt_and_process_text_match_reference_semantics():
    text = model.build_prompt("Describe the video in detail.", 3)
    assert text == O.build_prompt("Describe the video in detail.", 3)
    ids = model.process_text(text, FakeTok())
    ref = O.process_text(text, lambda s, sp: FakeTok().encode(s, add_special_tokens=sp))
    assert ids.shape == (1, len(ref)) and ids[0].tolist() == ref
    assert (ids == model.IMAGE_TOKEN_INDEX).sum() == 3


def test_checkpoint_directory_round_trip(tmp_path):
    from safetensors.torch import save_file
    vcfg = dict(hidden_size=64, num_attention_heads=4, num_hidden_layers=2, intermediate_size=128, patch_size=14, image_size=56,
                hidden_act="gelu", layer_norm_eps=1e-6)
    lcfg = dict(hidden_size=128, num_attention_heads=4, num_hidden_layers=2, intermediate_size=256, vocab_size=320,
                rms_norm_eps=1e-5, rope_theta=1e4, rope_factor=4.0)
    vw, lw, pw = rand_vit_weights(vcfg, 1), rand_llm_weights(lcfg, 2), rand_proj_weights(64, 128, 3)
    root = tmp_path / "ckpt"
    (root / "visual_encoder").mkdir(parents=True)
    (root / "projector").mkdir()
    vs = {"vision_model.embeddings.patch_embedding.weight": vw["patch_embedding.weight"],
          "vision_model.embeddings.class_embedding": vw["class_embedding"],
          "vision_model.embeddings.position_embedding.weight": vw["position_embedding.weight"],
          "vision_model.pre_layrnorm.weight": vw["pre_layrnorm.weight"], "vision_model.pre_layrnorm.bias": vw["pre_layrnorm.bias"],
          "vision_model.post_layernorm.weight": torch.ones(64), "vision_model.post_layernorm.bias": torch.zeros(64)}
    for i, l in enumerate(vw["layers"]):
        for k, t in l.items():
            mod = "self_attn." if "proj" in k else ("mlp." if k.startswith("fc") else "")
            vs[f"vision_model.encoder.layers.{i}.{mod}{k}"] = t
    save_file({k: v.contiguous() for k, v in vs.items()}, str(root / "visual_encoder" / "model.safetensors"))
    json.dump({"vision_config": dict(vcfg, num_channels=3)}, open(root / "visual_encoder" / "config.json", "w"))
    ls = {"model.embed_tokens.weight": lw["embed_tokens.weight"], "model.norm.weight": lw["norm.weight"], "lm_head.weight": lw["lm_head.weight"]}
    for i, l in enumerate(lw["layers"]):
        for k, t in l.items():
            mod = "self_attn." if k[0] in "qkvo" and "proj" in k else ("mlp." if "proj" in k else "")
            ls[f"model.layers.{i}.{mod}{k}"] = t
    save_file({k: v.contiguous() for k, v in ls.items()}, str(root / "model.safetensors"))
    json.dump(dict(hidden_size=128, num_attention_heads=4, num_key_value_heads=4, num_hidden_layers=2, intermediate_size=256,
                   vocab_size=320, rms_norm_eps=1e-5, rope_theta=1e4, rope_scaling={"type": "linear", "factor": 4.0},
                   eos_token_id=2), open(root / "config.json", "w"))
    save_file({k: v.contiguous() for k, v in pw.items()}, str(root / "projector" / "model.safetensors"))
    json.dump({"visual_hidden_size": pw["model.0.weight"].shape[1], "llm_hidden_size": 128, "depth": 2, "hidden_act": "gelu", "bias": True},
              open(root / "projector" / "config.json", "w"))            # what ProjectorConfig.save_pretrained writes (configuration_projector.py:9-22)
    cfg, w = checkpoint.load_auroracap(str(root))
    assert cfg["vit"]["hidden_act"] == "gelu" and cfg["vit"]["layer_norm_eps"] == 1e-6        # read from config, not hard-coded
    assert cfg["llm"]["rope_factor"] == 4.0 and cfg["llm"]["eos_token_id"] == 2
    assert torch.equal(w["vit"]["layers"][1]["fc2.weight"], vw["layers"][1]["fc2.weight"])
    assert torch.equal(w["llm"]["layers"][1]["down_proj.weight"], lw["layers"][1]["down_proj.weight"])
    assert torch.equal(w["projector"]["model.2.bias"], pw["model.2.bias"])
    # the oracle consumes the loaded dict directly (same naming)
    px = torch.randn(1, 3, 56, 56)
    f = O.vit_features(px, w["vit"], cfg["vit"], 0.5)
    assert f.shape[0] == 1 and f.shape[2] == 64
    with pytest.raises(FileNotFoundError):
        checkpoint.load_auroracap(str(tmp_path / "nope"))


def test_shard_and_merge_round_robin():
    for n, world in [(64, 8), (10, 4), (3, 8), (0, 2)]:
        shards = {r: parallel.shard_clips(n, r, world) for r in range(world)}
        assert sorted(sum(shards.values(), [])) == list(range(n))
        per_rank = {r: [[i, i + 1] for i in shards[r]] for r in range(world)}
        assert parallel.merge_round_robin(per_rank, n) == [[i, i + 1] for i in range(n)]


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_clips, max_new = 5, 6
    mine = parallel.shard_clips(n_clips, rank, world)
    local = [[c * 10 + k for k in range(1 + (c % max_new))] for c in mine]      # ragged lengths
    per_rank = parallel.gather_results(local, max_new, clips_per_rank=3, device="cpu")
    merged = parallel.merge_round_robin(per_rank, n_clips)
    q.put((rank, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_results_gloo_world2():
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [[c * 10 + k for k in range(1 + (c % 6))] for c in range(5)]
    assert res[0] == want and res[1] == want


def test_cli_argument_surface_matches_reference():
    """inference.py keeps the reference CLI's flags, types and defaults (reference inference.py:31-40), so existing
    command lines run unchanged; unknown flags are rejected; beams > 1 exits with a message."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "inference.py", "--help"], cwd=root, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--model_path", "--prompt", "--visual_input", "--num_frm", "--token_kept_ratio", "--temperature", "--top_p",
                 "--num_beams", "--max_new_tokens"):
        assert flag in out.stdout, flag
    import importlib.util
    spec = importlib.util.spec_from_file_location("inference_cli", os.path.join(root, "inference.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    args = mod.build_parser().parse_args([])
    assert (args.model_path, args.prompt, args.visual_input) == ("wchai/AuroraCap-7B-IMG-xtuner", "Describe the video in detail.", "output.png")
    assert (args.num_frm, args.token_kept_ratio, args.temperature, args.top_p, args.num_beams, args.max_new_tokens) == (8, 0.8, 0.0, 1.0, 1, 2048)
    bad = subprocess.run([sys.executable, "inference.py", "--num_beams", "4"], cwd=root, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "greedy" in (bad.stderr + bad.stdout)


def _shim_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      AURORA_DIST_BACKEND="gloo")
    import torch
    from types import SimpleNamespace
    from aurora_amd.lmms_plugin.models import auroracap_mi355x as P

    class Tok:
        bos_token_id, eos_token_id = 1, 2

    lm = P.AuroraCapMI355X(pretrained="unused", device="cpu", batch_size=1, _model=SimpleNamespace(), _tokenizer=Tok(),
                           _preprocessor=lambda f: f)
    # what the harness does with the adaptor when world_size > 1 (evaluator.py:426-428, 457)
    instances_rnk = torch.tensor(3 + lm.rank, device=lm.device)
    gathered = lm.accelerator.gather(instances_rnk).cpu().detach().numpy().tolist()
    lm.accelerator.wait_for_everyone()
    q.put((lm.rank, lm.world_size, gathered))
    lm.accelerator.wait_for_everyone()
    torch.distributed.destroy_process_group()


def test_lmms_adaptor_provides_the_accelerator_calls_world2():
    """with WORLD_SIZE > 1 the harness calls lm.accelerator.gather / wait_for_everyone - they must exist."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shim_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert got == [(0, 2, [3, 4]), (1, 2, [3, 4])]


def test_cu_mask_words_enable_k_cus_of_every_xcd():
    """aurora_amd/streams.py: the driver deals mask bit i to CU i // 8 of XCD i % 8, so "k CUs of every XCD" is bits with
    i // 8 < k (or >= 32 - k from the top); the two ends must be complementary when they add up to 32."""
    from aurora_amd.streams import cu_mask_words
    lo, hi = cu_mask_words(16), cu_mask_words(16, from_top=True)
    assert len(lo) == 8 and all(w == 0xFFFFFFFF for w in lo[:4]) and all(w == 0 for w in lo[4:])
    assert [a ^ b for a, b in zip(lo, hi)] == [0xFFFFFFFF] * 8 and [a & b for a, b in zip(lo, hi)] == [0] * 8
    for k in (4, 12, 20, 32):
        bits = [i for i in range(256) if cu_mask_words(k)[i >> 5] >> (i & 31) & 1]
        assert len(bits) == 8 * k and all(i // 8 < k for i in bits)
        for xcd in range(8):
            assert sum(1 for i in bits if i % 8 == xcd) == k
    import pytest
    with pytest.raises(ValueError):
        cu_mask_words(0)
    with pytest.raises(ValueError):
        cu_mask_words(33)


def test_bench_config_presets_name_the_baseline_configs(monkeypatch):
    """`bench.py --config cfgN` = BASELINE.json configs[N-1]; explicit flags win; the default is configs[1]."""
    import json as _json
    import sys as _sys
    import bench
    base = _json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "BASELINE.json")))["configs"]
    for name, c in bench.CONFIGS.items():
        txt = base[c["index"]]
        st = c["set"]
        assert f"{st['num_frm']}-frame" in txt, (name, txt)
        if "token_kept_ratio=" in txt or "r=" in txt:
            assert (f"token_kept_ratio={st['token_kept_ratio']}" in txt) or (f"r={st['token_kept_ratio']}" in txt), (name, txt)
        if "tokens" in txt:
            assert f"{st['max_new_tokens']} tokens" in txt, (name, txt)
    assert bench.CONFIGS["cfg4"]["set"]["batch"] * 8 == 64                      # "batch of 64 clips sharded 8-way"
    monkeypatch.setattr(_sys, "argv", ["bench.py", "--config", "cfg5", "--batch", "7"])
    a = bench.parse()
    assert (a.num_frm, a.token_kept_ratio, a.max_new_tokens, a.batch) == (8, 0.8, 2048, 7)
    monkeypatch.setattr(_sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.num_frm, a.token_kept_ratio, a.max_new_tokens, a.batch, a.config_name) == (8, 0.3, 256, 128, "cfg2")


def test_bench_schedule_rule_overlaps_only_kv_dominated_decodes():
    """`bench.py`'s default schedule runs the front ends beside the decode only where the decode step is dominated by the K / V
    stream (cfg2 / cfg3 / cfg5: 128 / 96 / 48 slots); a step of 8 slots (cfg4 = BASELINE configs[3]'s per-GPU share) is the
    weight stream, and its front ends go between the chunks (measured: 5.8 against 5.4 captions/s)."""
    import bench
    from aurora_amd import synthetic as S
    llm = S.VICUNA_7B_16K
    assert bench.overlap_pays(128, 2142, 256, llm) and bench.overlap_pays(96, 2766, 512, llm) and bench.overlap_pays(48, 4870, 2048, llm)
    assert not bench.overlap_pays(8, 2142, 256, llm) and not bench.overlap_pays(1, 2142, 256, llm)


def test_bench_masked_steps_follow_the_front_end():
    """The overlapped schedule masks the decode for as many steps of a chunk as the front end lasts beside them (cfg3: 376 ms / 34.9 ms
    -> 11 of 21; cfg5: 571 / 30.1 -> 19 of 170), never more than the chunk's whole steps (default config: 274 / 37.8 -> 8 needed,
    7 available) and never none."""
    import bench
    assert bench.masked_steps_per_chunk(376.1, 34.9, 21) == 11
    assert bench.masked_steps_per_chunk(570.9, 30.07, 170) == 19
    assert bench.masked_steps_per_chunk(273.8, 37.8, 7) == 7
    assert bench.masked_steps_per_chunk(5.0, 40.0, 7) == 1 and bench.masked_steps_per_chunk(0.0, 0.0, 3) == 1


def test_rank_cpu_slices_are_disjoint_and_cover_the_numa_node():
    """Round 5: ranks are bound to disjoint slices of the cores local to their GPU's NUMA node."""
    from aurora_amd.parallel import parse_cpulist, rank_cpu_slice
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    assert parse_cpulist("") == []
    node = parse_cpulist("0-63,128-191")                       # one socket of a 2 x 64-core box with SMT: 128 logical CPUs
    sharers = [0, 1, 2, 3]                                     # four GPUs hang off this node
    slices = [rank_cpu_slice(node, sharers, r) for r in sharers]
    assert all(len(sl) == 32 for sl in slices)
    flat = [c for sl in slices for c in sl]
    assert len(set(flat)) == 128 and sorted(flat) == node      # disjoint, complete
    assert rank_cpu_slice(node, sharers, 2, per_rank=8) == slices[2][:8]
    assert rank_cpu_slice(node, [5], 5) == node                # alone on its node: all of it
    assert rank_cpu_slice([7], [0, 1], 1) == [7]               # fewer cores than ranks: never empty
    assert rank_cpu_slice(node, [0, 1], 3) == node             # not among the sharers: untouched


def test_observe_records_the_worst_value_and_asserts_the_bound(tmp_path, monkeypatch):
    """tests.util.observe: the recorder behind profiles/r05_parity_observed.json (asserts the frozen bound, keeps the worst value per key,
    merges with the record of earlier test processes of the same run)."""
    import json
    import pytest
    from tests import util
    monkeypatch.setattr(util, "_OBSERVED", {})
    util.observe("k/le", 0.2, 1.0)
    util.observe("k/le", 0.7, 1.0)
    util.observe("k/le", 0.4, 1.0)
    util.observe("k/ge", 9, 5, at_least=True)
    util.observe("k/ge", 6, 5, at_least=True)
    assert util._OBSERVED["k/le"] == {"worst": 0.7, "n": 3, "bound": 1.0, "kind": "<=", "shared": 1.0}
    assert util._OBSERVED["k/ge"] == {"worst": 6.0, "n": 2, "bound": 5.0, "kind": ">=", "shared": 5.0}
    with pytest.raises(AssertionError):
        util.observe("k/le", 1.5, 1.0)
    with pytest.raises(AssertionError):
        util.observe("k/ge", 4, 5, at_least=True)
    # a comparison with its own entry is held to the tighter of the two bounds (tests/parity_bounds.PER_COMPARISON)
    from tests import parity_bounds as PB
    monkeypatch.setitem(PB.PER_COMPARISON, "k/own", 0.5)
    monkeypatch.setitem(PB.PER_COMPARISON, "k/own_ge", 7)
    assert util.observe("k/own", 0.4, 1.0) == 0.4 and util._OBSERVED["k/own"]["bound"] == 0.5 and util._OBSERVED["k/own"]["shared"] == 1.0
    with pytest.raises(AssertionError):
        util.observe("k/own", 0.6, 1.0)
    util.observe("k/own", 0.05, 0.1)                              # the shared constant wins when IT is tighter
    with pytest.raises(AssertionError):
        util.observe("k/own", 0.2, 0.1)
    util.observe("k/own_ge", 7, 5, at_least=True)
    with pytest.raises(AssertionError):
        util.observe("k/own_ge", 6, 5, at_least=True)
    # flush merges into an existing file of the same run
    root = tmp_path / "repo"
    (root / "tests").mkdir(parents=True)
    (root / "gpurun_out").mkdir()
    (root / "gpurun_out" / "parity_observed.json").write_text(json.dumps({"k/le": {"worst": 0.9, "n": 2, "bound": 1.0, "kind": "<="}}))
    monkeypatch.setattr(util.os.path, "abspath", lambda p: str(root / "tests" / "util.py"))
    monkeypatch.setattr(util, "_OBSERVED", {"k/le": {"worst": 0.7, "n": 3, "bound": 1.0, "kind": "<="}, "k/new": {"worst": 1.0, "n": 1, "bound": 2.0, "kind": "<="}})
    util._flush_observed()
    got = json.loads((root / "gpurun_out" / "parity_observed.json").read_text())
    assert got["k/le"]["worst"] == 0.9 and got["k/le"]["n"] == 5 and got["k/new"]["n"] == 1


def test_bench_child_point_reads_the_last_json_line_and_survives_failures(tmp_path, monkeypatch):
    """bench.child_point: the operating points a default run measures in child processes (`single_stream`, the latency-first frontier
    point) - the child's LAST JSON line is the result; a child that crashes or prints nothing yields None and the parent's line still prints."""
    import subprocess
    import bench
    calls = []

    class R:
        def __init__(self, out):
            self.stdout, self.stderr, self.returncode = out, "", 0

    def fake_run(cmd, **kw):
        calls.append(cmd)
        return R('warming up\n{"value": 1.0}\nnoise\n{"value": 2.5, "p50_ttft_ms": 40.0}\n')
    monkeypatch.setattr(subprocess, "run", fake_run)
    got = bench.child_point(["--batch", "1"])
    assert got == {"value": 2.5, "p50_ttft_ms": 40.0}
    assert calls[0][2:4] == ["--batch", "1"] and {"--no-cpu-baseline", "--no-single-stream", "--no-latency-point", "--no-sequential-point"} <= set(calls[0])
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: R("no json here\n"))
    assert bench.child_point(["--batch", "1"]) is None

    def boom(cmd, **kw):
        raise subprocess.TimeoutExpired(cmd, 1)
    monkeypatch.setattr(subprocess, "run", boom)
    assert bench.child_point(["--batch", "1"]) is None
    assert bench.masked_steps_per_chunk(240.0, 36.7, 8) == 7 and bench.masked_steps_per_chunk(10.0, 36.7, 8) == 1 and bench.masked_steps_per_chunk(900.0, 36.7, 8) == 8


def test_bench_gate_positions_bound_the_run_ahead():
    """bench.gate_positions: the host submits a front end `margin` decode steps before the point where it may start - inside the chunk's
    leading steps when that point is deep enough, otherwise in the previous chunk's tail; never outside [0, n]."""
    import bench
    # cfg2: chunks of 8 steps, 7 masked -> the front end may start after 1 leading step
    assert bench.gate_positions(8, 1, 1, 0) == (1, None)          # margin 0: at the start point itself
    assert bench.gate_positions(8, 1, 1, 1) == (0, None)          # one step earlier = right behind the commit
    assert bench.gate_positions(8, 1, 1, 2) == (None, 7)          # two steps: one step before the end of the previous chunk
    assert bench.gate_positions(8, 0, 0, 0) == (0, None)          # no leading step: behind the commit
    assert bench.gate_positions(8, 0, 0, 1) == (None, 7)
    # cfg5: 170-step chunks, the front end starts after 150 leading steps
    assert bench.gate_positions(170, 150, 150, 0) == (150, None)
    assert bench.gate_positions(170, 150, 150, 3) == (147, None)
    for n in range(1, 12):
        for lead in range(0, n + 1):
            for lead_next in range(0, n + 1):
                for margin in range(0, 14):
                    pi, pt = bench.gate_positions(n, lead, lead_next, margin)
                    assert pi is None or 0 <= pi <= lead
                    assert pt is None or 0 <= pt <= n
                    assert (pi is None) == (lead < margin) and (pt is None) == (lead_next >= margin)
