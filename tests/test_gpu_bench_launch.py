"""GPU: bench.py in the driver's launch modes (tiny model dims: plumbing, not the metric).

`python bench.py --gpus 2` with NO torchrun environment must launch its own two ranks (self_launch) and report n_gpus 2.

The two-rank case runs `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2` exactly as the
driver does, with AURORA_DIST_BACKEND=gloo so that both ranks can share the single GPU of the test box (collectives
carry CPU tensors instead of going through RCCL).  It guards the multi-rank control flow: barriers, the per-step result
gather, max-over-ranks timing, the rank-0-only instrumented pass (no collective may hide in it) and a clean exit."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = ["--tiny", "--batch", "4", "--num_frm", "2", "--max_new_tokens", "8", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_single_process_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + TINY, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["rccl_ranks"] == 1 and d["dist_backend"] == "none"
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["value"] > 0
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(d["roofline"])


@pytest.mark.parametrize("extra", [[], ["--pipeline"], ["--prefill-group", "2"]])
def test_two_ranks_torchrun_gloo(extra):
    env = dict(os.environ, AURORA_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2"] + TINY + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["clips_per_gpu_per_step"] == 4                       # weak scaling: per-GPU work fixed
    assert abs(d["value"] - 2 * 4 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]     # whole-job aggregate


def test_gpus_flag_self_launches_ranks_without_torchrun_env():
    """The driver may run `python bench.py --gpus N` as ONE plain process: bench.py then re-executes itself as N ranks
    (VERDICT r01 item 1).  gloo lets both ranks share this box's single GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["AURORA_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + TINY, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["dist_backend"] == "gloo" and d["value"] > 0
    assert d["config"]["clips_per_gpu_per_step"] == 4


def test_gpus_flag_must_agree_with_the_launcher():
    env = dict(os.environ, AURORA_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + TINY, cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "must agree" in (r.stderr + r.stdout)


def test_default_mode_is_steady_state_continuous_batching():
    """With >= 2 prefill groups the timed region is the steady-state continuous-batching loop (groups of --prefill-group clips
    collected and re-filled between decode chunks); its captions are verified inside bench.py against a batch-mode step, whose
    timing is reported beside it."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--prefill-group", "2"] + TINY, cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["config"]["mode"].startswith("continuous batching") and d["value"] > 0 and d["p50_ttft_ms"] > 0
    assert d["batch_mode"]["captions_per_s"] > 0 and d["batch_mode"]["p50_ttft_ms"] > 0
    assert abs(d["value"] - 4 * 2 / (d["ms_per_step"] * 2 / 1e3)) < 1e-6 * d["value"]
    # default: the front ends overlap the decode on CU-masked streams (ids verified inside bench.py against the batch-mode step),
    # with one cycle of the sequential schedule reported beside it
    assert "own stream" in d["config"]["mode"] and d["sequential_schedule"]["captions_per_s"] > 0
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--prefill-group", "2", "--overlap", "0"] + TINY, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d0 = _json_line(r.stdout)
    assert d0["config"]["mode"].startswith("continuous batching") and "own stream" not in d0["config"]["mode"] and "sequential_schedule" not in d0
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--prefill-group", "2", "--batch-mode"] + TINY, cwd=ROOT,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and _json_line(r.stdout)["config"]["mode"].startswith("batch")


def test_two_ranks_over_rccl_when_the_box_has_two_gpus():
    """The `nccl` (= RCCL) branch of bench.py and aurora_amd/parallel.py: one rank per GPU, ids gathered with a device-side
    all_gather over xGMI.  Needs two GPUs on the node (the builder's test box has one: skipped there; the driver's multi-GPU
    node runs it).  Also checks the self-diagnosing fields of the line: every rank's own ms_per_step and the gather latency."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs on this node for an RCCL run")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "AURORA_DIST_BACKEND")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + TINY, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["dist_backend"] == "nccl" and d["rccl_ranks"] == 2 and d["value"] > 0
    assert len(d["ms_per_step_per_rank"]) == 2 and max(d["ms_per_step_per_rank"]) == pytest.approx(d["ms_per_step"])
    assert d["ids_all_gather_us"] > 0


@pytest.mark.parametrize("extra", [[], ["--batch-mode"]])
def test_one_rank_over_rccl_runs_every_collective_of_the_multi_rank_path(extra):
    """The RCCL branch on a ONE-GPU box: `torchrun --nproc-per-node 1 bench.py --gpus 1` with AURORA_DIST_FORCE=1 opens the `nccl`
    process group on the device (`device_id=`), and every collective of the N-rank path really runs through RCCL with device
    tensors - the GPU-ordinal probe, the barriers around the timed steps, the per-cycle ids all_gather
    (`parallel.gather_results`), the per-rank clocks and the gather latency.  (What a 2-GPU node adds - xGMI transport between
    ranks - is the test above.)  The ids are the ones the same run produces without a process group."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "AURORA_DIST_BACKEND")}
    env.update(AURORA_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1"] + TINY + extra
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 1 and d["dist_backend"] == "nccl" and d["rccl_ranks"] == 1 and d["value"] > 0
    assert len(d["ms_per_step_per_rank"]) == 1 and d["ms_per_step_per_rank"][0] == pytest.approx(d["ms_per_step"])
    assert d["ids_all_gather_us"] > 0
    env.pop("AURORA_DIST_FORCE")
    r0 = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + TINY + extra, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r0.returncode == 0, r0.stderr[-3000:]
    d0 = _json_line(r0.stdout)
    assert d0["dist_backend"] == "none" and d0["ids_checksum_rank0"] == d["ids_checksum_rank0"]


def test_config_presets_and_self_diagnosing_fields():
    """`--config cfgN` names a BASELINE.json config; explicit flags still win; the two-rank line carries per-rank times and the
    gather latency; the overlapped schedule reports a host-observed TTFT beside the device-event one."""
    env = dict(os.environ, AURORA_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--config", "cfg4", "--prefill-group", "2"] + TINY
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["config"]["preset"] == "cfg4" and d["config"]["clips_per_gpu_per_step"] == 4      # --batch 4 given explicitly: it wins over the preset's 8
    assert len(d["ms_per_step_per_rank"]) == 2 and max(d["ms_per_step_per_rank"]) == pytest.approx(d["ms_per_step"])
    assert d["ids_all_gather_us"] > 0
    assert d["p50_ttft_host_ms"] is not None and d["p50_ttft_host_ms"] > 0
    assert d["ttft_host_ms_single_clip"] >= d["ttft_ms_single_clip"] * 0.9


def test_eight_rank_dry_run_of_the_configs3_command():
    """`python bench.py --gpus 8 --config cfg4` IS BASELINE configs[3] (64 clips sharded 8-way).  No 8-GPU node has been available, so the
    literal command runs here with tiny model dims and gloo: bench.py launches its own 8 ranks (8 processes sharing this box's GPU),
    they rendezvous on 127.0.0.1, shard the 64 clips round-robin (clip i -> rank i % 8, lmms_eval/utils.py:675-681), time their steps
    on their own clocks, all_gather the ids every cycle and rank 0 merges them back into clip order (evaluator.py:519-546).
    The merged order is checked against an independent run: a single process that takes rank 3's shard (AURORA_BENCH_SHARD=3/8) must
    produce exactly the ids the 8-rank job reports for clips 3, 11, 19, ..."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["AURORA_DIST_BACKEND"] = "gloo"
    common = ["--config", "cfg4", "--tiny", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-power"]
    for attempt in range(2):             # one retry: 8 fresh processes + a rendezvous right behind the previous tests' torchrun jobs flaked once in 5 runs
        r = subprocess.run([sys.executable, "bench.py", "--gpus", "8"] + common, cwd=ROOT, capture_output=True, text=True, timeout=1800, env=env)
        if r.returncode == 0:
            break
    if r.returncode != 0:                                                  # keep the whole log: a rank's traceback sits far above torchrun's summary
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "eight_rank_dry_run.err"), "w") as f:
            f.write(r.stdout + "\n==== stderr\n" + r.stderr)
    tb = [ln for ln in r.stderr.splitlines() if "Error" in ln or "error" in ln or "Traceback" in ln or "assert" in ln][:12]
    assert r.returncode == 0, (tb, r.stderr[-1500:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["dist_backend"] == "gloo" and d["config"]["preset"] == "cfg4" and d["config"]["clips_per_gpu_per_step"] == 8
    assert d["config"]["frames"] == 8 and d["config"]["max_new_tokens"] == 256 and d["config"]["token_kept_ratio"] == 0.3
    assert len(d["ms_per_step_per_rank"]) == 8 and max(d["ms_per_step_per_rank"]) == pytest.approx(d["ms_per_step"])
    assert d["value"] == pytest.approx(64 / (d["ms_per_step"] / 1e3), rel=1e-6)              # whole-job aggregate: 64 clips per step
    assert len(d["ids_crc_per_clip"]) == 64 and d["clips_of_rank0"][:4] == [0, 8, 16, 24]
    env1 = dict(env, AURORA_BENCH_SHARD="3/8")
    env1.pop("AURORA_DIST_BACKEND")
    r1 = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env1)
    assert r1.returncode == 0, (r1.stdout[-2000:], r1.stderr[-3000:])
    d1 = _json_line(r1.stdout)
    assert d1["clips_of_rank0"][:4] == [3, 11, 19, 27]
    assert d1["ids_crc_per_clip"] == d["ids_crc_per_clip"][3::8]


def test_eight_full_rate_enqueue_loops_share_one_host():
    """on an 8-rank node the risk is the HOST - every rank's Python thread issues ~1.6e4 kernel launches and ~2.6e2
    hipGraph replays per cycle of 128 captions.  Rehearsal without the node: `--tiny-deep` keeps the REAL layer counts (31 ViT + 32 Llama
    layers, 128 slots, groups of 4, 256 tokens: the same launches per cycle as the metric's run) at tiny widths, so that eight such loops
    run at full rate on this box's host (8 processes sharing its GPU, gloo).  The slowest rank's enqueue thread must need less than half
    of the real cycle (9.0 s on one MI355X) of CPU per cycle - otherwise eight ranks on one host would be enqueue-bound."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["AURORA_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, "bench.py", "--gpus", "8", "--tiny-deep", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-power", "--no-instrument"]
    for attempt in range(2):
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=2400, env=env)
        if r.returncode == 0:
            break
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["config"]["clips_per_gpu_per_step"] == 128 and d["config"]["prefill_group"] == 4
    per = d["host"]["per_rank"]
    assert len(per) == 8
    worst_cpu = max(p["enqueue_thread_cpu_s_per_cycle"] for p in per)
    worst_wall = max(p["enqueue_s_per_cycle"] for p in per)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "eight_rank_enqueue_rehearsal.json"), "w") as f:
        json.dump({"host": d["host"], "ms_per_step": d["ms_per_step"], "ms_per_step_per_rank": d["ms_per_step_per_rank"],
                   "cores": os.cpu_count(), "note": "bench.py --gpus 8 --tiny-deep (gloo, 8 processes on one GPU): real launch counts, tiny kernels"}, f, indent=1)
    print(f"\n8 ranks: slowest enqueue thread {worst_cpu:.2f} CPU-s per cycle ({worst_wall:.2f} s wall incl. queue back-pressure of the shared GPU)")
    # Round 6: a front end is one graph replay and the enqueue thread sleeps between submissions, so a rank's enqueue thread needs a
    # fraction of a second per cycle whatever the GPU does.  The bound is a property of the code only on a host with a core per rank to
    # spare: elsewhere the figure is recorded, not asserted.
    if (os.cpu_count() or 1) >= 32:
        assert worst_cpu < 0.5 * 9.0, per
