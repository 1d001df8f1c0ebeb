#!/usr/bin/env python3
"""AuroraCap-7B inference-path benchmark on MI355X (the metric BASELINE.json names).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic clips per GPU:
  ViT-H/14-378 + per-layer ToMe over all frames -> projector + splice -> Llama-7B prefill (groups of equal-length
  clips per pass) -> batched greedy decode to exactly max_new_tokens (EOS disabled: random weights would make the
  length arbitrary).
Workload = BASELINE.json configs[1]: AuroraCap-7B-VID, 8 frames, token_kept_ratio 0.3, 256 new tokens, 1 GPU.
`value` = captions/sec of the whole job (all ranks), inputs resident in HBM when the timed region starts.

Default schedule: steady-state continuous batching over `--batch` KV slots (128).  All slots decode all the time; they form
batch / G groups (G = `--prefill-group`, 4) whose captions end group by group; the front end of a group's next clips (ViT + ToMe +
splice + staged prefill into spare KV sequences) runs on its own stream restricted to 16 CUs of every XCD WHILE the slots decode
(decode is HBM-bound, the front end MFMA-bound), and is committed into the group's slots at its boundary.  One step = one cycle of
max_new_tokens - 1 decode steps = `batch` captions completed + `batch` front ends.  Every cycle's ids are checked against one
batch-mode step.  `--overlap 0` runs the front ends between decode chunks on one stream (reported beside the line as
`sequential_schedule`), `--batch-mode` times independent batches (`batch_mode`), `--pipeline` is round 1's two-bank variant.

For N > 1 clips shard across ranks (one process per GPU, no data-path collective); the only collective is the RCCL
all_gather of the generated ids at the end of each step (the reference's gather_object, evaluator.py:519-546).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


# BASELINE.json `configs` by name.  cfg2 is the metric's workload (the default); cfg3 / cfg5 are its merge-dominated and long-KV
# single-GPU cases (KV slots sized so that the pool fits 288 GB: 52 resp. 109 pages of 32 MiB per slot); cfg4 is cfg2 at 8 clips per GPU:
# `--gpus 8 --config cfg4` is literally "batch of 64 clips sharded 8-way".  configs[0] is the reference's CPU-runnable case: it is
# timed on the CPU oracle (`cpu_baseline.configs0`), on the GPU it is `--num_frm 1 --token_kept_ratio 1.0 --max_new_tokens 32 --batch 1`.
CONFIGS = {
    "cfg2": dict(what="configs[1]: 8 frames, ratio 0.3, 256 tokens, 128 slots per GPU", index=1,
                 set=dict(num_frm=8, token_kept_ratio=0.3, max_new_tokens=256, batch=128)),
    "cfg3": dict(what="configs[2]: 16 frames, ratio 0.2, 512 tokens (merge-dominated), 96 slots", index=2,
                 set=dict(num_frm=16, token_kept_ratio=0.2, max_new_tokens=512, batch=96)),
    "cfg4": dict(what="configs[3]: cfg2 at 8 clips per GPU - with --gpus 8 the batch of 64 clips sharded 8-way", index=3,
                 set=dict(num_frm=8, token_kept_ratio=0.3, max_new_tokens=256, batch=8)),
    "cfg5": dict(what="configs[4]: 8 frames, ratio 0.8 (OCR regime), 2048 tokens, long-KV hipGraph decode, 48 slots", index=4,
                 set=dict(num_frm=8, token_kept_ratio=0.8, max_new_tokens=2048, batch=48)),
}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=128,
                   help="clips per GPU per step = decode slots (<= 128).  128: 11.30 captions/s vs 11.07 at 96 and 10.74 at 64 on one MI355X "
                        "(the 13.2 GB of weights stream once per decode step however many sequences it serves); the KV pool is then 156 GB")
    p.add_argument("--num_frm", type=int, default=8)
    p.add_argument("--token_kept_ratio", type=float, default=0.3)
    p.add_argument("--max_new_tokens", type=int, default=256)
    p.add_argument("--prefill-group", type=int, default=0,
                   help="clips per front-end pass = per prefill pass (equal-length prompts); 0 = auto: 4 with --overlap (13.30 captions/s, p50 "
                        "TTFT 0.30 s; 8: 13.24 / 0.61 s; 2: 12.9 / 0.15 s on one MI355X), 8 otherwise")
    p.add_argument("--vit-chunk", type=int, default=16,
                   help="clips per ViT pass (0 = the whole batch at once); chunks interleave ViT with the prefill groups so the "
                        "first tokens of the early groups come sooner (measured at B=64: 16 -> p50 TTFT -11 %%, captions/s -0.2 %%)")
    p.add_argument("--pipeline", action="store_true",
                   help="overlap batch i's decode with batch i+1's ViT + prefill on two streams / two KV banks "
                        "(measured on MI355X: +2-3 %% captions/s, +37 %% p50 TTFT - off by default)")
    p.add_argument("--overlap", type=int, default=-1, choices=[-1, 0, 1],
                   help="continuous mode: 1 = a group's next front end (ViT + splice + staged prefill into spare KV sequences) runs on its own "
                        "CU-masked stream WHILE all slots decode, and is committed at the group's boundary; 0 = front ends between decode chunks "
                        "(11.4 captions/s against 13.6-13.7 with 1 on one MI355X); -1 = auto (1 whenever the continuous mode applies)")
    p.add_argument("--overlap-steps", type=int, default=-1,
                   help="--overlap 1: decode steps per chunk that run on the complementary CU mask (the rest of the chunk runs unmasked); "
                        "-1 = as many as the front end needs (measured in the warm-up cycle), at most a chunk's whole steps")
    p.add_argument("--half-grid", type=int, default=1, choices=[0, 1], nargs="?", const=1,
                   help="--overlap 1: masked decode steps launch the QKV / gate-up projections with half as many workgroups and twice the "
                        "tiles each (decode_half_grid: x crosses a CU's LDS once instead of twice; 41.5 vs 60.8 us and 48.0 vs 62.7 us on 16 "
                        "CUs per XCD at 10-12 %% less energy per launch, bitwise the same tokens).  Default 1 since round 3: 14.00-14.02 "
                        "against 13.64-13.85 captions/s, three alternating pairs on one box (round 2: 13.66 vs 13.72); 0 = full grid")
    p.add_argument("--sync-chunks", action="store_true",
                   help="--overlap 1: synchronise the decode stream after every chunk (profiling aid: rocprofv3 --kernel-trace needs the "
                        "queue of pending hipGraph launches kept short; costs a host round trip per chunk)")
    p.add_argument("--front-cus", type=int, default=0,
                   help="--overlap 1 / --pipeline: run the front-end stream on this many CUs of every XCD (hipExtStreamCreateWithCUMask; "
                        "--overlap 1 defaults to 16)")
    p.add_argument("--decode-cus", type=int, default=0,
                   help="with --pipeline: run the decode stream on this many CUs of every XCD (taken from the other end)")
    p.add_argument("--gemm-cus", type=int, default=0,
                   help="with --pipeline: run the 256x256 GEMM persistently on at most this many workgroups (= CUs), leaving the "
                        "other CUs to the concurrently decoding stream; 0 = one workgroup per tile")
    p.add_argument("--gemm-tail-split", type=int, default=-1, help="A/B: 0 = one 256x256 launch per GEMM, 1 = idle last rounds go to the 128x128 kernel")
    p.add_argument("--gemm-mode", type=int, default=-1, help="override the GEMM kernel choice (0: 128x128 only, 1: auto, 2: force 256x256)")
    p.add_argument("--prune-last", type=int, default=-1, help="A/B: 0 = the last prefill layer computes every row (as HF does), 1 (engine default) = K / V for every "
                   "row, the rest for each sequence's last 128 rows only (bitwise the same outputs)")
    p.add_argument("--fused-reduce", type=int, default=-1, help="A/B: 0 (engine default) = the split-K residual projections (o, down) are followed by a reduce launch, 1 = "
                   "the reduce runs inside the projection kernel (bitwise the same outputs; measured no faster)")
    p.add_argument("--gemm-nt-out", type=int, default=-1, help="A/B: GEMM output stores 0 = default cache policy, 1 = non-temporal, -1 (engine default) = non-temporal "
                   "for outputs larger than the L2s together")
    p.add_argument("--dec-attn-pps", type=int, default=0, help="A/B: KV pages per decode-attention split (0 = the engine's choice: ~512 waves on the GPU)")
    p.add_argument("--sync-front", action="store_true", help="batch mode: synchronise after every prefill group (profiling aid: keeps the queue of pending "
                                                            "launches short - rocprofv3's counter mode crashed with ~11 k launches queued ahead of the GPU)")
    p.add_argument("--no-power", action="store_true", help="do not sample rocm-smi during the timed steps (the sampler forks a subprocess every 1.5 s; "
                                                          "skipped automatically under rocprofv3, whose launch interception does not survive the fork)")
    p.add_argument("--no-graph", action="store_true")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-instrument", action="store_true", help="skip the event-bracketed roofline pass")
    p.add_argument("--decode-chunk", type=int, default=0,
                   help="synchronise every N decode steps (rocprofv3 --kernel-trace segfaults with > ~150 hipGraph launches queued)")
    p.add_argument("--batch-mode", action="store_true",
                   help="time K independent batches (front end of all B clips, then B-wide decode) instead of the default steady-state "
                        "continuous batching (same work per step, first tokens ~5x later)")
    p.add_argument("--no-latency-point", action="store_true", help="skip the latency-first point of the frontier (prefill groups of 2, one timed cycle in a child "
                   "process after the timed steps of the default cfg2 run)")
    p.add_argument("--tiny", action="store_true", help="tiny model dims (plumbing check only; result is not the metric)")
    p.add_argument("--tiny-deep", action="store_true", help="host-side rehearsal: tiny widths but the REAL layer counts (31 ViT + 32 Llama layers), so that a cycle "
                   "enqueues the real number of launches with kernels of a few microseconds - what the host of an 8-rank node has to sustain (`host` in the line)")
    p.add_argument("--no-sequential-point", action="store_true", help="skip the one cycle of the sequential schedule that a default run times beside the line "
                   "(the child processes of the frontier points pass it)")
    p.add_argument("--no-single-stream", action="store_true", help="skip the one-slot operating point (`single_stream`: the reference's own use, inference.py:89-96 / "
                   "EVAL.md 'batch size 1 only'), which the default cfg2 run measures in a child process after the timed steps")
    p.add_argument("--frontier", action="store_true", help="also measure the (captions/s, p50 TTFT) points of prefill groups 1 / 2 / 8 in child processes (minutes); "
                   "without it the line's `frontier` holds the three schedules this run times anyway")
    p.add_argument("--front-graph", type=int, default=1, choices=[0, 1],
                   help="overlapped schedule: 1 (default) = a group's front end (ViT + ToMe + projector / splice + staged prefill) is ONE captured hipGraph "
                        "(engine.FrontEndGraph: inputs copied into static buffers, one replay per group); 0 = ~500 eager launches per group.  Same ids")
    p.add_argument("--ttft-gate-steps", type=int, default=0,
                   help="overlapped schedule: bounded run-ahead - the host submits a group's front end only when the device is within this many decode "
                        "steps of the point where that front end may start (0, the default = at that point: the front end starts a host round trip - ~0.5 ms - late, "
                        "inside the slack it has before its boundary: 14.13-14.19 captions/s with 0 and with 1, TTFT 256 against 283 ms), so that "
                        "the host-observed submit -> first-token time stays close to the device interval; -1 = unbounded (rounds 2-5: the enqueue "
                        "thread ran > 1 s ahead)")
    p.add_argument("--no-stamps", action="store_true", help="do not stamp the decode attention of one layer inside the captured step (roofline.frac_in_timed_loop)")
    p.add_argument("--ttft-delay-steps", type=int, default=-1, help="overlapped schedule: the front end of a group starts this many decode steps after the previous "
                   "boundary instead of at it (-1 = calibrated in the warm-up cycle so that it finishes just before its own boundary: no commit wait)")
    p.add_argument("--config", choices=sorted(CONFIGS), default=None,
                   help="a BASELINE.json config by name (sets --num_frm / --token_kept_ratio / --max_new_tokens / --batch; flags given "
                        "explicitly on the command line still win): " + "; ".join(f"{k} = {v['what']}" for k, v in sorted(CONFIGS.items())))
    args = p.parse_args()
    args.config_name = args.config or "cfg2"
    if args.config:
        given = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
        for k, val in CONFIGS[args.config]["set"].items():
            if "--" + k not in given and "--" + k.replace("_", "-") not in given:
                setattr(args, k, val)
    return args


def cpu_baseline(cfg, args, n_kept):
    """SURVEY 8d "CPU baseline beside it": the oracle (oracle/aurora_oracle.py, the PyTorch-CPU fp32 port of the reference
    path; the reference's own Python cannot run on the GPU box) timed on this box's host cores on
      * configs[1] (the bench workload): ViT + per-layer ToMe on ALL frames, projector + splice and the Llama prefill IN
        FULL (32 layers), then 16 greedy decode tokens, extrapolated linearly to max_new_tokens (stated), and
      * configs[0] (1 frame, token_kept_ratio 1.0, 32 tokens): in full.
    Runs after the timed region on rank 0 only.  A host with too little free memory for the fp32 weights (27 GB) falls back to a
    bounded layer sample and says so."""
    import platform
    from aurora_amd import synthetic as S
    from oracle import aurora_oracle as O
    torch.set_grad_enabled(False)
    cores = torch.get_num_threads()
    cpu_model = platform.processor() or "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                cpu_model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    v, l = cfg["vit"], cfg["llm"]
    f32 = lambda d: {k: ([f32(x) for x in val] if isinstance(val, list) else val.float().cpu()) for k, val in d.items()}
    nl_full = l["num_hidden_layers"]
    try:
        import psutil
        free_gb = psutil.virtual_memory().available / 2**30
    except Exception:
        free_gb = 0.0
    need_gb = 4.0 * (nl_full * (4 * l["hidden_size"] ** 2 + 3 * l["hidden_size"] * l["intermediate_size"]) + 2 * l["vocab_size"] * l["hidden_size"]) / 2**30
    nl = nl_full if free_gb > 1.5 * need_gb + 16 else 2
    wdev = "cuda" if torch.cuda.is_available() else "cpu"               # same generator stream as the engine's weights
    vw = f32(S.vit_weights(v, device=wdev))
    pw = f32(S.projector_weights(v["hidden_size"], l["hidden_size"], device=wdev))
    lw = f32(S.llm_weights(l, device=wdev, num_layers=nl))
    if wdev == "cuda":
        torch.cuda.empty_cache()
    sub = dict(l, num_hidden_layers=nl)
    scale = nl_full / nl

    def caption(frames_n, ratio, n_dec, clip):
        """-> (seconds: vision, project+splice, prefill, per decoded token [measured over n_dec tokens]), prefix length."""
        px = S.frames(frames_n, clip, v["image_size"], device="cpu").float()
        ids = S.prompt_ids(frames_n, clip, 30, l["vocab_size"])
        t0 = time.perf_counter()
        feats = O.vit_features(px, vw, v, ratio)
        t1 = time.perf_counter()
        f, n, dv = feats.shape
        vis = O.projector(feats.reshape(1, f * n, dv), pw).reshape(f, n, -1)
        emb = O.splice(torch.tensor(ids), lw["embed_tokens.weight"], vis)
        t2 = time.perf_counter()
        h, kv = O.llama_forward(emb, lw, sub, None, 0)
        logits = torch.nn.functional.linear(h[-1:], lw["lm_head.weight"])
        nxt = int(logits.argmax())                                     # first token: end of TTFT
        t3 = time.perf_counter()
        pos = emb.shape[0]
        for _ in range(n_dec):
            h, kv = O.llama_forward(lw["embed_tokens.weight"][nxt][None], lw, sub, kv, pos)
            nxt = int(torch.nn.functional.linear(h[-1:], lw["lm_head.weight"]).argmax())
            pos += 1
        t4 = time.perf_counter()
        t_lm = 0.0
        if nl != nl_full:                                              # layer sample: the lm_head does not scale with depth
            t5 = time.perf_counter()
            torch.nn.functional.linear(h[-1:], lw["lm_head.weight"])
            t_lm = time.perf_counter() - t5
        per_tok = ((t4 - t3) / max(n_dec, 1) - t_lm) * scale + t_lm
        return dict(vision_s=t1 - t0, project_s=t2 - t1, prefill_s=(t3 - t2 - t_lm) * scale + t_lm, s_per_token=per_tok, prefix=emb.shape[0])

    ndec = 16
    c2 = caption(args.num_frm, args.token_kept_ratio, ndec, 0)
    assert c2["prefix"] == 30 + args.num_frm * n_kept
    ttft2 = c2["vision_s"] + c2["project_s"] + c2["prefill_s"]
    total2 = ttft2 + c2["s_per_token"] * (args.max_new_tokens - 1)
    c1 = caption(1, 1.0, 31, 1)                                         # configs[0]: 1 frame, ratio 1.0, 32 tokens, in full
    ttft1 = c1["vision_s"] + c1["project_s"] + c1["prefill_s"]
    total1 = ttft1 + c1["s_per_token"] * 31
    depth = "all %d layers" % nl_full if nl == nl_full else "%d of %d Llama layers (host has %.0f GB free, fp32 weights need %.0f GB), scaled by depth" % (nl, nl_full, free_gb, need_gb)
    return dict(value=1.0 / total2, unit="captions/s", cores=cores, kind="port", cpu_model=cpu_model,
                sample=(f"oracle (PyTorch-CPU fp32 port of the reference path, {cores} threads, {cpu_model}): configs[1] one clip - ViT + ToMe on all "
                        f"{args.num_frm} frames {c2['vision_s']:.1f}s, projector + splice {c2['project_s']:.2f}s, Llama prefill of {c2['prefix']} tokens "
                        f"{c2['prefill_s']:.1f}s, {ndec} decode tokens at {c2['s_per_token']:.3f}s/token, {depth}; decode extrapolated linearly from "
                        f"{ndec} to {args.max_new_tokens - 1} steps"),
                ttft_s=ttft2, s_per_caption=total2,
                configs0={"workload": "1 frame, token_kept_ratio 1.0, greedy 32 tokens (BASELINE configs[0]), timed in full",
                          "s_per_caption": total1, "ttft_s": ttft1, "captions_per_s": 1.0 / total1, "prefix": c1["prefix"],
                          "vision_s": c1["vision_s"], "prefill_s": c1["prefill_s"], "s_per_token": c1["s_per_token"]})


class PowerSampler:
    """Socket power and shader clock of this rank's GPU during the timed region (rank 0 only).  Round 3 found the overlapped schedule
    running AT the 1400 W socket cap with sclk ~2.1 GHz: the line carries the evidence.
    Source: the amdgpu hwmon files of the device (`power1_average` / `power1_input` in microwatts, `freq1_input` in Hz) read in
    place - no subprocess in the timed region (the rocm-smi fork every 1.5 s sat on the process that enqueues the launches).
    Only where sysfs does not expose them does it fall back to forking `rocm-smi`; `how` in the result says which one ran.  Any
    failure just leaves the fields out; `--no-power` switches it off and the line says so (`power: null`, `power_sampling: "off"`)."""

    def __init__(self, card: int):
        import threading
        self.card, self.samples, self._stop = card, [], threading.Event()
        self._hw = self._find_hwmon(card)
        self.how = "off"
        self._t = threading.Thread(target=self._run, daemon=True)

    @staticmethod
    def _find_hwmon(card: int):
        import glob
        try:
            pr = torch.cuda.get_device_properties(card)
            bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            cands = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*")
        except Exception:                                          # noqa: BLE001
            cands = []
        if not cands and torch.cuda.device_count() == 1:
            cands = [h for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")]
        for h in cands:
            pw = next((os.path.join(h, n) for n in ("power1_average", "power1_input") if os.path.exists(os.path.join(h, n))), None)
            fq = os.path.join(h, "freq1_input")
            if pw and os.path.exists(fq):
                return pw, fq
        return None

    def _read_sysfs(self):
        pw, fq = self._hw
        return float(open(pw).read()) / 1e6, float(open(fq).read()) / 1e6          # W, MHz

    def _read_smi(self):
        import subprocess
        out = subprocess.run(["rocm-smi", "-d", str(self.card), "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
        pw = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k), None)
        sclk = next((float(str(v).strip("()").lower().replace("mhz", "")) for k, v in card.items() if k.lower().startswith("sclk clock speed")), None)
        return pw, sclk

    def _run(self):
        read, period = (self._read_sysfs, 0.5) if self._hw else (self._read_smi, 1.5)
        self.how = ("amdgpu hwmon sysfs (power1_average / freq1_input) read every ~0.5 s by a helper thread: no subprocess"
                    if self._hw else "rocm-smi --showpower --showclocks forked every ~1.5 s (sysfs hwmon files not found)")
        while not self._stop.is_set():
            try:
                pw, sclk = read()
                if pw is not None and sclk is not None and sclk > 200:
                    self.samples.append((pw, sclk))
            except Exception:                                      # noqa: BLE001
                if read == self._read_sysfs:                       # sysfs unreadable after all: one fall-back, then give up
                    read, period, self.how = self._read_smi, 1.5, "rocm-smi --showpower --showclocks forked every ~1.5 s (sysfs read failed)"
                    continue
                return
            self._stop.wait(period)

    def start(self):
        self._t.start()
        return self

    def stop(self):
        self._stop.set()
        self._t.join(timeout=30)
        if not self.samples:
            return None
        pw, ck = sorted(p for p, _ in self.samples), sorted(c for _, c in self.samples)
        return {"socket_power_w_p50": pw[len(pw) // 2], "socket_power_w_max": pw[-1], "sclk_mhz_p50": ck[len(ck) // 2], "sclk_mhz_min": ck[0],
                "samples": len(pw), "how": self.how + ", during the timed steps"}


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` with no torchrun environment: re-exec this command line as N ranks of ONE node under
    torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1), exactly the launch the driver uses.  Rank 0's
    JSON line passes through on stdout.  Mirrors the reference harness's one-process-per-GPU `accelerate launch`
    (lmms_eval/utils.py:675-681 shard, evaluator.py:519-546 gather)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def child_point(flags, timeout=1200):
    """One more operating point of this benchmark in a child process (its own engine); -> the child's JSON line, or None."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__)] + list(flags) + ["--no-cpu-baseline", "--no-power", "--no-single-stream", "--no-latency-point", "--no-sequential-point"]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        return json.loads(lines[-1]) if lines else None
    except Exception:                                              # noqa: BLE001 - the parent's line must still print
        return None


def thread_cpu_seconds():
    """{tid: (name, user + system CPU seconds)} of every thread of this process (/proc/self/task): who burns the host while the GPU works"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                st = open(f"/proc/self/task/{tid}/stat").read()
                name = st[st.index("(") + 1:st.rindex(")")]
                f = st[st.rindex(")") + 2:].split()
                out[int(tid)] = (name + (" (main)" if int(tid) == os.getpid() else ""), (int(f[11]) + int(f[12])) / tick)
            except (OSError, ValueError):
                pass
    except OSError:
        pass
    return out


def sleep_wait(ev, dt: float = 0.0005):
    """Wait for a HIP event WITHOUT spinning: hipEventSynchronize burns a core for the whole wait (also with hipEventBlockingSync on this
    runtime: measured 8.8 of 9.1 s of thread CPU time per cycle), and on an 8-rank host every rank has three such waiters."""
    while not ev.query():
        time.sleep(dt)


def gate_positions(n: int, lead: int, lead_next: int, margin: int):
    """Where a chunk of `n` decode steps records the two events of the bounded run-ahead (positions = steps of the chunk enqueued so far):
    `pos_in` - the host may submit THIS chunk's front end (it may start after `lead` steps) once the device is `margin` steps before that
    point, if that point lies inside this chunk (else None: the previous chunk's tail event is the gate); `pos_tail` - the event the NEXT
    chunk needs when its own start point is less than `margin` steps into it (else None)."""
    pos_in = lead - margin if lead - margin >= 0 else None
    pos_tail = max(n - (margin - lead_next), 0) if lead_next - margin < 0 else None
    return pos_in, pos_tail


def masked_steps_per_chunk(front_ms: float, step_ms: float, chunk_steps: int) -> int:
    """Decode steps of a chunk that run on the decode mask: as many as the front end lasts beside them (both measured in the warm-up
    cycle), at most the chunk's whole steps, at least one."""
    need = int(np.ceil(front_ms / max(step_ms, 1e-3)))
    return max(1, min(int(chunk_steps), need))


def overlap_pays(slots: int, prefill_len: int, new_tokens: int, llm: dict) -> bool:
    """Schedule rule of the default (`--overlap -1`) mode: front ends beside the decode only where the decode step is dominated by the
    K / V stream (aurora_amd.streams.overlap_pays at the cycle's mean context; cfg4's 8 slots keep their front ends out of the decode)."""
    from aurora_amd.streams import overlap_pays as pays
    return pays(slots, prefill_len + new_tokens / 2, llm)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; they must agree")
    # AURORA_DIST_BACKEND=gloo is a TEST hook: it lets `torchrun --nproc-per-node 2 bench.py --gpus 2 --tiny` exercise the
    # multi-rank control flow on a one-GPU box (ranks share the device, collectives carry CPU tensors).  Default: RCCL.
    backend = os.environ.get("AURORA_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    # AURORA_DIST_FORCE=1 is a TEST hook too: a ONE-rank launch still opens the process group and makes every collective of the multi-rank
    # path (the only way to run the RCCL branch on a one-GPU box: `torchrun --nproc-per-node 1 bench.py --gpus 1`)
    use_dist = world > 1 or os.environ.get("AURORA_DIST_FORCE") == "1"
    dist = None
    if use_dist:
        import torch.distributed as dist
        if backend == "nccl":
            if torch.cuda.device_count() < world:
                raise SystemExit(f"bench.py: {world} ranks need {world} GPUs on this node, found {torch.cuda.device_count()}")
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus and dist.get_rank() == rank, (dist.get_world_size(), args.gpus)
        # one real collective before any timing: every rank contributes its GPU ordinal, all must be distinct under RCCL
        probe = torch.tensor([local], dtype=torch.int32, device=(f"cuda:{local}" if backend == "nccl" else "cpu"))
        seen = [torch.empty_like(probe) for _ in range(world)]
        dist.all_gather(seen, probe)
        gpus_seen = sorted(int(t.item()) for t in seen)
        if backend == "nccl":
            assert gpus_seen == list(range(world)), f"ranks are not bound one per GPU: {gpus_seen}"
    from aurora_amd import parallel
    pinned = None
    if world > 1 and os.environ.get("AURORA_PIN", "1") != "0" and backend == "nccl":
        pinned = parallel.pin_rank_to_gpu_numa(local, world)       # disjoint NUMA-local core slices: eight enqueue loops share one host
    from aurora_amd import synthetic as S
    from aurora_amd.engine import AuroraCapEngine, _rup, tokens_at_layer, tome_r

    if args.tiny_deep:
        args.tiny = True
    if args.tiny:
        cfg = {"vit": dict(hidden_size=128, num_attention_heads=4, num_hidden_layers=32 if args.tiny_deep else 4, intermediate_size=256, patch_size=14,
                           image_size=112, hidden_act="quick_gelu", layer_norm_eps=1e-5),
               "llm": dict(hidden_size=256, num_attention_heads=4, num_hidden_layers=32 if args.tiny_deep else 2, intermediate_size=512, vocab_size=1024,
                           rms_norm_eps=1e-5, rope_theta=1e4, rope_factor=4.0)}
    else:
        cfg = S.AURORACAP_7B
    v, l = cfg["vit"], cfg["llm"]
    B, F, N = args.batch, args.num_frm, args.max_new_tokens
    pipe = args.pipeline
    dev = f"cuda:{local}"
    cdev = dev if backend == "nccl" else "cpu"                     # where collective payloads live
    weights = {"vit": S.vit_weights(v, device=dev), "projector": S.projector_weights(v["hidden_size"], l["hidden_size"], device=dev),
               "llm": S.llm_weights(l, device=dev)}
    t0tok = (v["image_size"] // v["patch_size"]) ** 2 + 1
    r = tome_r(v["image_size"], v["image_size"], v["patch_size"], args.token_kept_ratio, v["num_hidden_layers"])
    n_kept = tokens_at_layer(t0tok, r, v["num_hidden_layers"] - 1) - 1
    L0 = 30 + F * n_kept
    max_ctx = _rup(L0 + N, 64)
    want_overlap = args.overlap != 0 and not (args.batch_mode or pipe or args.decode_chunk > 0)
    if args.overlap < 0 and want_overlap and not args.tiny:        # (tiny dims are plumbing runs: they keep the default schedule)
        want_overlap = overlap_pays(B, L0, N, l)
    G = max(1, min(args.prefill_group if args.prefill_group > 0 else (4 if want_overlap else 8), B))
    VC = B if args.vit_chunk <= 0 else max(G, args.vit_chunk // G * G)       # clips per ViT pass: a multiple of G
    overlap = want_overlap and B % G == 0 and B // G >= 2 and N >= 2 * (B // G)         # = the continuous mode applies
    eng = AuroraCapEngine(cfg, weights, max_frames=max(VC, G) * F, max_batch=B, max_ctx=max_ctx, max_new_tokens=N,
                          use_graph=not args.no_graph, num_banks=2 if pipe else 1, device=dev, spare_slots=G if overlap else 0)
    del weights
    torch.cuda.empty_cache()
    if args.gemm_mode >= 0:
        eng.set_option("gemm_mode", args.gemm_mode)
    if args.gemm_cus > 0:
        eng.set_option("gemm_max_wgs", args.gemm_cus)
    if args.gemm_tail_split >= 0:
        eng.set_option("gemm_tail_split", args.gemm_tail_split)
    if args.prune_last >= 0:
        eng.set_option("prefill_prune_last", args.prune_last)
    if args.fused_reduce >= 0:
        eng.set_option("decode_fused_reduce", args.fused_reduce)
    if args.gemm_nt_out >= 0:
        eng.set_option("gemm_nt_out", args.gemm_nt_out)
    if args.dec_attn_pps > 0:
        eng.set_option("dec_attn_pps", args.dec_attn_pps)
    stamp_layer = -1 if args.no_stamps else l["num_hidden_layers"] // 2
    if stamp_layer >= 0:
        eng.set_option("decode_stamp_layer", stamp_layer)         # two one-thread launches around that layer's attention, part of the captured step
    stamps_timed = None

    # synthetic inputs, resident in HBM before the timed region.  Clip i of the job's world * B clips belongs to rank i % world - the
    # reference harness's round-robin `islice(docs, rank, None, world_size)` (lmms_eval/utils.py:675-681, aurora_amd.parallel.shard_clips).
    # AURORA_BENCH_SHARD="k/n" is a TEST hook: a single process takes the clips of rank k of an n-rank job (tests compare them with the
    # n-rank run's merged result clip by clip).
    shard_rank, shard_world = rank, world
    if os.environ.get("AURORA_BENCH_SHARD") and world == 1:
        shard_rank, shard_world = (int(x) for x in os.environ["AURORA_BENCH_SHARD"].split("/"))
    clip_ids = parallel.shard_clips(shard_world * B, shard_rank, shard_world)
    assert len(clip_ids) == B
    pixels = torch.cat([S.frames(F, c, v["image_size"], device=dev) for c in clip_ids], 0)     # [B*F, 3, H, W]
    ids = [S.prompt_ids(F, c, 30, l["vocab_size"]) for c in clip_ids]
    Mseq = _rup(L0, 32)
    emb_all = torch.zeros(G * Mseq, l["hidden_size"], dtype=torch.float16, device=dev)
    plans = [eng.splice_plan(ids[b], F, n_kept) for b in range(B)]   # static per prompt: uploaded once, outside the loop
    torch.cuda.synchronize()
    ttft_ms = []
    host_ttft, submit_ttft = [], []
    power = None
    want_power = not args.no_power and not any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ)

    def front(bank, record_ttft=False):
        """ViT + projector/splice + prefill of one batch into generation bank `bank` (enqueue only)."""
        eng.select_bank(bank)
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        eng.begin_batch(B, N, None)
        evs = []
        vis, v0 = None, 0
        for b0 in range(0, B, G):                                  # equal-length prompts: G clips per prefill pass
            n = min(G, B - b0)
            if vis is None or b0 + n > v0 + VC:                    # ViT for the next VC clips (all B by default)
                v0 = b0
                vis = eng.vit_encode(pixels[v0 * F:min(B, v0 + VC) * F], r)      # [clips*F, n_kept, Dv]
            for j in range(n):
                _, L = eng.project_splice(vis[(b0 - v0 + j) * F:(b0 - v0 + j + 1) * F], plan=plans[b0 + j], out=emb_all[j * Mseq:(j + 1) * Mseq])
                assert L == L0
            eng.prefill_batch(b0, n, emb_all, L0)
            if args.sync_front:
                torch.cuda.current_stream().synchronize()
            if record_ttft:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.extend([e] * n)
        return ev0, evs

    def back(bank, gather=True):
        """Decode of the batch in `bank`, result copy (synchronises the decode stream) and the cross-rank gather.
        gather=False for rank-local passes (the instrumented step runs on rank 0 only: no collective may be in it)."""
        eng.select_bank(bank)
        if args.decode_chunk > 0 or args.sync_chunks:              # profiling aid: bound the number of queued graph launches
            dc = args.decode_chunk if args.decode_chunk > 0 else 64
            for s0 in range(0, N - 1, dc):
                eng.decode(min(dc, N - 1 - s0))
                torch.cuda.current_stream().synchronize()
        else:
            d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            d0.record()
            eng.decode(N - 1)
            d1.record()
            dec_ev.append((d0, d1))
        out = eng.outputs()
        if use_dist and gather:
            parallel.gather_results(out, N, B, cdev)               # RCCL all_gather over xGMI
        return out

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    out = None
    dec_ev = []                                                    # (start, end) events around every un-chunked decode of N - 1 steps
    host_enq = {"enqueue_s": 0.0, "cycles": 0}                     # host seconds spent ENQUEUEING a cycle (before the blocking read-back)
    serving_split = None
    NG, S = (B // G if B % G == 0 else 0), N - 1                   # groups of G slots; decode steps per caption after its prefill
    continuous = not (args.batch_mode or pipe or args.decode_chunk > 0) and NG >= 2 and N >= 2 * NG
    batch_ref = None

    def step(record_ttft=False):
        ev0, evs = front(0, record_ttft)
        o = back(0)
        if record_ttft:
            ttft_ms.extend(ev0.elapsed_time(e) for e in evs)
        return o

    if continuous:
        # ---- steady-state continuous batching: all B KV slots decode all the time; the slots form B / G groups whose captions
        #      end group by group, evenly spaced over the N - 1 decode steps of a caption.  A finished group is collected on the
        #      device (aur_slot_collect: no host sync), reset and re-filled with its next G clips (ViT + splice + ONE prefill pass
        #      of G sequences) between two decode chunks.  One "step" = one cycle of N - 1 decode steps = B captions completed
        #      and B front ends run: exactly the work of one batch-mode step, but a clip's first token waits for its own
        #      group's front end only.  (What the reference harness does one clip at a time, evaluator.py:406-411.)
        t_b = time.perf_counter()
        batch_ref = step(True)                                     # one batch-mode step: reference captions + its timing
        torch.cuda.synchronize()
        batch_ms, batch_ttft = 1e3 * (time.perf_counter() - t_b), float(np.median(ttft_ms))
        ttft_ms.clear()
        offs = [g * S // NG for g in range(NG)]                    # refill instants of the groups inside a cycle of S steps
        ids_out = torch.zeros(NG, G, N, dtype=torch.int32, device=dev)
        len_out = torch.zeros(NG, G, dtype=torch.int32, device=dev)
        eng.select_bank(0)
        eng.begin_batch(B, N, None)
        for sl in range(B):
            eng.slot_retire(sl)                                    # empty slots: finished, positions frozen
        lat_ev = []

        def refill(g, collect, timed):
            if collect:
                eng.slot_collect(g * G, G, ids_out[g], len_out[g])
            for sl in range(g * G, (g + 1) * G):
                eng.slot_reset(sl)
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            vis = eng.vit_encode(pixels[g * G * F:(g + 1) * G * F], r)
            for j in range(G):
                eng.project_splice(vis[j * F:(j + 1) * F], plan=plans[g * G + j], out=emb_all[j * Mseq:(j + 1) * Mseq])
            eng.prefill_batch(g * G, G, emb_all, L0)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            if timed:
                lat_ev.append((e0, e1))

        def cycle(fill, timed):
            """S decode steps; group g is (re)filled before step offs[g].  Returns the captions collected in this cycle."""
            for g in range(NG):
                refill(g, collect=not fill, timed=timed)
                eng.decode((offs[g + 1] if g + 1 < NG else S) - offs[g])
                if args.sync_chunks:
                    torch.cuda.current_stream().synchronize()
            if fill:
                return None
            got_ids, got_len = ids_out.cpu().numpy(), len_out.cpu().numpy()         # synchronises: the cycle has run
            o = [got_ids[g, j, :got_len[g, j]].tolist() for g in range(NG) for j in range(G)]
            if use_dist:
                parallel.gather_results(o, N, B, cdev)             # RCCL all_gather over xGMI, once per cycle
            return o

        cycle(True, False)                                         # fill: the groups enter one after another (untimed set-up)
        seq_sched = None
        masked = None
        if overlap:
            try:
                from aurora_amd.streams import cu_masked_stream
                fc = args.front_cus if args.front_cus > 0 else 16
                masked = (cu_masked_stream(fc, device=dev), cu_masked_stream(32 - fc, from_top=True, device=dev))
            except (RuntimeError, OSError, AttributeError) as e:      # no CU-mask support here: keep the sequential schedule
                print(f"bench.py: CU-masked streams unavailable ({e}); front ends run between decode chunks", file=sys.stderr)
                overlap = False
        if overlap and not args.no_sequential_point:
            # one cycle of the sequential schedule first (front ends BETWEEN decode chunks on one stream): the latency-oriented
            # operating point, reported beside the line
            torch.cuda.synchronize()
            t_s = time.perf_counter()
            o_seq = cycle(False, True)
            torch.cuda.synchronize()
            seq_ms = 1e3 * (time.perf_counter() - t_s)
            assert o_seq == batch_ref, "continuous batching (sequential schedule) produced different captions than the batch-mode step"
            seq_sched = {"captions_per_s": B / (seq_ms / 1e3), "ms_per_step": seq_ms,
                         "p50_ttft_ms": float(np.median([a.elapsed_time(b) for a, b in lat_ev])),
                         "note": "one cycle with each group's front end run between two decode chunks on the decode stream (--overlap 0)"}
            lat_ev.clear()
        if overlap:
            # ---- the front end of a group's NEXT clips runs on its own stream, restricted to `fc` CUs of every XCD, while all B
            #      slots keep decoding (HBM-bound decode next to MFMA-bound ViT / prefill): aur_llm_prefill_stage writes spare KV
            #      sequences B .. B + G - 1, and at the group's boundary aur_llm_prefill_commit (decode stream) exchanges page-table
            #      rows and produces the first tokens.  The first `k_masked` decode steps of a chunk run on the complementary
            #      mask (the front end is in flight), the rest unmasked.
            half_grid = 1 if args.half_grid else 0                # default 1
            # cap: the whole steps of a chunk (chunks are S // NG or S // NG + 1 steps long).  Rounds 2-3 used four fifths of that; with the
            # masked steps on the half grid one more masked step pays (7 against 6 of ~8: +0.45 %, three alternating pairs, r03_sched_sweep4)
            k_masked = args.overlap_steps if args.overlap_steps >= 0 else max(1, S // NG)
            # ... but never more steps than the front end needs: a chunk of a small-batch or long-caption config (cfg4: 127 steps, cfg5: 170)
            # is much longer than its front end, and every masked step beyond it runs on half of the CUs for nothing.  The warm-up
            # cycle measures the front end's duration beside the masked decode and the masked step: k = ceil(T_front / t_step),
            # capped by the chunk's whole steps (which is what binds in the default config, where a chunk IS a front end long).
            k_cal = {"front_ev": [], "dec_ev": [], "on": False, "info": None}
            sD = torch.cuda.current_stream()
            sF = masked[0]
            sDm = masked[1] if k_masked > 0 else None
            if args.gemm_cus <= 0:
                eng.set_option("gemm_max_wgs", 8 * fc)
            pending = [None]
            stage_ev = []
            stage_ev_eager = []

            # host-observed TTFT ("submit -> first generated token id on host"): the host clock from the moment a group's
            # front end is submitted to the moment its first token ids have landed in pinned host memory.  A side stream waits for the
            # commit, copies the group's ids (aur_slot_collect, device to device) and their first column to the host; a helper thread
            # waits for that copy and stamps the clock, so the enqueueing thread never blocks.
            import queue
            import threading
            sC = torch.cuda.Stream()
            first_dev = torch.zeros(NG, G, N, dtype=torch.int32, device=dev)
            first_len = torch.zeros(NG, G, dtype=torch.int32, device=dev)
            first_host = torch.zeros(NG, G, dtype=torch.int32).pin_memory()
            host_q, start_q, start_stamp = queue.Queue(), queue.Queue(), {}

            def start_poller():                                   # host clock at which a group's front end STARTED on the device
                while True:
                    item = start_q.get()
                    if item is None:
                        return
                    key, ev = item
                    sleep_wait(ev, 0.0002)
                    start_stamp[key] = time.perf_counter()
                    start_q.task_done()

            def poller():
                while True:
                    item = host_q.get()
                    if item is None:
                        return
                    key, t_submit, ev = item
                    sleep_wait(ev, 0.0002)
                    now = time.perf_counter()
                    submit_ttft.append(1e3 * (now - t_submit))
                    for _ in range(200):                          # the start stamp comes from the other helper thread
                        if key in start_stamp:
                            break
                        time.sleep(0.001)
                    if key in start_stamp:
                        host_ttft.append(1e3 * (now - start_stamp[key]))
                    host_q.task_done()

            poll_thread = threading.Thread(target=poller, daemon=True)
            start_thread = threading.Thread(target=start_poller, daemon=True)
            poll_thread.start()
            start_thread.start()
            front_seq = [0]

            # the captured front end (engine.FrontEndGraph): one hipGraph per shape bucket - here the single bucket (G clips of F frames,
            # this r, this prompt structure, spare sequences B ..) - replayed once per group after four device-to-device copies of its inputs
            fe = {"graph": None, "use": False, "launches": 0}
            plan_vr = torch.stack([pl["vis_rows"] for pl in plans]) if args.front_graph else None
            plan_ti = torch.stack([pl["text_ids"] for pl in plans]) if args.front_graph else None
            plan_tr = torch.stack([pl["text_rows"] for pl in plans]) if args.front_graph else None

            def front_async(g):
                t_host = time.perf_counter()
                with torch.cuda.stream(sF):
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(sF)
                    if fe["use"]:
                        fe["graph"].load(pixels[g * G * F:(g + 1) * G * F], vis_rows=plan_vr[g * G:(g + 1) * G], text_ids=plan_ti[g * G:(g + 1) * G],
                                         text_rows=plan_tr[g * G:(g + 1) * G])
                        fe["graph"].launch()
                        fe["launches"] += 1
                        ev_v = ev_p = None
                    else:
                        vis = eng.vit_encode(pixels[g * G * F:(g + 1) * G * F], r)
                        ev_v = torch.cuda.Event(enable_timing=True)
                        ev_v.record(sF)
                        for j in range(G):
                            eng.project_splice(vis[j * F:(j + 1) * F], plan=plans[g * G + j], out=emb_all[j * Mseq:(j + 1) * Mseq])
                        ev_p = torch.cuda.Event(enable_timing=True)
                        ev_p.record(sF)
                        eng.prefill_stage(B, G, emb_all, L0)
                    evf = torch.cuda.Event(enable_timing=True)
                    evf.record(sF)
                if k_cal["on"]:
                    k_cal["front_ev"].append((e0, evf))
                front_seq[0] += 1
                start_q.put((front_seq[0], e0))
                return e0, evf, (front_seq[0], t_host), (ev_v, ev_p)

            def cycle(fill, timed):                                # noqa: F811 - the overlapped cycle replaces the sequential one
                t_enq, c_enq = time.perf_counter(), time.thread_time()
                for g in range(NG):
                    eng.slot_collect(g * G, G, ids_out[g], len_out[g])
                    e0, evf, t_host, (ev_v, ev_p) = pending[0]
                    sD.wait_event(evf)
                    ec0 = torch.cuda.Event(enable_timing=True)
                    ec0.record(sD)                                  # the decode stream has reached the group's boundary AND the front end is done
                    eng.prefill_commit(g * G, G, B, emb_all, L0)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(sD)
                    if timed:
                        lat_ev.append((e0, e1))
                        stage_ev.append((e0, ev_v, ev_p, evf, ec0, e1))
                        with torch.cuda.stream(sC):
                            sC.wait_event(e1)
                            eng.slot_collect(g * G, G, first_dev[g], first_len[g])
                            first_host[g].copy_(first_dev[g, :, 0], non_blocking=True)
                            e2 = torch.cuda.Event()
                            e2.record(sC)
                        host_q.put((t_host[0], t_host[1], e2))
                    # One chunk of decode steps: `lead` steps on the whole chip, then - `go` - the next group's front end may start on the front-end
                    # stream while `k1` steps run on the complementary CU mask, then the rest unmasked.  Latency-priority order (round 5): the
                    # steps WITHOUT a front end beside them come first, so the front end ends at its own boundary (no commit wait).
                    def chunk_plan(gi):
                        n_ = (offs[gi + 1] if gi + 1 < NG else S) - offs[gi]
                        k_ = min(n_, k_masked) if sDm is not None else 0
                        l_ = (n_ - k_ if args.ttft_delay_steps < 0 else min(args.ttft_delay_steps, n_ - k_)) if k_ > 0 else 0
                        return n_, k_, l_

                    n, k1, lead = chunk_plan(g)
                    rest = n - k1 - lead
                    # bounded run-ahead (TTFT runs from the SUBMISSION of a request).  The front end of group g + 1 may start after
                    # `lead` steps of this chunk; the host submits it when the device is `margin` = --ttft-gate-steps steps before that
                    # point - an event recorded inside this chunk's lead segment, or (lead < margin) inside the PREVIOUS chunk's tail - and
                    # never earlier.  The decode replays themselves are enqueued at most three blocks of 8 steps ahead of the device (the
                    # enqueue thread sleeps instead of spinning in the runtime's back-pressure: cfg5's chunks are 170 steps long).
                    margin = max(0, args.ttft_gate_steps)
                    marks = {}                                             # steps of this chunk enqueued -> events to record there, on the current stream
                    gate_in, tail = [None], [None]
                    if gate_on:
                        pos_in, pos_tail = gate_positions(n, lead, chunk_plan((g + 1) % NG)[2], margin)
                        if pos_in is not None:
                            marks.setdefault(pos_in, []).append(lambda: gate_in.__setitem__(0, _record()))
                        if pos_tail is not None:
                            marks.setdefault(pos_tail, []).append(lambda: tail.__setitem__(0, _record()))

                    def _record():
                        ev_ = torch.cuda.Event()
                        ev_.record(torch.cuda.current_stream())
                        return ev_

                    def decode_seg(steps, base):
                        i_ = 0
                        while True:
                            for fn_ in marks.pop(base + i_, ()):
                                fn_()
                            if i_ >= steps:
                                break
                            stops = [m - base for m in marks if base + i_ < m <= base + steps] + [steps] + ([i_ + 8] if gate_on else [])
                            nxt_ = min(stops)
                            eng.decode(nxt_ - i_)
                            i_ = nxt_
                            if gate_on:                                     # pacing: at most three blocks of replays queued ahead of the device
                                pace_q.append(_record())
                                if len(pace_q) > 3:
                                    sleep_wait(pace_q.pop(0))

                    go = e1
                    if lead > 0:
                        decode_seg(lead, 0)
                        go = torch.cuda.Event()
                        go.record(sD)
                    else:
                        for fn_ in marks.pop(0, ()):                        # lead == 0: a mark at the chunk's start sits right behind the commit
                            fn_()
                    sF.wait_event(go)
                    gate = gate_in[0] if gate_in[0] is not None else gate_prev[0]
                    if gate_on and gate is not None:
                        t_w = time.perf_counter()
                        sleep_wait(gate)                                    # the device is `margin` steps from the point where this front end may start
                        if timed:
                            host_enq["gate_wait_s"] = host_enq.get("gate_wait_s", 0.0) + time.perf_counter() - t_w
                    # the front end is ONE graph replay now: it is submitted before the rest of the chunk's replays (eagerly enqueued - a few
                    # hundred launches on the host - it goes last: the decode must never sit behind the host's enqueue time)
                    if fe["use"]:
                        pending[0] = front_async((g + 1) % NG)
                    if k1 > 0:
                        sDm.wait_event(go)
                        with torch.cuda.stream(sDm):
                            if k_cal["on"]:
                                evm0 = torch.cuda.Event(enable_timing=True)
                                evm0.record(sDm)
                            eng.set_option("decode_half_grid", half_grid)   # half as many workgroups, twice the tiles each
                            try:
                                decode_seg(k1, lead)
                            finally:                                        # a failed decode must not leave the ctx on the half grid
                                eng.set_option("decode_half_grid", 0)
                            evm = torch.cuda.Event(enable_timing=k_cal["on"])
                            evm.record(sDm)
                            if k_cal["on"]:
                                k_cal["dec_ev"].append((evm0, evm, k1))
                        sD.wait_event(evm)
                    if rest > 0:
                        decode_seg(rest, lead + k1)
                    gate_prev[0] = tail[0]
                    if not fe["use"]:
                        pending[0] = front_async((g + 1) % NG)
                    if args.sync_chunks:
                        sD.synchronize()
                if timed:
                    host_enq["enqueue_s"] += time.perf_counter() - t_enq
                    host_enq["enqueue_cpu_s"] = host_enq.get("enqueue_cpu_s", 0.0) + time.thread_time() - c_enq
                    host_enq["cycles"] += 1
                if gate_on:                                                 # the read-back below blocks: sleep until the cycle has run instead of spinning in it
                    e_end = torch.cuda.Event()
                    e_end.record(sD)
                    sleep_wait(e_end)
                got_ids, got_len = ids_out.cpu().numpy(), len_out.cpu().numpy()
                o = [got_ids[g, j, :got_len[g, j]].tolist() for g in range(NG) for j in range(G)]
                if use_dist:
                    parallel.gather_results(o, N, B, cdev)
                return o

            sF.wait_stream(sD)
            gate_on = args.ttft_gate_steps >= 0
            gate_prev = [None]
            pace_q = []
            if args.front_graph:
                from aurora_amd.engine import FrontEndGraph
                t_cap = time.perf_counter()
                try:
                    with torch.cuda.stream(sF):                    # captured on the stream that replays it, with the front end's GEMM grid (gemm_max_wgs)
                        fe["graph"] = FrontEndGraph(eng, G, F, v["image_size"], v["image_size"], r, plans[0], seq0=B, embeds=emb_all)
                        fe["graph"].launch()                       # first replay (uploads the graph) outside every timed interval
                        fe["nodes"] = fe["graph"].nodes
                    sF.synchronize()
                    fe["capture_s"] = time.perf_counter() - t_cap
                except Exception as ex:                            # noqa: BLE001 - a runtime that cannot capture: the line still gets measured, eagerly
                    print(f"bench.py: front-end graph capture failed ({ex!r}); front ends are enqueued eagerly", file=sys.stderr)
                    fe["graph"], fe["capture_error"] = None, repr(ex)
            pending[0] = front_async(0)
            k_cal["on"] = args.overlap_steps < 0 and sDm is not None      # measured during the warm-up cycle(s) below, applied after them
        n_warm = max(args.warmup - 1, 1 if overlap else 0)          # overlap: the first front end above overlapped nothing
        for wi in range(n_warm):
            # the FIRST warm-up cycle enqueues its front ends eagerly with per-stage events (the ViT / splice / prefill split of a serving
            # TTFT, `ttft_stage_ms.serving`: events cannot be recorded inside the captured graph); every later cycle replays the graph
            cycle(False, bool(overlap and fe["graph"] is not None and wi == 0))
            if overlap and fe["graph"] is not None:
                if wi == 0:
                    stage_ev_eager = list(stage_ev)
                    stage_ev.clear()
                    lat_ev.clear()
                fe["use"] = True
        if overlap and fe["graph"] is not None and n_warm == 0:
            fe["use"] = True
        fence()
        if overlap:
            start_q.join()                                         # the helper threads have stamped everything the warm-up cycles submitted
            host_q.join()
            submit_ttft.clear()
            host_ttft.clear()
            for k_ in ("enqueue_s", "enqueue_cpu_s", "gate_wait_s"):
                host_enq[k_] = 0.0
            host_enq["cycles"] = 0
        stamps_before = eng.decode_stamps()[1] if stamp_layer >= 0 else 0
        if overlap and k_cal["on"]:
            k_cal["on"] = False
            t_front = float(np.median([a.elapsed_time(b) for a, b in k_cal["front_ev"][:-1]]))      # the last one ran beside nothing (fence)
            t_step = float(np.median([a.elapsed_time(b) / k for a, b, k in k_cal["dec_ev"]]))
            k_cal["info"] = {"front_end_ms_beside_masked_decode": t_front, "masked_decode_step_ms": t_step,
                             "steps_needed": int(np.ceil(t_front / max(t_step, 1e-3))), "cap_whole_steps_of_a_chunk": k_masked}
            k_masked = masked_steps_per_chunk(t_front, t_step, k_masked)
            k_cal["front_ev"].clear()
            k_cal["dec_ev"].clear()
        sampler = PowerSampler(local).start() if (rank == 0 and want_power) else None
        t_start = time.perf_counter()
        cpu_start = time.process_time()
        thr0 = thread_cpu_seconds()
        outs = []
        for _ in range(args.steps):
            out = cycle(False, True)
            outs.append(out)
        fence()
        elapsed = time.perf_counter() - t_start
        host_enq["cpu_s"] = time.process_time() - cpu_start
        thr1 = thread_cpu_seconds()
        host_enq["threads"] = sorted(((n, round((c - thr0.get(t, (n, 0.0))[1]) / args.steps, 3)) for t, (n, c) in thr1.items()), key=lambda x: -x[1])[:6]
        power = sampler.stop() if sampler else None
        if stamp_layer >= 0:
            us_, tot_ = eng.decode_stamps()
            n_t = int(min(tot_ - stamps_before, len(us_)))
            stamps_timed = us_[len(us_) - n_t:] if n_t > 0 else None
        # same clips in the same slots as the batch-mode step: batch-invariant kernels must give the same ids, every cycle
        assert all(o == batch_ref for o in outs), "continuous batching produced different captions than the batch-mode step"
        if overlap and args.gemm_cus <= 0:
            eng.set_option("gemm_max_wgs", 0)                      # the instrumented pass below runs alone on the whole GPU
        ttft_ms.extend(a.elapsed_time(b) for a, b in lat_ev for _ in range(G))
        if overlap and stage_ev:
            med = lambda i, j, evs=stage_ev: float(np.median([ev[i].elapsed_time(ev[j]) for ev in evs]))
            inner = stage_ev_eager if fe["use"] else stage_ev       # events inside the front end exist for eagerly enqueued front ends only
            serving_split = {"vit_and_tome": med(0, 1, inner) if inner else None, "projector_splice": med(1, 2, inner) if inner else None,
                             "prefill_layers": med(2, 3, inner) if inner else None, "front_end": med(0, 3),
                             "commit_wait": med(3, 4), "lm_head_argmax_commit": med(4, 5), "total": med(0, 5),
                             "note": "medians over the timed cycles' groups of %d clips, device events: the front-end stream (16 CUs of every XCD, beside the "
                                     "masked decode) runs ViT + ToMe, projector + splice and the staged prefill layer stack; `commit_wait` is the time the "
                                     "finished front end waits for the decode stream to reach the group's boundary; the commit (decode stream) swaps the "
                                     "page-table rows and produces the first tokens" % G
                                     + ("; the front end of a timed cycle is ONE hipGraph replay (`front_end`), so its inner split (vit_and_tome / "
                                        "projector_splice / prefill_layers) comes from the first warm-up cycle, whose front ends were enqueued eagerly "
                                        "with events between the stages" if fe["use"] else "")}
        if overlap:
            host_q.put(None)
            start_q.put(None)
            poll_thread.join()
            start_thread.join()
            # the copied first ids are the captions' first ids (the read-back is real, not a timestamp of nothing)
            assert all(int(first_host[g, j]) == batch_ref[g * G + j][0] for g in range(NG) for j in range(G)), "host read-back of the first tokens disagrees"
    elif not pipe:
        for _ in range(args.warmup):
            step()
        fence()
        sampler = PowerSampler(local).start() if (rank == 0 and want_power) else None
        t_start = time.perf_counter()
        outs = []
        for _ in range(args.steps):
            out = step(True)
            outs.append(out)
        fence()
        elapsed = time.perf_counter() - t_start
        power = sampler.stop() if sampler else None
        # every step captions the same synthetic clips: the ids must repeat exactly (a race or a stale buffer would show)
        assert all(o == outs[0] for o in outs), "generated ids differ between identical steps"
    else:
        sD = torch.cuda.current_stream()                           # the engine's stream: decode
        sP = torch.cuda.Stream()                                   # front end of the next batch
        if args.front_cus > 0 or args.decode_cus > 0:
            from aurora_amd.streams import cu_masked_stream
            if args.front_cus > 0:
                sP = cu_masked_stream(args.front_cus, device=dev)
                if args.gemm_cus <= 0:
                    eng.set_option("gemm_max_wgs", 8 * args.front_cus)
            if args.decode_cus > 0:
                sDm = cu_masked_stream(args.decode_cus, from_top=True, device=dev)
                sDm.wait_stream(sD)
                torch.cuda.set_stream(sDm)
                sD = sDm
        sP.wait_stream(sD)
        ev_front = [None, None]
        ev_back = [None, None]
        pending = []

        def iteration(i, record_ttft):
            bank = i & 1
            with torch.cuda.stream(sP):                            # front end of batch i+1 into the other bank
                if ev_back[bank ^ 1] is not None:
                    sP.wait_event(ev_back[bank ^ 1])               # its previous occupant must have finished decoding
                ev0, evs = front(bank ^ 1, record_ttft)
                ev_front[bank ^ 1] = torch.cuda.Event()
                ev_front[bank ^ 1].record(sP)
                if record_ttft:
                    pending.append((ev0, evs))
            sD.wait_event(ev_front[bank])                          # batch i was prefetched one iteration ago
            o = back(bank)
            ev_back[bank] = torch.cuda.Event()
            ev_back[bank].record(sD)
            return o

        with torch.cuda.stream(sP):                                # prime: front end of batch 0
            front(0)
            ev_front[0] = torch.cuda.Event()
            ev_front[0].record(sP)
        for i in range(args.warmup):
            iteration(i, False)
        fence()
        t_start = time.perf_counter()
        outs = []
        for i in range(args.warmup, args.warmup + args.steps):     # K front ends + K decodes
            out = iteration(i, True)
            outs.append(out)
        fence()
        elapsed = time.perf_counter() - t_start
        assert all(o == outs[0] for o in outs), "generated ids differ between identical steps (pipelined banks)"
        for ev0, evs in pending:
            ttft_ms.extend(ev0.elapsed_time(e) for e in evs)
        eng.select_bank(0)
    assert all(len(o) == N for o in out), [len(o) for o in out]        # EOS disabled: every clip produced N tokens
    import zlib
    ids_checksum = zlib.crc32(np.asarray(out, dtype=np.int32).tobytes())     # of this rank's last step: the same command must reproduce it
    # the job's result in CLIP order: every rank's ids gathered (the path's one collective) and the round-robin shard undone
    # (parallel.merge_round_robin; evaluator.py:519-546 gathers and re-orders the same way); one crc per clip goes into the line
    per_rank_out = parallel.gather_results(out, N, B, cdev) if use_dist else {0: out}
    job = parallel.merge_round_robin(per_rank_out, world * B) if use_dist else out
    assert all(c is not None and len(c) == N for c in job), "a clip of the job has no result after the gather"
    crc_per_clip = [zlib.crc32(np.asarray(c, dtype=np.int32).tobytes()) for c in job]
    per_rank_ms, gather_us = None, None
    if use_dist:
        # every rank's own clock around the same K steps (the line's time is their MAX), so that a slow rank or a slow link shows in
        # the record itself; and the cost of the one collective of the path, timed on its own after the steps
        mine = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        allt = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank_ms = [1e3 * float(t.item()) / args.steps for t in allt]
        elapsed = max(float(t.item()) for t in allt)
        cyc = max(host_enq["cycles"], 1)
        mine_h = torch.tensor([host_enq["enqueue_s"] / cyc, host_enq.get("enqueue_cpu_s", 0.0) / cyc, host_enq.get("cpu_s", 0.0) / args.steps],
                              dtype=torch.float64, device=cdev)
        allh = [torch.empty_like(mine_h) for _ in range(world)]
        dist.all_gather(allh, mine_h)
        host_enq["per_rank"] = [[float(x) for x in t.tolist()] for t in allh]
        fence()
        t_g = time.perf_counter()
        for _ in range(5):
            parallel.gather_results(out, N, B, cdev)
        if cdev != "cpu":
            torch.cuda.synchronize()
        gather_us = 1e6 * (time.perf_counter() - t_g) / 5

    result = None
    if rank == 0:
        captions = world * B * args.steps
        value = captions / elapsed
        result = {
            "metric": "captions/sec (AuroraCap-7B, %d-frame clips, token_kept_ratio %g, %d new tokens) + p50 TTFT" % (F, args.token_kept_ratio, N),
            "value": value, "unit": "captions/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "rccl_ranks": world if (use_dist and backend == "nccl") else (0 if use_dist else 1), "dist_backend": backend if use_dist else "none",
            "ms_per_step": 1e3 * elapsed / args.steps, "ms_per_step_per_rank": per_rank_ms, "ids_all_gather_us": gather_us, "ids_checksum_rank0": ids_checksum,
            "clips_of_rank0": clip_ids[:4] + (["..."] if len(clip_ids) > 4 else []), "ids_crc_per_clip": crc_per_clip,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": ("AuroraCap-7B-VID %d-frame video, token_kept_ratio=%g, greedy %d tokens (BASELINE configs[%d]%s)"
                                    % (F, args.token_kept_ratio, N, CONFIGS[args.config_name]["index"],
                                       "" if (F, args.token_kept_ratio, N) == tuple(CONFIGS[args.config_name]["set"][k] for k in ("num_frm", "token_kept_ratio", "max_new_tokens"))
                                       else " with overridden flags")) if not args.tiny else "tiny plumbing config (NOT the metric)",
                       "preset": args.config_name,
                       "clips_per_gpu_per_step": B, "frames": F, "token_kept_ratio": args.token_kept_ratio, "r_per_layer": r,
                       "visual_tokens_per_clip": F * n_kept, "prefill_len": L0, "max_new_tokens": N, "parallelism": f"clip-parallel x{world}",
                       "decode": "hipGraph" if not args.no_graph else "eager", "prefill_group": G, "vit_chunk": G if continuous else VC,
                       "mode": ("continuous batching, steady state: %d KV slots always decoding; groups of %d clips are collected and re-filled "
                                "(ViT + splice + one prefill pass) every %d decode steps; one step = one cycle of %d decode steps = %d captions "
                                "completed + %d front ends" % (B, G, S // NG, S, B, B)
                                + ("; the front ends run on their own stream on %d CUs of every XCD while all slots decode (staged prefill into "
                                   "spare KV sequences, committed at the group's boundary; %d decode steps per chunk on the other %d CUs per XCD)"
                                   % (fc, k_masked, 32 - fc) if overlap else "")) if continuous else "batch: front end of all clips, then B-wide decode",
                       "decode_half_grid_on_masked_steps": bool(args.half_grid) if (continuous and overlap) else None,
                       "pipeline": "decode(batch i) || ViT+prefill(batch i+1) on two streams / two KV banks" if pipe else "none"},
            "overlap_steps_calibration": (k_cal["info"] if (continuous and overlap) else None),
            # TTFT = submit -> first generated token id on host.  The overlapped schedule bounds the host's run-ahead
            # (--ttft-gate-steps) and reports the host clock from the SUBMISSION of a group's front end to its first ids in pinned host
            # memory; the device-event interval (what rounds 1-5 reported under this name) stays beside it
            "p50_ttft_ms": (float(np.median(submit_ttft)) if (continuous and overlap and gate_on and submit_ttft) else (float(np.median(ttft_ms)) if ttft_ms else None)),
            "p50_ttft_definition": ("host clock: submission of a group's front end (host call) -> its first token ids in pinned host memory; the host submits a "
                                    "front end only when the device is within %d decode step(s) of the point where that front end may start (bounded run-ahead)"
                                    % max(args.ttft_gate_steps, 0)) if (continuous and overlap and gate_on and submit_ttft) else "device-event interval (see ttft_note)",
            "p50_ttft_device_ms": float(np.median(ttft_ms)) if ttft_ms else None,
            "p90_ttft_ms": (float(np.percentile(submit_ttft, 90)) if (continuous and overlap and gate_on and submit_ttft) else None),
            "power": power, "power_sampling": ("off" if not (rank == 0 and want_power) else (power or {}).get("how", "no samples")),
            "p50_ttft_host_ms": (float(np.median(host_ttft)) if (continuous and overlap and host_ttft) else None),
            "ttft_host_note": ("host clock from the moment a group's front end STARTS on the device (a helper thread waits for its first event) to its "
                               "first token ids sitting in pinned host memory (a side stream copies them after the commit; a second helper thread "
                               "waits for the copy), same cycles as the timed region.  p50_submit_to_first_token_host_ms starts the clock at the host "
                               "call that SUBMITS the front end instead (= p50_ttft_ms when the run-ahead is bounded, --ttft-gate-steps >= 0; with "
                               "--ttft-gate-steps -1 the enqueue thread runs a cycle ahead of the device and that figure is its queue depth)"
                               if (continuous and overlap) else None),
            "p50_submit_to_first_token_host_ms": (float(np.median(submit_ttft)) if (continuous and overlap and submit_ttft) else None),
            "ttft_note": (("device-event interval from the start of a group's front end (ViT + ToMe + projector + splice + prefill of its %d clips) to "
                           "its first tokens; a request arriving while a decode chunk is queued also waits for that chunk (<= %d steps here)" % (G, S // NG + 1)
                           + ("; the front end runs on the front-end stream during the chunk before the group's boundary and its first tokens "
                              "are produced by the commit at that boundary on the decode stream" if overlap else ""))
                          if continuous else
                          "time from the start of a batch's front end to each clip's first token, batch of %d clips (ViT in chunks of %d clips, "
                          "prefill in groups of %d)" % (B, VC, G) + ("; the front end shares the GPU with the previous batch's decode" if pipe else "")),
        }
        # ---- the host side of a cycle (on an 8-rank node the host is the shared resource)
        try:
            aff = sorted(os.sched_getaffinity(0))
        except AttributeError:
            aff = []
        result["host"] = {
            "enqueue_s_per_cycle": (host_enq["enqueue_s"] / host_enq["cycles"]) if host_enq["cycles"] else None,
            "enqueue_frac_of_cycle": (host_enq["enqueue_s"] / host_enq["cycles"] / (elapsed / args.steps)) if host_enq["cycles"] else None,
            "enqueue_thread_cpu_s_per_cycle": (host_enq["enqueue_cpu_s"] / host_enq["cycles"]) if host_enq.get("cycles") and "enqueue_cpu_s" in host_enq else None,
            "per_rank": ([{"enqueue_s_per_cycle": a, "enqueue_thread_cpu_s_per_cycle": b, "process_cpu_s_per_cycle": c_} for a, b, c_ in host_enq["per_rank"]]
                         if "per_rank" in host_enq else None),
            "process_cpu_s_per_cycle": (host_enq["cpu_s"] / args.steps) if "cpu_s" in host_enq else None,
            "process_cpu_util": (host_enq["cpu_s"] / elapsed) if "cpu_s" in host_enq else None,
            "busiest_threads_cpu_s_per_cycle": host_enq.get("threads"),
            "busiest_threads_note": "user + system seconds per cycle of the process's threads over the timed steps (/proc/self/task); the thread that is busy for the "
                                    "whole cycle is not Python's: the enqueue thread is the one marked (main), the two TTFT helpers and the power sampler follow",
            "gate_wait_s_per_cycle": (host_enq.get("gate_wait_s", 0.0) / host_enq["cycles"]) if host_enq["cycles"] else None,
            "front_end": ({"form": "hipGraph replay per group (engine.FrontEndGraph)", "graph_nodes": fe.get("nodes"), "replays_per_cycle": NG,
                           "capture_s": fe.get("capture_s"), "decode_graph_replays_per_cycle": S,
                           "eager_launches_per_cycle_besides": "per group: 4 input copies, 2 result copies, page-table swap + 4 resets + first-token kernels of the commit"}
                          if (continuous and overlap and fe["use"]) else ({"form": "eager launches", "capture_error": (fe.get("capture_error") if (continuous and overlap) else None)} if continuous else None)),
            "cores_allowed": len(aff), "pinned_to_gpu_numa_cores": pinned,
            "note": "enqueue = host wall time of the Python thread that issues a cycle's launches, up to (not including) the blocking read-back of "
                    "the ids; process_cpu = user + system seconds of the whole process (enqueue thread, the two TTFT helper threads, the power sampler) "
                    "per cycle.  Ranks of a multi-GPU job are bound to disjoint slices of their GPU's NUMA-local cores (aurora_amd.parallel.pin_rank_to_gpu_numa)"}
        if dec_ev:
            dms = float(np.median([a.elapsed_time(b) for a, b in dec_ev]))
            d_, mlp_ = l["hidden_size"], l["intermediate_size"]
            wb = 2.0 * (l["num_hidden_layers"] * (4 * d_ * d_ + 3 * d_ * mlp_) + l["vocab_size"] * d_)
            kvb = B * 2.0 * l["num_hidden_layers"] * d_ * 2 * (L0 + N / 2.0)
            result["decode_step"] = {"bound": "hbm", "slots": B, "ms_per_step": dms / (N - 1), "tokens_per_s": B * (N - 1) / (dms * 1e-3),
                                     "ideal_ms_per_step": 1e3 * (wb + kvb) / 8e12, "frac": (1e3 * (wb + kvb) / 8e12) / (dms / (N - 1)),
                                     "ideal_tokens_per_s": B * 8e12 / (wb + kvb), "weight_bytes_per_step": wb, "kv_bytes_per_step_mean": kvb,
                                     "how": "device events around the %d hipGraph replays of one generation on %d slots, nothing else on the GPU "
                                            "(median of %d); ideal = weights once + K / V of every cached token at 8 TB/s (SURVEY 8d)" % (N - 1, B, len(dec_ev))}
        if continuous and overlap:
            result["sequential_schedule"] = seq_sched
        if continuous:
            result["batch_mode"] = {"captions_per_s": B / (batch_ms / 1e3), "ms_per_step": batch_ms, "p50_ttft_ms": batch_ttft,
                                    "note": "one step of the non-continuous schedule (all %d front ends, then the %d-wide decode), same captions; "
                                            "host-timed around a single step after warm kernels" % (B, B)}

    if rank == 0:
        pts = [{"schedule": "this run (`value`)", "prefill_group": G, "overlap": bool(continuous and overlap), "captions_per_s": result["value"],
                "p50_ttft_ms": result["p50_ttft_ms"]}]
        if continuous and overlap and seq_sched:
            pts.append({"schedule": "sequential (front ends between decode chunks)", "prefill_group": G, "overlap": False,
                        "captions_per_s": seq_sched["captions_per_s"], "p50_ttft_ms": seq_sched["p50_ttft_ms"]})
        if continuous:
            pts.append({"schedule": "batch mode (all front ends, then the decode)", "prefill_group": G, "overlap": False,
                        "captions_per_s": B / (batch_ms / 1e3), "p50_ttft_ms": batch_ttft})
        result["frontier"] = pts

    # ---- instrumented pass (rank 0, after the timed region, not pipelined): HIP events per stage and around the
    #      two HBM-bound decode kernels
    if rank == 0 and not args.no_instrument:
        torch.cuda.synchronize()
        eng.select_bank(0)
        eng.profile(True)
        front(0)
        back(0, gather=False)
        stages = {k: eng.profile_read(k)[0] for k in ("vit", "project", "prefill", "decode")}
        tome_ms_batch, tome_n_batch = eng.profile_read("vit_tome")
        ams, an = eng.profile_read("decode_attn")
        kms, kn = eng.profile_read("decode_gemm_gateup")
        eng.profile(False)
        result["stage_ms_instrumented_step"] = stages
        d, mlp = l["hidden_size"], l["intermediate_size"]
        how = "HIP events around every launch of this kernel in one extra eager, un-pipelined step on the launch stream (after the timed steps)"
        # rocprofv3 --pmc summary of this same command (tools/profile_round.sh -> profiles/r02_pmc.json), if committed: per-kernel
        # HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE KiB, separate passes) measured at 6 new tokens, scaled to this run below
        pmc, pmc_ctx, pmc_tag, pmc_build, pmc_all = {}, None, "r02", None, []
        try:
            pj = None
            for tag in ("r06", "r05", "r04", "r03", "r02"):                   # the newest committed counter summary
                fp = os.path.join(ROOT, "profiles", f"{tag}_pmc.json")
                if os.path.exists(fp):
                    pj, pmc_tag = json.load(open(fp)), tag
                    break
            pmc_ctx = pj.get("_batch_x_ctx")
            pmc_build = (pj.get("_build") or {}).get("lib_sha16")
            for ent in pj["kernels"]:
                if "hbm_bytes_per_launch" in ent:
                    pmc_all.append(ent)
                    pmc.setdefault(ent["kernel"].split("(")[0].replace("void ", ""), ent)
        except Exception:
            pass
        pmc_of = lambda needle: next((v for k, v in pmc.items() if needle in k), None)
        import hashlib
        from aurora_amd import _lib as _L
        lib_sha = hashlib.sha256(open(_L.SO_PATH, "rb").read()).hexdigest()[:16]
        roof = {}
        if an > 0:
            # decode attention: K + V of every cached token of every sequence, all heads, read once per layer-step; the
            # context of a sequence at decode step s (1..N-1) is L0 + s keys -> mean over one generation
            mean_ctx = L0 + N / 2.0
            alg = B * mean_ctx * 2 * d * 2                               # K + V bytes (q and the split partials are < 0.1 %)
            avg_s = ams / an * 1e-3
            att_pmc = pmc_of("decode_attn_dot_kernel")
            roof["decode_attn"] = {"bound": "hbm", "kernel": "decode_attn_dot_kernel<4, 8> (paged decode attention, v_dot2c page pipeline)",
                                   "achieved": alg / avg_s / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / avg_s / 8e12,
                                   # HBM bytes per launch from the PMC counters, UNSCALED, at the counter pass's own context (the pass decodes
                                   # 6 tokens after the same prefill: B x 2145 cached tokens), with the algorithmic bytes at that context beside it
                                   "traffic": att_pmc["hbm_bytes_per_launch"] if att_pmc else None,
                                   "traffic_context": {"batch_x_cached_tokens": pmc_ctx, "algorithmic_bytes_at_that_context": (pmc_ctx * 2 * d * 2) if pmc_ctx else None,
                                                       "traffic_over_algorithmic": (att_pmc["hbm_bytes_per_launch"] / (pmc_ctx * 2 * d * 2)) if (att_pmc and pmc_ctx) else None,
                                                       "this_run_mean_batch_x_cached_tokens": B * mean_ctx},
                                   "traffic_note": "rocprofv3 --pmc (2*FETCH_SIZE + WRITE_SIZE) KiB of profiles/%s_pmc.json: separate passes of this command at the "
                                                   "same batch and 6 new tokens (tools/profile_round.sh); counters taken on library build %s, this run loaded %s (%s)"
                                                   % (pmc_tag, pmc_build, lib_sha, "the same build" if pmc_build == lib_sha else "a different build"),
                                   "algorithmic_bytes_per_launch": alg,
                                   "avg_launch_us": avg_s * 1e6, "launches_timed": an, "how": how}
            if stamps_timed is not None and len(stamps_timed):
                # the same kernel INSIDE the timed loop: replayed from the captured step, beside the front ends, partly on the half-chip
                # mask - two one-thread launches in the captured step stamp the device clock around layer `stamp_layer`'s attention launch
                st = np.asarray(stamps_timed)
                roof["decode_attn"].update({
                    "frac_in_timed_loop": alg / (st.mean() * 1e-6) / 8e12, "achieved_in_timed_loop": alg / (st.mean() * 1e-6) / 1e9,
                    "avg_launch_us_in_timed_loop": float(st.mean()),
                    "in_timed_loop": {"steps_stamped": int(len(st)), "layer": stamp_layer, "p10_us": float(np.percentile(st, 10)),
                                      "p50_us": float(np.percentile(st, 50)), "p90_us": float(np.percentile(st, 90)),
                                      "how": "device constant-rate clock stored by two one-thread launches that bracket this layer's attention launch inside the "
                                             "captured decode step (option decode_stamp_layer), every step of the timed cycles (the last %d kept); the interval "
                                             "includes two kernel boundaries (~3 us); algorithmic bytes at the mean context of a generation" % len(st)}})
        if kn > 0:
            alg = 2 * mlp * d * 2 + B * d * 2 + B * mlp * 2           # gate+up rows fp16 + x in + h out
            avg_s = kms / kn * 1e-3
            gu_name = "skinny_lds_kernel<6, 1, 2" if eng.max_batch > 64 else "skinny_lds_kernel<6, 2, 2"     # k phases per tile: a constant of the engine
            roof["decode_gateup"] = {"bound": "hbm", "kernel": gu_name + ", ...> (decode gate/up projection, x through LDS)",
                                     "achieved": alg / avg_s / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / avg_s / 8e12,
                                     "traffic": pmc_of(gu_name)["hbm_bytes_per_launch"] if pmc_of(gu_name) else None,
                                     "algorithmic_bytes_per_launch": alg,
                                     "avg_launch_us": avg_s * 1e6, "launches_timed": kn, "how": how}
        if roof:    # the dominant kernel = the one with the larger total time in the decode loop
            dom = max(roof, key=lambda k: roof[k]["avg_launch_us"] * roof[k]["launches_timed"])
            result["roofline"] = roof[dom]
            for k, val in roof.items():
                if k != dom:
                    result["roofline_" + k] = val
        # ---- MFMA-bound stages and the whole step against the roofline (SURVEY 8d work model; peaks: 8 TB/s, 2.5 PF dense fp16)
        if not args.tiny:
            Dv, mlpv, Lv = v["hidden_size"], v["intermediate_size"], v["num_hidden_layers"]
            hdv = Dv // v["num_attention_heads"]
            vit_fl = 2.0 * (t0tok - 1) * (3 * v["patch_size"] ** 2) * Dv
            t = t0tok
            for _ in range(Lv - 1):
                rl = min(r, (t - 1) // 2)
                vit_fl += 8.0 * t * Dv * Dv + 4.0 * t * t * Dv + 2.0 * ((t + 1) // 2) * (t // 2) * hdv + 4.0 * (t - rl) * Dv * mlpv
                t -= rl
            vit_fl *= F                                                                  # per clip
            proj_fl = 2.0 * F * n_kept * (Dv * d + d * d)
            per_tok = 2.0 * l["num_hidden_layers"] * (4 * d * d + 3 * d * mlp)
            pre_fl = per_tok * L0 + 2.0 * l["num_hidden_layers"] * d * L0 * L0 + 2.0 * l["vocab_size"] * d
            w_bytes = 2.0 * (l["num_hidden_layers"] * (4 * d * d + 3 * d * mlp) + l["vocab_size"] * d)
            dec_bytes = (N - 1) * w_bytes + B * sum(2.0 * l["num_hidden_layers"] * d * 2 * (L0 + s_) for s_ in range(1, N))   # per batch of B
            result["roofline_vit"] = {"bound": "mfma", "stage": "ViT-H + per-layer ToMe, %d clips (instrumented eager step)" % B,
                                      "achieved": B * vit_fl / (stages["vit"] * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                                      "frac": B * vit_fl / (stages["vit"] * 1e-3) / 2.5e15, "algorithmic_flops_per_clip": vit_fl, "traffic": None}
            result["roofline_prefill"] = {"bound": "mfma", "stage": "Llama prefill of %d x %d tokens in groups of %d (instrumented eager step)" % (B, L0, G),
                                          "achieved": B * pre_fl / (stages["prefill"] * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                                          "frac": B * pre_fl / (stages["prefill"] * 1e-3) / 2.5e15, "algorithmic_flops_per_clip": pre_fl, "traffic": None}
            try:                                                   # the prefill GEMMs alone, back to back (aur_microbench)
                eng.set_option("microbench_prefill_nseq", G)
                Mg = G * Mseq
                shapes = {"pre_qkv": 3 * d * d, "pre_o": d * d, "pre_gateup": 2 * mlp * d, "pre_down": mlp * d}
                us = {k: eng.microbench(k, 32) for k in shapes}
                fl = {k: 2.0 * Mg * nk for k, nk in shapes.items()}
                tot_fl, tot_us = sum(fl.values()), sum(us.values())
                g_ent = pmc_of("gemm256_kernel<0, 7>")
                result["roofline_prefill_gemm"] = {"bound": "mfma", "kernel": "gemm256_kernel (qkv, o, gate/up, down at M = %d)" % Mg,
                                                   "achieved": tot_fl / tot_us / 1e6, "peak": 2500.0, "unit": "TFLOP/s", "frac": tot_fl / tot_us / 2.5e9,
                                                   "per_projection_tflops": {k: fl[k] / us[k] / 1e6 for k in shapes}, "avg_launch_us": us,
                                                   "mfma_util_pmc": g_ent.get("mfma_util") if g_ent else None,
                                                   "clock_ghz_pmc": g_ent.get("clock_ghz") if g_ent else None,
                                                   "traffic": g_ent.get("hbm_bytes_per_launch") if g_ent else None,
                                                   "traffic_is": "L2-to-fabric bytes (FETCH_SIZE / WRITE_SIZE count the TCC_EA0 interface: reads served by the memory-side "
                                                                 "cache are included; this rocprofv3 has no DRAM-side counter, profiles/r06_rocprof_ea_counters.txt)",
                                                   "fabric_read_latency_l2_clk": ({"this_gemm": g_ent.get("ea_read_latency_clk"),
                                                                                   "decode_attention_streaming_from_hbm": (pmc_of("decode_attn_dot_kernel") or {}).get("ea_read_latency_clk"),
                                                                                   "how": "TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum of the counter pass: mean latency of an L2-to-fabric read; a "
                                                                                          "stream from HBM is the yardstick - the GEMM's operand re-reads come back several times sooner, i.e. mostly "
                                                                                          "from the memory-side cache, in a pass where nothing else streams through it"}
                                                                                  if g_ent and g_ent.get("ea_read_latency_clk") else None),
                                                   "how": "aur_microbench: 32 back-to-back launches per projection cycling through the layers (HIP events)"}
            except Exception as ex:                                # the bench line must still print
                result["roofline_prefill_gemm"] = {"error": repr(ex)}
            ideal_s = dec_bytes / 8e12 + B * (pre_fl + vit_fl + proj_fl) / 2.5e15
            result["roofline_step"] = {"ideal_ms": 1e3 * ideal_s, "measured_ms": result["ms_per_step"], "frac": 1e3 * ideal_s / result["ms_per_step"],
                                       "model": "decode: weights once per step + K/V of every cached token, at 8 TB/s; ViT, projector and prefill "
                                                "flops at 2.5 PFLOP/s (SURVEY 8d)"}
            # the stricter bound for a schedule that OVERLAPS the two resources (the default one does): the slower of the HBM-bound decode and
            # the MFMA-bound front ends, not their sum (quoted beside the sum model)
            hbm_s, mfma_s = dec_bytes / 8e12, B * (pre_fl + vit_fl + proj_fl) / 2.5e15
            result["roofline_step_overlap"] = {"ideal_ms": 1e3 * max(hbm_s, mfma_s), "measured_ms": result["ms_per_step"],
                                               "frac": 1e3 * max(hbm_s, mfma_s) / result["ms_per_step"], "hbm_ms": 1e3 * hbm_s, "mfma_ms": 1e3 * mfma_s,
                                               "model": "max(decode bytes at 8 TB/s, front-end flops at 2.5 PFLOP/s): what a schedule that overlaps "
                                                        "the two perfectly would take; the chip cannot hold both peaks at once (1400 W socket cap, DESIGN section 4)"}
        # single-clip latency (batch 1): TTFT without queueing behind other clips' ViT/prefill
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        lat = []
        for _ in range(3):
            e0.record()
            vis = eng.vit_encode(pixels[:F], r)
            eng.begin_batch(1, N, None)
            emb, L = eng.project_splice(vis, ids[0])
            eng.prefill(0, emb, L)
            e1.record()
            torch.cuda.synchronize()
            lat.append(e0.elapsed_time(e1))
        result["ttft_ms_single_clip"] = float(np.median(lat))
        # where a single clip's time to first token goes: one more pass with the stage timers on
        eng.profile(True)
        vis = eng.vit_encode(pixels[:F], r)
        eng.begin_batch(1, N, None)
        emb, L = eng.project_splice(vis, ids[0])
        eng.prefill(0, emb, L)
        torch.cuda.synchronize()
        st1 = {k: eng.profile_read(k)[0] for k in ("vit", "vit_tome", "project", "prefill", "first_token")}
        eng.profile(False)
        # the ToMe launches at the frames per launch of the TIMED loop (one group of G clips per ViT pass), alone on the whole chip
        tome_group_ms = None
        if continuous and G * F <= eng.c.max_frames:
            eng.profile(True)
            eng.vit_encode(pixels[:G * F], r)
            torch.cuda.synchronize()
            tome_group_ms = eng.profile_read("vit_tome")[0]
            eng.profile(False)
        # ---- token merge against its HBM bound (north_star: "evidenced by rocprof HBM GB/s"; SURVEY 8d bytes per frame-layer =
        #      2 (t c + t D + (t - r) D) + 4 (2 t - r): the metric, the state read and the merged state written, the index arrays)
        if not args.tiny:
            hdv_ = v["hidden_size"] // v["num_attention_heads"]
            tb_, tt_ = 0.0, t0tok
            for _ in range(v["num_hidden_layers"] - 1):
                rl_ = min(r, (tt_ - 1) // 2)
                if rl_ > 0:
                    tb_ += 2.0 * (tt_ * hdv_ + tt_ * v["hidden_size"] + (tt_ - rl_) * v["hidden_size"]) + 4.0 * (2 * tt_ - rl_)
                tt_ -= rl_
            tome_bytes_clip = F * tb_
            # the counter file holds one entry per (kernel, grid): the LARGEST grid of each ToMe kernel is the first layer (t = t0) of the pass
            tome_pmc = [max((e for e in pmc_all if k in e["kernel"]), key=lambda e: e.get("grid", 0), default=None)
                        for k in ("tome_prep_kernel", "tome_match_kernel", "tome_select_kernel", "tome_merge_kernel")]
            tome_traffic = sum(e["hbm_bytes_per_launch"] for e in tome_pmc) if all(tome_pmc) else None
            sv_clips = min(VC, B)
            rl0 = min(r, (t0tok - 1) // 2)
            tome_alg_l0 = sv_clips * F * (2.0 * (t0tok * hdv_ + t0tok * v["hidden_size"] + (t0tok - rl0) * v["hidden_size"]) + 4.0 * (2 * t0tok - rl0))
            result["roofline_tome"] = {
                "bound": "hbm", "kernel": "tome_prep / tome_match (v_mfma_f32_16x16x4_f32) / tome_select / tome_merge (+ LayerNorm 2), four launches per layer",
                "achieved": (B * tome_bytes_clip / (tome_ms_batch * 1e-3) / 1e9) if tome_ms_batch > 0 else None, "peak": 8000.0, "unit": "GB/s",
                "frac": (B * tome_bytes_clip / (tome_ms_batch * 1e-3) / 8e12) if tome_ms_batch > 0 else None,
                "traffic": tome_traffic,
                "traffic_context": {"layer": "first (t = %d -> %d)" % (t0tok, t0tok - rl0), "frames_per_launch": sv_clips * F,
                                    "algorithmic_bytes_of_that_layer_launch": tome_alg_l0,
                                    "traffic_over_algorithmic": (tome_traffic / tome_alg_l0) if tome_traffic else None},
                "traffic_note": "sum over the four launches of the FIRST layer of (2*FETCH_SIZE + WRITE_SIZE) KiB in profiles/%s_pmc.json (%d frames per "
                                "launch; counters taken on library build %s, this run loaded %s).  Above the SURVEY 8d bytes by what the formula does not count: "
                                "prep reads the layer's K fragments of ALL heads to form the metric (16x the metric's own bytes) and merge also "
                                "writes the LayerNorm 2 rows" % (pmc_tag, sv_clips * F, pmc_build, lib_sha),
                "algorithmic_bytes_per_clip": tome_bytes_clip, "algorithmic_bytes_per_layer_launch_mean": sv_clips * tome_bytes_clip / max(v["num_hidden_layers"] - 1, 1),
                "ms_per_clip": (tome_ms_batch / B) if tome_ms_batch > 0 else None, "frames_per_launch": sv_clips * F,
                "single_clip": {"ms": st1["vit_tome"], "achieved": tome_bytes_clip / (st1["vit_tome"] * 1e-3) / 1e9 if st1["vit_tome"] > 0 else None,
                                "frac": tome_bytes_clip / (st1["vit_tome"] * 1e-3) / 8e12 if st1["vit_tome"] > 0 else None, "frames_per_launch": F},
                "timed_loop_group": ({"ms": tome_group_ms, "clips": G, "frames_per_launch": G * F, "achieved": G * tome_bytes_clip / (tome_group_ms * 1e-3) / 1e9,
                                     "frac": G * tome_bytes_clip / (tome_group_ms * 1e-3) / 8e12,
                                     "note": "the launch shape of the timed loop (one group of %d clips per ViT pass), measured alone on the whole chip; in the loop it "
                                             "runs on 16 CUs of every XCD beside the decode" % G} if tome_group_ms else None),
                "how": "HIP events around the ToMe launches of every layer (aur_profile 'vit_tome') in the instrumented eager step (%d clips per ViT pass) and in "
                       "one single-clip pass; the merge launch also writes LayerNorm 2's output (not counted in the algorithmic bytes)" % sv_clips}
        result["ttft_stage_ms"] = {
            "single_clip": {"vit_without_tome": st1["vit"] - st1["vit_tome"], "tome": st1["vit_tome"], "projector_splice": st1["project"],
                            "prefill_layers": st1["prefill"] - st1["first_token"], "lm_head_argmax": st1["first_token"],
                            "sum": st1["vit"] + st1["project"] + st1["prefill"],
                            "note": "one clip alone on the whole GPU, stage timers (HIP events) of one extra pass; ToMe = the four launches per layer (the last one also writes LayerNorm 2)"},
            "serving": serving_split}
        hl = []
        for _ in range(3):                                            # the same, on the host clock, including the read-back of the slot state
            torch.cuda.synchronize()
            t_h = time.perf_counter()
            vis = eng.vit_encode(pixels[:F], r)
            eng.begin_batch(1, N, None)
            emb, L = eng.project_splice(vis, ids[0])
            eng.prefill(0, emb, L)
            lens_h, _ = eng.slot_state()
            hl.append(1e3 * (time.perf_counter() - t_h))
            assert int(lens_h[0]) == 1
        result["ttft_host_ms_single_clip"] = float(np.median(hl))

    if continuous and overlap and fe.get("graph") is not None:
        fe["graph"].eng = None                                     # the graph object holds the engine (and with it the 156 GB KV pool): the child runs below need the memory
        fe["graph"] = None
    eng.close()
    del eng
    torch.cuda.empty_cache()
    profiled = any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ)
    if rank == 0 and world == 1 and not args.tiny and not profiled and args.config_name == "cfg2" and B > 1:
        # ---- operating points measured in child processes (each builds its own engine: the kernel structure is a constant of the engine)
        if not args.no_single_stream:
            # the reference's own use: one clip at a time (inference.py:89-96; docs/auroracap/EVAL.md:61 "batch size 1 only")
            c = child_point(["--batch", "1", "--steps", "2", "--warmup", "1", "--no-instrument"])
            ds = (c or {}).get("decode_step") or {}
            result["single_stream"] = ({"error": "the child run produced no line"} if not c or "value" not in c else {
                "captions_per_s": c["value"], "ms_per_caption": c["ms_per_step"], "p50_ttft_ms": c["p50_ttft_ms"],
                "decode_tokens_per_s": ds.get("tokens_per_s"), "decode_ms_per_step": ds.get("ms_per_step"),
                "ideal_tokens_per_s": ds.get("ideal_tokens_per_s"), "frac": ds.get("frac"), "bound": "hbm",
                "how": "`python bench.py --batch 1` in a child process (one KV slot, hipGraph decode, 2 timed captions); frac = (13.21 GB of weights + "
                       "K / V of the mean context) / 8 TB/s over the measured decode step"})
        # the latency-first operating point (groups of 2: half the TTFT for ~6 % of the throughput) is measured in every default run;
        # --frontier adds groups of 1 and 8
        for g_ in ((2,) if not args.no_latency_point else ()) + ((1, 8) if args.frontier else ()):
            if True:
                if g_ == G:
                    continue
                c = child_point(["--prefill-group", str(g_), "--steps", "1", "--warmup", "1", "--no-instrument"])
                if c and "value" in c:
                    result["frontier"].append({"schedule": "default schedule, groups of %d (child process, 1 timed cycle)" % g_, "prefill_group": g_,
                                               "overlap": True, "captions_per_s": c["value"], "p50_ttft_ms": c["p50_ttft_ms"]})
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.tiny:
        try:
            result["cpu_baseline"] = cpu_baseline(cfg, args, n_kept)
        except Exception as ex:                                    # the bench line must still print
            result["cpu_baseline"] = {"value": None, "error": repr(ex)}
    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        dist.barrier()                                             # rank 0 arrives last (instrumented pass): leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
