"""Socket power while ONE decode kernel loops (rocm-smi), for energy A/Bs between two builds:
    AURORA_HIP_SO=<path> python tools/gpu/power_microbench.py dec_gateup [dec_qkv ...]"""
import os
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aurora_amd import synthetic as S                     # noqa: E402
from aurora_amd.engine import AuroraCapEngine, _rup       # noqa: E402


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    w = [ln.split(":")[-1].strip() for ln in out.splitlines() if "Power (W)" in ln]
    sc = [ln.split("(")[-1].split(")")[0] for ln in out.splitlines() if "sclk clock level" in ln]
    return (w[0] if w else "?"), (sc[0] if sc else "?")


l = S.VICUNA_7B_16K
B, L0 = 128, 2142
eng = AuroraCapEngine({"vit": None, "llm": l}, {"llm": S.llm_weights(l)}, max_frames=1, max_batch=B, max_ctx=_rup(L0 + 256, 64), max_new_tokens=256)
eng.begin_batch(B, 256, None)
emb0 = (torch.randn(_rup(L0, 32), l["hidden_size"], device="cuda") * 0.02).half()
for b in range(8):
    eng.prefill(b, emb0.clone(), L0)
torch.cuda.synchronize()
from aurora_amd.streams import shared_cu_masked_stream   # noqa: E402
masked = "--masked" in sys.argv
half = "--half-grid" in sys.argv
stream = shared_cu_masked_stream(16, from_top=True) if masked else torch.cuda.current_stream()
eng.set_option("decode_half_grid", 1 if half else 0)


def run(k, n):
    with torch.cuda.stream(stream):
        return eng.microbench(k, n)


for k in [a for a in sys.argv[1:] if not a.startswith("--")]:
    for rep in range(2):
        box = {}
        th = threading.Thread(target=lambda: box.setdefault("us", run(k, 100000)))
        th.start()
        time.sleep(1.5)
        samples = []
        while th.is_alive() and len(samples) < 4:
            samples.append(smi())
            time.sleep(0.7)
        th.join()
        print(os.path.basename(os.environ.get("AURORA_HIP_SO", "default")), "masked" if masked else "all CUs", "half grid" if half else "full grid", k, round(box["us"], 2), "us", samples, flush=True)
eng.close()
