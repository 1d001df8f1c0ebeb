import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, "/root/repo")
import torch
from aurora_amd import synthetic as S
from aurora_amd.engine import AuroraCapEngine, tokens_at_layer
name, frames, ratio, kept = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
cfg = S.AURORACAP_7B
w = {"vit": S.vit_weights(cfg["vit"]), "projector": S.projector_weights(1280, 4096), "llm": S.llm_weights(cfg["llm"])}
L0 = 30 + frames * kept
print("build", flush=True)
eng = AuroraCapEngine(cfg, w, max_frames=2 * frames, max_batch=2, max_ctx=-(-(L0 + 8) // 64) * 64, max_new_tokens=8, spare_slots=1)
del w
torch.cuda.empty_cache()
r = eng.tome_r(ratio)
print("r", r, flush=True)
clips = [(S.frames(frames, 20 + i), S.prompt_ids(frames, 20 + i)) for i in range(2)]
vis = eng.vit_encode(clips[0][0], r)
torch.cuda.synchronize()
print("vit", tuple(vis.shape), flush=True)
alone = [eng.caption_ids(px, ids, ratio, 8, eos_id=None) for px, ids in clips]
print("alone", alone, flush=True)
b = eng.caption_batch(clips, ratio, 8, eos_id=None)
print("batch", b == alone, flush=True)
got = dict(eng.caption_stream(clips + clips[:1], ratio, 8, eos_id=None, check_every=4))
print("stream", [got[i] for i in range(3)] == alone + alone[:1], flush=True)
eng.close()
print("closed", flush=True)
