import torch, sys
sys.path.insert(0, ".")
from tests.test_gpu_front_graph import build, group_clips
eng = build(2, 0)
px, ids = group_clips(1, 2, 3)[0]
want = eng.caption_ids(px, ids, 0.5, 6, eos_id=None)
try:
    eng.graph_capture(lambda: torch.cuda.synchronize())
    print("capture with a device sync: no error?!")
except BaseException as e:
    print("capture with a device sync raised:", type(e).__name__, str(e)[:200])
try:
    torch.cuda.synchronize()
    got = eng.caption_ids(px, ids, 0.5, 6, eos_id=None)
    print("ctx after the broken capture:", "ok" if got == want else "WRONG IDS")
except BaseException as e:
    print("ctx after the broken capture FAILED:", type(e).__name__, str(e)[:300])
