import json
import sys
for l in sys.stdin:
    l = l.strip()
    if not l.startswith("{"):
        if l:
            print("  |", l[:300])
        continue
    d = json.loads(l)
    print("  value %.3f captions/s  ms_per_step %.1f  p50_ttft %.1f ms" % (d["value"], d["ms_per_step"], d.get("p50_ttft_ms") or -1))
