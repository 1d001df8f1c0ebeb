"""What the vendor library reaches on the path's GEMM shapes (torch.matmul -> hipBLASLt / rocBLAS): the yardstick for
gemm256_kernel, never part of the product.  python tools/lib_gemm_probe.py"""
import time

import torch

SHAPES = [("llm qkv", 17152, 12288, 4096), ("llm o", 17152, 4096, 4096), ("llm gate/up", 17152, 22016, 4096),
          ("llm down", 17152, 4096, 11008), ("vit qkv", 81920, 3840, 1280), ("vit out", 81920, 1280, 1280),
          ("vit fc1", 81920, 5120, 1280), ("vit fc2", 81920, 1280, 5120)]


def main():
    dev = torch.device("cuda:0")
    for dt in (torch.float16,):
        for name, m, n, k in SHAPES:
            a = torch.randn(m, k, device=dev, dtype=dt)
            ws = [torch.randn(n, k, device=dev, dtype=dt) * 0.02 for _ in range(4)]
            out = torch.empty(m, n, device=dev, dtype=dt)
            for w in ws:
                torch.matmul(a, w.t(), out=out)
            torch.cuda.synchronize()
            for secs in (0.05, 1.0):                       # a burst, then a sustained run (DVFS)
                n_it = 0
                t0 = time.perf_counter()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                while True:
                    for w in ws:
                        torch.matmul(a, w.t(), out=out)
                    n_it += len(ws)
                    if n_it % 16 == 0:
                        torch.cuda.synchronize()
                        if time.perf_counter() - t0 > secs:
                            break
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / n_it
                print(f"{name:12s} M={m} N={n} K={k} {'burst' if secs < 0.5 else 'sustained'}: {ms * 1e3:8.1f} us "
                      f"{2 * m * n * k / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
