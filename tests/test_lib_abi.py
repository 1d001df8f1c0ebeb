"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/aurora_hip.h declares, the pure-integer schedule entry points agree with the oracle and the
golden table, and the product path refuses to run without a GPU (no silent fallback)."""
import os
import re

import numpy as np
import pytest
import torch

from aurora_amd import _lib
from oracle import aurora_oracle as O
from tests.util import golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "aurora_hip.h")).read()
    return sorted(set(re.findall(r"\b(aur_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    import ctypes
    assert os.path.exists(_lib.SO_PATH), "run `python -m aurora_amd.build` (or __graft_entry__.build())"
    raw = ctypes.CDLL(_lib.SO_PATH)
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/aurora_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes binding and header disagree"
    assert b"gfx950" in _lib.lib().aur_version()


def test_schedule_matches_reference_table():
    L = _lib.lib()
    tab = golden("g1_schedule.npz")["table"]
    for layers, ratio, r, kept in tab:
        layers = int(layers)
        assert L.aur_tome_r(378, 378, 14, float(ratio), layers) == int(r)
        assert L.aur_tokens_at_layer(730, int(r), layers - 1) - 1 == int(kept)
    for t0, r, lay in [(17, 2, 4), (9, 5, 3), (730, 22, 31), (3, 1, 5), (2, 1, 2)]:
        assert L.aur_tokens_at_layer(t0, r, lay) == O.token_schedule(t0, r, lay)[lay]


def test_no_cpu_fallback():
    from aurora_amd.engine import AuroraCapEngine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.AuroraHipError):
        AuroraCapEngine({"vit": None, "llm": None}, {})


def test_product_path_never_imports_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "aurora_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "tome_ref" not in src, f
            elif f.endswith((".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()       # comments may cite the oracle; code may not use it
                assert not re.search(r'#include\s+"[^"]*oracle', src) and "libtome_ref" not in src, f
    for f in ("inference.py",):
        p = os.path.join(ROOT, f)
        if os.path.exists(p):
            assert "oracle" not in open(p).read()


def test_llama_row_map_is_a_permutation_with_rope_pairs():
    from aurora_amd.engine import AuroraCapEngine
    d, heads = 256, 2
    rm = AuroraCapEngine.llama_qkv_row_map(d, heads, 768)
    assert sorted(rm.tolist()) == list(range(768))
    hd = d // heads
    for h in range(heads):
        for p in range(hd // 32):
            a = rm[h * hd + 32 * p: h * hd + 32 * p + 16]
            b = rm[h * hd + 32 * p + 16: h * hd + 32 * p + 32]
            assert (b - a == hd // 2).all() and (a == h * hd + 16 * p + np.arange(16)).all()


def test_loader_brings_torch_hip_runtime_first():
    """`__graft_entry__.build()` loads the library in a process that has not imported torch yet. The loader must make
    torch (and with it torch's libamdhip64) resident first, otherwise the process ends up with two HIP runtimes and the
    first `aur_create` on a GPU box fails with "no ROCm-capable device"."""
    import subprocess
    import sys
    code = ("import sys; from aurora_amd import _lib; assert 'torch' not in sys.modules; _lib.lib(); "
            "assert 'torch' in sys.modules; "
            "maps = open('/proc/self/maps').read(); "
            "hips = {l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}; "
            "assert len(hips) == 1, hips; print(sorted(hips)[0])")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "torch" in out.stdout, out.stdout


def _header_config_fields():
    """[(name, 'int32_t' | 'float'), ...] of struct aur_config, in declaration order, parsed from include/aurora_hip.h"""
    import re
    src = open(os.path.join(ROOT, "include", "aurora_hip.h")).read()
    body = src[src.index("typedef struct aur_config {") + len("typedef struct aur_config {"):src.index("} aur_config;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        ctype, names = decl.split(" ", 1)
        assert ctype in ("int32_t", "float"), decl
        fields += [(n.strip(), ctype) for n in names.split(",")]
    return fields


def test_documented_ctypes_stub_matches_the_header():
    """INTEGRATION.md's `AurConfig` stub is what a maintainer copies: its fields (names, order, widths) and its example
    initialiser must match struct aur_config in include/aurora_hip.h - and so must aurora_amd/_lib.py (the stub had
    fallen one field behind the header, which shifts every later read of aur_create)."""
    import ctypes as C
    import re
    from aurora_amd import _lib
    want = _header_config_fields()
    ct = {"int32_t": C.c_int32, "float": C.c_float}
    assert [(n, t) for n, t in _lib.AurConfig._fields_] == [(n, ct[t]) for n, t in want]
    assert C.sizeof(_lib.AurConfig) == 4 * len(want)
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    start = doc.index("class AurConfig(C.Structure):")
    block = doc[start:doc.index("\n\n", start)]
    ns = {"C": C}
    exec(block, ns)                                                # the documented class itself
    assert [(n, t) for n, t in ns["AurConfig"]._fields_] == [(n, ct[t]) for n, t in want]
    init = re.search(r"cfg = AurConfig\(([^)]*)\)", doc).group(1)
    vals = [v.strip() for v in init.split(",")]
    assert len(vals) == len(want), (len(vals), len(want))
    cfg = ns["AurConfig"](*[float(v) if t == "float" else int(v) for v, (_, t) in zip(vals, want)])
    assert cfg.vit_image == 378 and cfg.llm_vocab == 32000 and cfg.page_tokens == 64 and cfg.spare_slots == 0
