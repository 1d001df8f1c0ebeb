"""Static checks of the compiled gfx950 code (CPU: hipcc cross-compiles; no GPU needed).

Round 3's fused split-K reduce wrote its partial tiles with inline asm (`global_store_dwordx4 .. sc1`) and was caught producing
different captions in one serving run of three.  Root cause (profiles/r04_fused_reduce_rootcause.txt): hipcc does not pad hazards
inside an asm statement, and the compiled code rewrote the store's DATA registers one wait state after the store.  The rule these
tests pin: in the product library every store of more than 64 bits is an instruction the COMPILER emitted (it pads and counts those);
no inline-asm store of more than 64 bits lives in the product sources."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "aurora_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

WIDE_STORE = re.compile(r"\b(global|buffer|flat|scratch)_store_(dwordx3|dwordx4|b96|b128)\b")


def _asm_blocks(asm_text):
    """[(first line number, [instructions])] of every ;;#ASMSTART .. ;;#ASMEND block"""
    blocks, cur, start = [], None, 0
    for i, ln in enumerate(asm_text.splitlines(), 1):
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            cur, start = [], i
        elif t.startswith(";;#ASMEND"):
            if cur is not None:
                blocks.append((start, cur))
            cur = None
        elif cur is not None and t and not t.startswith(";"):
            cur.append(t)
    return blocks


def hazards(asm_text):
    """inline-asm blocks that hold a > 64-bit VMEM store not followed (inside the block) by >= 2 wait states"""
    bad = []
    for start, ins in _asm_blocks(asm_text):
        for k, t in enumerate(ins):
            if not WIDE_STORE.search(t):
                continue
            states = 0
            for u in ins[k + 1:]:
                m = re.match(r"s_nop\s+(\d+)", u)
                states += int(m.group(1)) + 1 if m else 1
            if states < 2:
                bad.append((start, t))
    return bad


def test_the_checker_flags_the_round3_form_and_accepts_the_padded_one():
    r03 = ";;#ASMSTART\n\tglobal_store_dwordx4 v[18:19], v[10:13], off sc1\n;;#ASMEND\n\ts_mov_b64 s[0:1], 0x400\n\tv_lshl_add_u64 v[10:11], v[18:19], 0, s[0:1]\n"
    ok = ";;#ASMSTART\n\tglobal_store_dwordx4 v[18:19], v[10:13], off sc1\n\ts_nop 1\n;;#ASMEND\n"
    assert len(hazards(r03)) == 1
    assert hazards(ok) == []


def test_product_sources_hold_no_inline_asm_wide_store():
    pat = re.compile(r"asm\s+volatile\s*\(\s*\"[^;]*?_store_(dwordx3|dwordx4|b96|b128)")
    for f in sorted(os.listdir(CSRC)):
        src = open(os.path.join(CSRC, f)).read()
        assert "AUR_LABS" not in src, f"{f}: lab-only code in a product source (round 6 removed the lab build)"
        assert not pat.search(src), f"{f}: an inline-asm store of more than 64 bits in product code (use __builtin_amdgcn_raw_buffer_store_b128, or end the string with s_nop 1)"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_compiled_decode_kernels_have_no_unpadded_inline_asm_store(tmp_path):
    """decode.hip holds the cross-workgroup split-K hand-over: compile it as the product build does and scan the gfx950 assembly"""
    out = tmp_path / "decode.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "decode.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    assert hazards(asm) == []
    # the hand-over's partial stores are there, as compiler-emitted write-through buffer stores
    assert re.search(r"buffer_store_dwordx4 .* sc1", asm)
    assert not re.search(r"global_store_dwordx4 .* sc1", asm)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_the_front_end_gemm_keeps_its_accumulators_out_of_scratch(tmp_path):
    """A regression caught inside round 4 before it shipped: a third activation branch in the wide epilogue pushed its `#pragma unroll`
    loop over the 8 row blocks past LLVM's pragma-unroll threshold; the loop stayed rolled, `acc[t][u]` became a runtime index and
    the 128 accumulators went to scratch (`.private_segment_fixed_size 528`, ViT fc1 938 -> 590 TF/s).  The row block is a template
    parameter now (gemm256.hip g2_epilogue_row_u); no kernel of the file may own scratch or spill, whatever the epilogue grows to."""
    out = tmp_path / "gemm256.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-o", str(out), os.path.join(CSRC, "gemm256.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asm = out.read_text()
    sizes = dict(re.findall(r"\.name:\s*(_Z14gemm256_kernel\w+)\n(?:.*\n)*?\s*\.private_segment_fixed_size:\s*(\d+)", asm))
    sizes = {k: int(v) for k, v in sizes.items()}
    # a few scalars parked across the K loop (spilled before it, reloaded in the epilogue) are tolerated; an accumulator array (512 B) is not.
    # The generic kernel (GT_OTHER: two epilogue copies, the lean one and the run-time activation ladder) parks a few more.
    generic = "_Z14gemm256_kernelILi0ELi0EEv8GemmArgs"
    assert len(sizes) >= 10 and all(v <= (128 if k == generic else 64) for k, v in sizes.items()), sizes
    # ... and nothing touches scratch between a kernel's first and last MFMA (the K loop)
    kernels = re.split(r"^(_Z14gemm256_kernel\w+):.*$", asm, flags=re.M)
    assert len(kernels) >= 21
    for name, body in zip(kernels[1::2], kernels[2::2]):
        lines = body.splitlines()
        mf = [i for i, ln in enumerate(lines) if "v_mfma" in ln]
        assert mf, name
        assert not any("scratch_" in ln for ln in lines[mf[0]:mf[-1] + 1]), f"{name}: scratch access inside the K loop"
        assert sum("scratch_" in ln for ln in lines) <= (40 if name == generic else 16), f"{name}: more than a few parked scalars go through scratch"
