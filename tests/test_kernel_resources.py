"""What the compiler made of the kernels (libzl_amd/lib/libzlhip_kernel_resources.txt, written by libzl_amd/build.py from hipcc's
-Rpass-analysis=kernel-resource-usage remarks), held to what DESIGN.md states.  The guard exists because the resident real-time kernel
once grew 1 KB of scratch memory per lane unnoticed -- the planner and the assembler had stopped being inlined into it, so their objects
and the cycle's ZlBatch lived in memory -- and a cycle went from 26 to 39 us (found late in round 3, profiles/round3_rt_inline_ab.txt)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "libzl_amd", "lib", "libzlhip_kernel_resources.txt")


def _rows(built):
    if not os.path.exists(PATH):
        pytest.skip("no kernel_resources.txt (the library was built by an older build.py)")
    rows = {}
    for line in open(PATH):
        name, *kv = line.split()
        rows[name] = {k: int(v) for k, v in (x.split("=") for x in kv)}
    return rows


def test_resident_kernel_keeps_its_state_in_registers(built):
    rows = _rows(built)
    rt = {n: r for n, r in rows.items() if "zl_k_rt_loop" in n}
    assert len(rt) == 16                                           # 8 modes x narrow / wide
    for n, r in rt.items():
        assert r["scratch"] <= 64, (n, r)                          # (56 bytes: a small local array; 1 KB meant calls and +13 us per cycle)
        # (held to two waves per SIMD -- amdgpu_waves_per_eu(2, 2): the capacity rule counts on two resident workgroups per CU.  Under that
        # budget the one-workgroup-per-bus variants -- the reference's 12 x 8 -- keep everything in registers; the one-workgroup-per-voice
        # variants of wide buses park two to four registers in 12-20 bytes of scratch)
        wide = "ELb1E" in n
        assert r["vgpr_spill"] <= (4 if wide else 0) and r["waves"] == 2, (n, r)


def test_render_kernels_do_not_spill_where_the_design_says_so(built):
    rows = _rows(built)
    k2 = {n: r for n, r in rows.items() if "zl_k2_render" in n}
    assert len(k2) == 48                                           # 8 modes x 3 blocks-per-workgroup x (register gather, LDS-staged)
    headline = next(r for n, r in k2.items() if re.search(r"zl_k2_renderILj0ELi1ELb0E", n))
    assert headline["scratch"] == 0 and headline["vgprs"] <= 96 and headline["waves"] >= 5, headline   # 5 waves per SIMD: the LDS cap of the launch
    hermite = next(r for n, r in k2.items() if re.search(r"zl_k2_renderILj4ELi1ELb0E", n))
    assert hermite["scratch"] == 0 and hermite["waves"] >= 5, hermite
    for n, r in k2.items():
        assert r["scratch"] <= 64, (n, r)                          # (two- and four-blocks-per-workgroup linear: 60 bytes, a measured trade)
    for name in ("zl_k0_apply_ops", "zl_k1c_assemble", "zl_k3_finalize", "zl_k3_scan", "zl_k_reduce_scan", "zl_k_passthrough", "zl_k_deliver"):
        for n, r in rows.items():
            if name in n:
                assert r["scratch"] == 0, (n, r)
