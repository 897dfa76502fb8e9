"""CPU tier: the C-ABI shared library builds for gfx950, loads, exports every symbol the headers under include/
declare, and refuses to run without a GPU (no CPU fallback)."""
import ctypes as C
import glob
import os
import re

import pytest

from libzl_amd import _abi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(header).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    return sorted(set(n for n in names if not n.startswith("__") and n not in ("defined", "sizeof")))


@pytest.fixture(scope="module")
def lib(built):
    return C.CDLL(build.build_engine())


def test_every_declared_symbol_is_exported(lib):
    # the C headers of the boundary (zlhip_voice_adapter.h is a C++ template on top of them, not part of the C-ABI)
    headers = [h for h in sorted(glob.glob(os.path.join(ROOT, "include", "*.h"))) if "template <" not in open(h).read()]
    assert len(headers) == 2
    total = 0
    for h in headers:
        fns = declared_functions(h)
        assert fns, h
        for name in fns:
            assert hasattr(lib, name), f"{os.path.basename(h)} declares {name} but libzlhip.so does not export it"
            total += 1
    assert total >= 30


def test_headers_are_strict_c_and_cpp(tmp_path):
    """the boundary is a C ABI: both headers compile as pedantic C99 / C11 and as C++11 (what a cgo / JNI / ctypes-less host would include)"""
    import subprocess
    inc = os.path.join(ROOT, "include")
    src = tmp_path / "p.c"
    src.write_text('#include "zlhip.h"\n#include "libzl_hotpath.h"\nint main(void) { return zlhip_abi_version() == 0 && libzl_hotpath_status() == 12345; }\n')
    for cmd in (["gcc", "-std=c99"], ["gcc", "-std=c11"], ["g++", "-std=c++11", "-x", "c++"]):
        res = subprocess.run(cmd + ["-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)], capture_output=True, text=True)
        assert res.returncode == 0, " ".join(cmd) + "\n" + res.stderr


def test_binding_table_covers_zlhip_header(lib):
    declared = set(declared_functions(os.path.join(ROOT, "include", "zlhip.h")))
    assert declared == set(_abi.SIGNATURES), declared ^ set(_abi.SIGNATURES)
    _abi.bind(lib)


def test_struct_sizes_match_the_c_layout(lib):
    # sizes the C compiler produces for include/zlhip.h (checked with a tiny compiled probe)
    import subprocess, tempfile, textwrap
    src = textwrap.dedent("""
        #include <stdio.h>
        #include "zlhip.h"
        int main(void) {
            printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(zlhip_config), sizeof(zlhip_clock), sizeof(zlhip_clip_params),
                   sizeof(zlhip_clip_command), sizeof(zlhip_voice_report), sizeof(zlhip_levels), sizeof(zlhip_passthrough_params),
                   sizeof(zlhip_timings), sizeof(zlhip_rt_cycle_trace));
            return 0;
        }""")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        sizes = [int(x) for x in subprocess.check_output([os.path.join(d, "p")]).split()]
    got = [C.sizeof(t) for t in (_abi.Config, _abi.Clock, _abi.ClipParams, _abi.ClipCommand, _abi.VoiceReport, _abi.Levels,
                                 _abi.PassthroughParams, _abi.Timings, _abi.RtCycleTrace)]
    assert got == sizes


def test_defaults_and_plain_helpers_work_without_a_gpu(lib):
    _abi.bind(lib)
    assert lib.zlhip_abi_version() == 3
    cfg = _abi.Config()
    lib.zlhip_config_default(C.byref(cfg))
    assert (cfg.num_buses, cfg.voices_per_bus) == (12, 8)      # SamplerSynth.cpp:23,258
    p = _abi.ClipParams()
    lib.zlhip_clip_params_default(C.byref(p), 2.0)
    assert p.num_slice_positions == 16 and p.slice_positions[8] == 0.5 and p.length_in_beats == -1.0
    assert abs(p.adsr_release - 0.05) < 1e-9 and p.adsr_attack == 0.0 and p.root_note == 60
    c = _abi.ClipCommand()
    lib.zlhip_clip_command_clear(C.byref(c))
    assert (c.clip, c.midi_note, c.midi_channel, c.slice) == (-1, -1, -1, -1)
    assert b"no CPU render path" in lib.zlhip_strerror(_abi.ZLHIP_ERR_NO_DEVICE)


def test_engine_creation_fails_loudly_without_a_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    _abi.bind(lib)
    cfg = _abi.Config()
    lib.zlhip_config_default(C.byref(cfg))
    e = C.c_void_p()
    assert lib.zlhip_engine_create(C.byref(cfg), C.byref(e)) == _abi.ZLHIP_ERR_NO_DEVICE and not e.value
    from libzl_amd import SamplerSynth, ZlHipError
    with pytest.raises(ZlHipError):
        SamplerSynth(2, 8)


def test_product_package_never_touches_the_oracle():
    """The oracle / CPU harness are test infrastructure: nothing under libzl_amd/ or include/ may reference them."""
    bad = []
    for path in glob.glob(os.path.join(ROOT, "libzl_amd", "**", "*"), recursive=True) + glob.glob(os.path.join(ROOT, "include", "*")):
        if os.path.isfile(path) and path.endswith((".py", ".h", ".cpp", ".hip")):
            text = open(path, errors="ignore").read()
            for needle in ("zl_oracle", "oracle/", "np_restatement", "plan_host", "cpu_harness", "zlsim_"):
                if needle in text and not (path.endswith("build.py") and needle in ("oracle/", "cpu_harness", "plan_host", "zl_oracle")):
                    if path.endswith((".h", ".cpp", ".hip")) and needle in ("oracle/", "cpu_harness", "zl_oracle") and "//" in text:
                        # comments may cite the oracle as the checker; code may not include or call it
                        code = "\n".join(l.split("//")[0] for l in text.splitlines())
                        if needle not in code:
                            continue
                    bad.append((os.path.relpath(path, ROOT), needle))
    assert not bad, bad
