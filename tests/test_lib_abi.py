"""CPU-only checks of the C-ABI boundary: libdosx.so loads, exports every symbol include/dosx.h
declares, and the ctypes mirrors of its structs have the C layout (checked with a gcc-compiled
probe of the real header).  No compute calls (there is no GPU here)."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "dosx.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dosx_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from dostransformer_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dosx.h but not exported by libdosx.so"
    assert set(names) == set(_lib.EXPORTS), set(names) ^ set(_lib.EXPORTS)
    assert lib.dosx_version() >= 100


def test_ctypes_structs_match_c_layout(tmp_path):
    from dostransformer_amd import _lib
    probe = tmp_path / "probe.c"
    probe.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "dosx.h"\n'
        'int main(void){\n'
        ' printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(DosxRowMap), sizeof(DosxSeg), sizeof(DosxGemm), sizeof(DosxWgrad),'
        ' sizeof(DosxReduceJob), sizeof(DosxAttn), sizeof(DosxFfn), sizeof(DosxCall));\n'
        ' printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", offsetof(DosxGemm, w), offsetof(DosxGemm, out_map), offsetof(DosxGemm, partials),'
        ' offsetof(DosxWgrad, slab), offsetof(DosxAttn, x), offsetof(DosxAttn, partials_kv), offsetof(DosxFfn, out),'
        ' offsetof(DosxCall, iarg), offsetof(DosxCall, farg));\n return 0; }\n')
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(probe), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    sizes = [int(x) for x in out[:8]]
    offs = [int(x) for x in out[8:]]
    assert sizes == [C.sizeof(_lib.RowMap), C.sizeof(_lib.Seg), C.sizeof(_lib.Gemm), C.sizeof(_lib.Wgrad),
                     C.sizeof(_lib.ReduceJob), C.sizeof(_lib.Attn), C.sizeof(_lib.Ffn), C.sizeof(_lib.Call)]
    assert offs == [_lib.Gemm.w.offset, _lib.Gemm.out_map.offset, _lib.Gemm.partials.offset, _lib.Wgrad.slab.offset,
                    _lib.Attn.x.offset, _lib.Attn.partials_kv.offset, _lib.Ffn.out.offset, _lib.Call.iarg.offset,
                    _lib.Call.farg.offset]


def test_argument_validation_needs_no_gpu():
    """Entry points validate descriptors before touching the device and report through dosx_last_error."""
    from dostransformer_amd import _lib
    lib = _lib.load()
    g = _lib.Gemm()
    g.M, g.N, g.K, g.nseg = 4, 8, 0, 1
    assert lib.dosx_gemm(C.byref(g), None) != 0
    assert b"K=0" in lib.dosx_last_error()
    assert lib.dosx_gemm(None, None) != 0
    assert lib.dosx_wgrad_splits(9000, 256, 384) >= 1
    # one partial row per workgroup: 16-row tiles for small grids, 48-row tiles for M = 9000 (gemm_rt in csrc/gemm.hip)
    assert lib.dosx_gemm_partial_rows(100, 256, 2) == 7 and lib.dosx_gemm_partial_rows(100, 256, 5) == 14
    assert lib.dosx_gemm_partial_rows(9000, 256, 2) == 188 and lib.dosx_gemm_partial_rows(13056, 128, 4) == 204
    assert lib.dosx_ffn_bwd_partial_rows(3264) == 204 and lib.dosx_ffn_bwd_partial_rows(6528) == 204


def test_product_path_fails_loudly_without_gpu():
    import torch
    from dostransformer_amd import synth
    from dostransformer_amd.embedder_phDOS.DOSTransformer_phonon import DOSTransformer_phonon
    from dostransformer_amd.layers import TransformerEncoder
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    model = DOSTransformer_phonon(3, 1, 118, 4, 16, "cpu", 0.0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(synth.phonon_batch(2, seed=0, dtype=torch.float32))
    enc = TransformerEncoder(16, 1, 1)
    x = torch.zeros(3, 2, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        enc(x, x, x)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dostransformer_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py") and f != "smoke.py":       # smoke.py is __graft_entry__.smoke()'s body (checker allowed)
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, flags=re.M), (dp, f)


def test_replay_thunks_cover_the_header_and_are_typed():
    """dosx_replay dispatches through thunks GENERATED from include/dosx.h (tools/gen_replay_thunks.py): every `int
    dosx_*` entry point that fits a DosxCall has an op, with the argument-class counts of its prototype; the committed
    replay_thunks.inc is what the generator produces from the committed header."""
    from dostransformer_amd import _lib
    lib = _lib.load()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_replay_thunks as gen
    protos = list(gen.prototypes(open(HEADER).read()))
    assert len(protos) >= 40
    seen = set()
    for name, params in protos:
        ni = sum(1 for t in params if t not in gen.FLOAT_TYPES)
        nf = len(params) - ni
        a, b = C.c_int(-1), C.c_int(-1)
        op = lib.dosx_replay_op(name.encode(), C.byref(a), C.byref(b))
        if ni > 19 or nf > 6:
            assert op == -1, name                       # does not fit a DosxCall: not replayable
            continue
        assert op >= 0 and (a.value, b.value) == (ni, nf), (name, op, a.value, b.value, ni, nf)
        assert op not in seen
        seen.add(op)
    for name, n in (("hipEventRecord", 2), ("hipStreamWaitEvent", 3)):
        a = C.c_int(-1)
        assert lib.dosx_replay_op(name.encode(), C.byref(a), None) >= 1000 and a.value == n
    assert lib.dosx_replay_op(b"no_such_entry", None, None) == -1
    # a call whose argument counts do not match its entry point is rejected before anything is launched
    c = _lib.Call()
    c.op, c.nint, c.nflt = lib.dosx_replay_op(b"dosx_fill", None, None), 2, 1
    failed = C.c_int(-1)
    assert lib.dosx_replay((_lib.Call * 1)(c), 1, C.byref(failed)) != 0 and failed.value == 0
    assert b"dosx_fill" in lib.dosx_last_error()
    out = os.path.join(ROOT, "dostransformer_amd", "csrc", "replay_thunks.inc")
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        sys.argv, argv = ["gen", HEADER, os.path.join(d, "t.inc")], sys.argv
        try:
            gen.main()
        finally:
            sys.argv = argv
        assert open(os.path.join(d, "t.inc")).read() == open(out).read()


def test_gemm_kernel_name_matches_the_tile_choice():
    from dostransformer_amd import _lib
    lib = _lib.load()
    g = _lib.Gemm()
    g.M, g.N, g.K, g.nseg = 9000, 256, 384, 1
    g.w_layout, g.pro, g.epi = 0, 0, 1
    g.a[0].ld, g.a[0].width = 384, 384
    g.ldw = 384
    buf = C.create_string_buffer(96)
    assert lib.dosx_gemm_kernel_name(C.byref(g), buf, 96) == 0
    assert buf.value == b"gemm_kernel<3, 2, 0, 0, 1, 1>"       # 48-row tiles, 256 columns, LN epilogue (edge GEMM1 at cfg2)
