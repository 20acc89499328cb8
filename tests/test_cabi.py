"""The drop-in boundary itself, without a GPU: the shared library loads, exports every function include/flamingo_fusion.h
declares (and the ctypes table binds exactly those), reports its ABI, sizes its workspaces, and argument errors come back
as negative codes + a message before anything touches a device."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "flamingo_fusion.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ff_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_exactly_the_header():
    from flamingo_mini_amd import ffi
    lib = ffi.lib()
    names = declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
    assert sorted(ffi.EXPORTED_SYMBOLS) == names, set(ffi.EXPORTED_SYMBOLS) ^ set(names)
    assert lib.ff_version() == ffi.ABI_VERSION and lib.ff_arch() == b"gfx950"


def test_dynamic_symbol_table_holds_nothing_but_the_header():
    """`nm -D` of the shipped library: every defined global function is one the header declares (the library is built with
    -fvisibility=hidden and linked with csrc/exports.map: no mangled C++ internals, no per-kernel host handles, no helper leaks)."""
    import shutil
    import subprocess
    from flamingo_mini_amd import ffi
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    ffi.lib()
    out = subprocess.run([nm, "-D", "--defined-only", ffi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = sorted(line.split()[-1] for line in out.splitlines() if len(line.split()) >= 3)     # functions AND objects, any binding
    assert defined == declared_functions(), sorted(set(defined) ^ set(declared_functions()))


def _header_prototypes():
    """name -> (return kind, [parameter kinds]) parsed from the header; kinds: 'ptr', 'int', 'size', 'i64', 'float', 'str'."""
    text = open(os.path.join(ROOT, "include", "flamingo_fusion.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)

    def kind(decl: str) -> str:
        decl = decl.strip()
        if "*" in decl or decl.startswith("ff_stream_t"):
            return "str" if decl.replace("const", "").strip().startswith("char") else "ptr"
        base = re.sub(r"\b[a-z_][a-z0-9_]*$", "", decl).strip() or decl      # drop the parameter name
        base = base.replace("const", "").strip()
        return {"int": "int", "size_t": "size", "long long": "i64", "float": "float", "unsigned": "int", "void": "void"}[base]

    protos = {}
    for ret, name, params in re.findall(r"^\s*([a-z_][a-z0-9_ ]*?\**)\s*\b(ff_[a-z0-9_]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.M | re.S):
        params = " ".join(params.split())
        plist = [] if params in ("", "void") else [kind(q) for q in params.split(",")]
        protos[name] = (kind(ret + " x") if "*" in ret else kind(ret + " x"), plist)
    return protos


def test_ctypes_signatures_match_the_header_prototypes():
    """Every binding in ffi._SIGNATURES has the arity and the argument classes (pointer / int / size_t / long long / float) of its C
    prototype: a mismatch here would not fail loudly at run time, it would shift arguments."""
    from flamingo_mini_amd import ffi
    protos = _header_prototypes()
    assert set(protos) == set(ffi._SIGNATURES), set(protos) ^ set(ffi._SIGNATURES)

    def ckind(t):
        if t is None:
            return "void"
        if t is C.c_char_p:
            return "str"
        if t in (C.c_void_p,) or hasattr(t, "contents") or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return "ptr"
        return {C.c_int: "int", C.c_size_t: "size", C.c_longlong: "i64", C.c_float: "float", C.c_uint: "int"}[t]

    for name, (res, args) in ffi._SIGNATURES.items():
        want_ret, want_args = protos[name]
        got_args = [ckind(a) for a in args]
        assert got_args == want_args, (name, got_args, want_args)
        if res is not None:
            assert ckind(res) == want_ret, (name, ckind(res), want_ret)


def test_ctypes_struct_layouts_match_the_header(tmp_path):
    """Field names, order, offsets and sizes of every descriptor struct: the header is compiled (as C, with gcc) into a program that prints
    offsetof / sizeof of each field, and the ctypes mirror in ffi.py must agree byte for byte."""
    import shutil
    import subprocess
    from flamingo_mini_amd import ffi
    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest
        pytest.skip("no gcc")
    text = open(os.path.join(ROOT, "include", "flamingo_fusion.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s+ff_[a-z_]+\s*\{(.*?)\}\s*(ff_[a-z_]+)\s*;", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if decl:
                names = decl.replace("*", " ").split(",")
                fields += [names[0].split()[-1]] + [n.strip() for n in names[1:]]
        structs[name] = fields
    mirror = {"ff_rowmap": ffi.RowMap, "ff_gemm_desc": ffi.GemmDesc, "ff_gemm_profile_record": ffi.GemmProfileRecord, "ff_ln_desc": ffi.LnDesc,
              "ff_reduce_desc": ffi.ReduceDesc, "ff_strides": ffi.Strides, "ff_attn_desc": ffi.AttnDesc, "ff_resampler_desc": ffi.ResamplerDesc,
              "ff_adamw_desc": ffi.AdamWDesc, "ff_kvproj_desc": ffi.KvProjDesc, "ff_xattn_desc": ffi.XattnDesc}
    assert set(structs) == set(mirror), set(structs) ^ set(mirror)
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "flamingo_fusion.h"', "int main(void) {"]
    for sname, fields in structs.items():
        lines.append(f'  printf("{sname} . %zu\\n", sizeof({sname}));')
        for f in fields:
            lines.append(f'  printf("{sname} {f} %zu %zu\\n", offsetof({sname}, {f}), sizeof((({sname}*)0)->{f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([gcc, "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    for line in out.strip().splitlines():
        parts = line.split()
        cls = mirror[parts[0]]
        if parts[1] == ".":
            assert C.sizeof(cls) == int(parts[2]), (parts[0], C.sizeof(cls), parts[2])
            assert [f[0] for f in cls._fields_] == structs[parts[0]], (parts[0], [f[0] for f in cls._fields_], structs[parts[0]])
        else:
            fld = getattr(cls, parts[1])
            assert (fld.offset, fld.size) == (int(parts[2]), int(parts[3])), (parts[0], parts[1], fld.offset, fld.size, parts[2:])


def test_workspace_queries_and_error_codes_without_a_device():
    from flamingo_mini_amd import ffi
    lib = ffi.lib()
    d = ffi.ResamplerDesc(ffi.DTYPE_BF16, 32, 1, 257, 1024, 6, 8, 64, 64, 4, 4, ffi.ACT_GELU)
    saved, scratch = lib.ff_resampler_saved_bytes(d), lib.ff_resampler_scratch_bytes(d)
    assert 0.5e9 < saved < 4e9 and 0.05e9 < scratch < 2e9          # config B: ~1 GB of saved activations
    bad = ffi.ResamplerDesc(ffi.DTYPE_BF16, 32, 5, 257, 1024, 6, 8, 64, 64, 4, 4, ffi.ACT_GELU)   # 5 frames > 4 time embeddings
    assert lib.ff_resampler_saved_bytes(bad) == 0
    assert lib.ff_resampler_fwd(bad, None, None, None, None, 0, None, 0, None) == -1
    assert b"num_time_embeds" in lib.ff_last_error()
    x = ffi.XattnDesc(ffi.DTYPE_BF16, 32, 32, 1280, 1024, 1, 64, 8, 64, 4, ffi.ACT_GELU, 32, 0)
    assert lib.ff_xattn_saved_bytes(x) > 0 and lib.ff_xattn_kv_offset(x) == 0
    assert lib.ff_xattn_block_fwd(x, None, None, None, None, None, None, None, None, 0, None, 0, None) == -1     # null arguments
    g = ffi.GemmDesc(ffi.DTYPE_BF16, 1024, 1280, 5120, 0, 0, ffi.rowmap(5120), ffi.rowmap(5120), ffi.rowmap(1280), 1.0, -1, -1, 0)
    assert lib.ff_gemm_workspace_bytes(g) > 0                                                                      # long K, few tiles -> split-K
    assert lib.ff_gemm(g, None, None, None, None, None, None, None, None, 0, None) == -1
    assert lib.ff_text_time(0, 0, None, 8, None, None) == -1
    a = ffi.AttnDesc(7, 1, 1, 64, 8, 8, 0, 0, 0, 0)                                                                # dtype 7 does not exist
    assert lib.ff_attention_fwd(a, C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), None, C.c_void_p(16), None, None) == -2


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from flamingo_mini_amd import ffi
    monkeypatch.setattr(ffi, "_lib", None)
    monkeypatch.setattr(ffi, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(ffi.FusionLibraryError, match="no CPU/PyTorch fallback"):
        ffi.lib()


def test_media_locations_from_known_ids():
    """flamingo_processor.py:53-61,120-121 of the reference: '<' is 27 / ' <' is 1279 for GPT-2, 51552 / 28696 for OPT."""
    import torch
    from flamingo_mini_amd.flamingo_processor import KNOWN_LEQ_IDS, FlamingoProcessor
    assert KNOWN_LEQ_IDS == {"gpt2": (27, 1279), "facebook/opt": (51552, 28696)}
    ids = torch.tensor([[50256, 27, 9060, 29, 257, 1279, 27], [1, 2, 3, 4, 5, 6, 7]])
    ml = FlamingoProcessor.media_locations_from_ids(ids, KNOWN_LEQ_IDS["gpt2"])
    assert ml.tolist() == [[0, 1, 0, 0, 0, 1, 1], [0] * 7] and ml.dtype == ids.dtype


def test_gemm_plan_for_the_benchmark_shapes():
    """ff_gemm_plan (host only) pins the measured tile / split-K plan of DESIGN.md section 4 for config B's GEMM shapes."""
    import ctypes as C
    from flamingo_mini_amd import ffi
    lib = ffi.lib()

    def plan(M, N, K, al=0, bl=0):
        d = ffi.GemmDesc(ffi.DTYPE_BF16, M, N, K, al, bl, ffi.rowmap(K if al == 0 else M), ffi.rowmap(K if bl == 0 else N), ffi.rowmap(N),
                         1.0, ffi.ACT_NONE, ffi.ACT_NONE, 0)
        bm, bn, sk = C.c_int(), C.c_int(), C.c_int()
        assert lib.ff_gemm_plan(d, bm, bn, sk) == 0
        return bm.value, bn.value, sk.value

    assert plan(1024, 5120, 1280) == (128, 160, 1)           # FFW up-projection: 8 x 32 tiles of 128 x 160 = one per CU (producer / consumer kernel)
    assert plan(1024, 5120, 1280, 0, 1) == (128, 160, 1)     # the same shape with an N-contiguous weight (FFW2 dgrad): the same grid, the B tile staged as 128 + 32 columns
    assert plan(1280, 5120, 1024, 1, 1) == (128, 128, 1)     # its weight gradient (transposed A): 400 tiles of 128^2
    assert plan(1024, 1280, 5120) == (128, 160, 4)           # FFW down-projection: 64 tiles x split-K 4 = 256 workgroups
    assert plan(1024, 1280, 5120, 0, 1) == (128, 160, 4)     # FFW1 dgrad: 64 tiles x split-K 4
    assert plan(1024, 512, 1280) == (64, 64, 2)              # q projection: small, 64^2 tiles + 2 splits
    assert plan(2048, 4096, 4096) == (128, 128, 1)
    assert plan(4096, 16384, 4096) == (256, 256, 1) and plan(4096, 4096, 16384) == (256, 256, 1)            # config E's feed-forward products: 256 x 256 tiles on sixteen waves ...
    assert plan(4096, 16384, 4096, 0, 1) == (256, 128, 1) and plan(4096, 4096, 16384, 0, 1) == (256, 128, 1)  # ... their data gradients (N-contiguous weight): 256 x 128
    assert plan(10272, 1024, 1024) == (256, 128, 1)                                                            # (config B's resampler K / V projection: fewer 256 x 256 tiles than CUs)
    assert plan(16384, 4096, 4096, 1, 1) == (128, 128, 1)   # ... its weight gradients (M-major A): 128 x 128 (the 256 x 128 tile exists for them, measured equal: selectable, not planned)
    assert plan(4096, 512, 4096)[:2] == (128, 128) and plan(4096, 4096, 512) == (128, 128, 1)      # ... its q projection (too narrow) and to_out (short K)
    bm, bn, sk = plan(512, 1024, 10272, 1, 1)                # resampler dWk/dWv: 32 tiles, K = 10272
    assert (bm, bn) == (128, 128) and sk >= 4
    # decode (at most 32 rows)
    assert plan(32, 1280, 512) == (32, 16, 1) and plan(1, 1280, 1024) == (32, 16, 1)      # short K (to_out): weight-streaming kernel
    assert plan(32, 5120, 1280) == (32, 64, 1) and plan(32, 1280, 5120)[:2] == (32, 64)      # long K: 32 x 64 tiles (activation re-reads, see ff_gemm.hip)
    assert plan(32, 1280, 520) == (32, 64, 1)                # K % 32 != 0: the 32 x 64 tiles
    assert plan(33, 5120, 1280)[0] > 32


def test_reference_parameter_counts():
    """The reference's own known-answer pins (examples/model_stats.ipynb:1605): a depth-6 resampler at dim_visual 1024 has
    63 023 104 parameters, one gated cross-attention block at (dim 1280, dim_visual 1024) 15 471 618 - re-derived from the reference
    classes by tests/golden/make_golden.py:param_count_pins."""
    from flamingo_mini_amd import GatedCrossAttentionBlock, PerceiverResampler
    rs = PerceiverResampler(dim=1024, depth=6)
    xa = GatedCrossAttentionBlock(dim=1280, dim_visual=1024)
    assert sum(p.numel() for p in rs.parameters()) == 63023104
    assert sum(p.numel() for p in xa.parameters()) == 15471618
    assert len(rs.fused_params()) == len(list(rs.parameters())) and len(xa.fused_params()) == len(list(xa.parameters()))


def test_sync_status_word_position_follows_the_header():
    """functional._status_word() (where the Python layer's non-blocking probe reads the in-launch hand-offs' error word) must be the word
    the header documents: behind the first two banks of FF_XATTN_SYNC_SLOTS counters; and with no sync buffer allocated the checks are no-ops."""
    import re
    from flamingo_mini_amd import ffi, functional as F
    header = open(os.path.join(ROOT, "include", "flamingo_fusion.h")).read()
    slots = int(re.search(r"#define FF_XATTN_SYNC_SLOTS (\d+)", header).group(1))
    assert int(ffi.lib().ff_xattn_sync_bytes()) == (4 * slots + 64) * 4
    assert F._status_word() == 2 * slots
    if not F._sync_buffers:
        F.check_sync_exchange("cpu")
        F.poll_sync_exchange("cpu")
        assert F.sync_exchange_status() == 0
