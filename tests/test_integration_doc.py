"""CPU: the binding INTEGRATION.md tells a reference maintainer to write is executed as documented.

The `binding` block is extracted from INTEGRATION.md, pointed at the built library and run; the struct it declares must
have the size AND field offsets `gcc` computes for `SvkFlashDecodeStage1Args` from include/svk.h (a truncated struct - the
round-4 finding - would make the library read `new_k` / `direct_o` from whatever follows it on the caller's stack); its
ABI check must accept this library and refuse another version.  The documented call example and the "fused store" snippet
are executed on CPU tensors against a recording stand-in for the entry point, so a field name that does not exist in the
struct is an error here and not on a maintainer's machine."""

import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOC = os.path.join(ROOT, "INTEGRATION.md")
HEADER = os.path.join(ROOT, "include", "svk.h")
LIB = os.path.join(ROOT, "sparse_vllm_amd", "libsvk.so")


def _block(name: str) -> str:
    text = open(DOC).read()
    m = re.search(rf"<!-- {name}:begin -->\s*```python\n(.*?)```\s*<!-- {name}:end -->", text, flags=re.S)
    assert m, f"INTEGRATION.md has no `{name}` block"
    return m.group(1)


def _header_struct_fields(name: str):
    src = open(HEADER).read()
    body = re.search(rf"typedef struct {name} \{{(.*?)\}} {name};", src, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = [n.strip().lstrip("*") for n in decl.split(",")]
        names[0] = names[0].split()[-1].lstrip("*")
        fields.extend(names)
    return fields


@pytest.fixture(scope="module")
def binding():
    src = _block("binding").replace("/path/to/libsvk.so", LIB)
    ns = {}
    exec(compile(src, "INTEGRATION.md:binding", "exec"), ns)
    return ns


def test_documented_struct_is_the_whole_header_struct(binding, tmp_path):
    S = binding["SvkFlashDecodeStage1Args"]
    fields = _header_struct_fields("SvkFlashDecodeStage1Args")
    assert [n for n, _ in S._fields_] == fields, "INTEGRATION.md struct differs from include/svk.h (names / order)"
    prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "svk.h"', "int main(void){",
            'printf("%zu\\n", sizeof(SvkFlashDecodeStage1Args));']
    prog += [f'printf("%zu\\n", offsetof(SvkFlashDecodeStage1Args, {f}));' for f in fields]
    prog.append("return 0;}")
    c = tmp_path / "off.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "off"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = [int(x) for x in subprocess.check_output([str(exe)]).decode().split()]
    assert C.sizeof(S) == out[0]
    assert [getattr(S, f).offset for f in fields] == out[1:]
    # and it is the struct the product binds
    from sparse_vllm_amd import _lib
    assert C.sizeof(S) == C.sizeof(_lib.SvkFlashDecodeStage1Args)
    assert [n for n, _ in S._fields_] == [n for n, _ in _lib.SvkFlashDecodeStage1Args._fields_]


def test_documented_binding_checks_the_abi_version(binding):
    from sparse_vllm_amd import _lib
    header_version = int(re.search(r"#define SVK_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert binding["SVK_ABI_VERSION"] == header_version == _lib.SVK_ABI_VERSION == binding["_lib"].svk_abi_version()
    stale = _block("binding").replace("/path/to/libsvk.so", LIB).replace(f"SVK_ABI_VERSION = {header_version}",
                                                                        f"SVK_ABI_VERSION = {header_version - 1}")
    with pytest.raises(ImportError, match="ABI"):
        exec(compile(stale, "INTEGRATION.md:binding(stale)", "exec"), {})


def test_documented_error_mapping(binding):
    """`_check` maps the library's return codes to the reference's exception classes with the library's message."""
    lib, S = binding["_lib"], binding["SvkFlashDecodeStage1Args"]
    a = S(head_dim=128, block_seq=24, num_kv_heads=4, num_q_heads=28, batch=1, max_len_in_batch=8)
    with pytest.raises(AssertionError, match="block_seq"):
        binding["_check"](lib.svk_flash_decode_stage1(C.byref(a), None))
    binding["_check"](0)


def test_documented_call_example_and_fused_store_snippet_run(binding):
    torch = pytest.importorskip("torch")
    ns = dict(binding)
    calls = []

    class Recorder:
        def svk_flash_decode_stage1(self, ref, stream):
            a = ref._obj
            calls.append({n: getattr(a, n) for n, _ in a._fields_})
            return 0

    ns["_lib"] = Recorder()
    ns["_stream"] = lambda: None
    example = _block("example")
    fused = _block("fused-store")
    # the snippet continues the example's function body ("fill five more fields" before the launch)
    head, launch = example.split("    _check(_lib.svk_flash_decode_stage1", 1)
    launch = "    _check(_lib.svk_flash_decode_stage1" + launch.split("\n\ndef _stream", 1)[0]
    src = (head.replace("block_seq):", "block_seq, k_new=None, v_new=None, slot_mapping=None):").replace(
        "def gqa_flash_decode_stage1_with_score(", "def documented(")
           + "    if k_new is not None:\n        k, v = k_new, v_new\n"
           + "\n".join("    " + ln if ln.strip() else ln for ln in fused.splitlines()) + "\n" + launch + "\n")
    exec(compile(example, "INTEGRATION.md:example", "exec"), ns)         # as written
    ns["_stream"] = lambda: None
    exec(compile(src, "INTEGRATION.md:example+fused-store", "exec"), ns)
    B, Hq, Hkv, D, W = 2, 28, 4, 128, 96
    q = torch.zeros((B, Hq, D), dtype=torch.bfloat16)
    k = torch.zeros((64, Hkv, D), dtype=torch.bfloat16)
    tab = torch.zeros((3, W), dtype=torch.int32)
    rows = torch.arange(B, dtype=torch.int32)
    lens = torch.full((B,), 40, dtype=torch.int32)
    mid = torch.zeros((B, Hq, 2, D))
    lse = torch.zeros((B, Hq, 2))
    score = torch.zeros((B, W))
    ns["gqa_flash_decode_stage1_with_score"](q, k, k, tab, rows, lens, 40, mid, lse, score, 32)
    plain = calls[-1]
    assert plain["q"] == q.data_ptr() and plain["score_mode"] == 2 and plain["block_seq"] == 32 and plain["kv_num_slots"] == 64
    assert plain["new_k"] is None and plain["direct_o"] is None and plain["score_overwrite"] == 0 and plain["slot_page_size"] == 0
    kn = torch.zeros((B, Hkv, D), dtype=torch.bfloat16)
    sm = torch.zeros((B,), dtype=torch.int32)
    ns["documented"](q, k, k, tab, rows, lens, 40, mid, lse, score, 32, k_new=kn, v_new=kn, slot_mapping=sm)
    fusedc = calls[-1]
    assert fusedc["new_k"] == kn.data_ptr() and fusedc["slot_mapping"] == sm.data_ptr()
    assert (fusedc["new_stride_b"], fusedc["new_stride_h"]) == (kn.stride(0), kn.stride(1))
