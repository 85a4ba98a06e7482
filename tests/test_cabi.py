"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/svk.h declares, and the ctypes structs match the C layouts (no compute calls)."""

import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "svk.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(svk_[a-z0-9_]+)\s*\(", src)))


def _declared_structs():
    src = open(HEADER).read()
    return sorted(set(re.findall(r"typedef struct (Svk\w+)", src)))


def test_library_is_built_and_exports_every_declared_symbol():
    from sparse_vllm_amd import _lib
    lib = _lib.load()
    names = _declared_functions()
    assert names, "no functions parsed from include/svk.h"
    for n in names:
        assert hasattr(lib, n), f"libsvk.so does not export {n}"
        assert n in _lib.ENTRY_POINTS, f"ctypes binding lacks {n}"
    assert set(_lib.ENTRY_POINTS) == set(names)
    assert lib.svk_abi_version() == _lib.SVK_ABI_VERSION


def test_ctypes_struct_layouts_match_the_header(tmp_path):
    from sparse_vllm_amd import _lib
    structs = _declared_structs()
    prog = ["#include <stdio.h>", "#include <stddef.h>", '#include "svk.h"', "int main(void){"]
    for s in structs:
        prog.append(f'printf("{s} %zu\\n", sizeof({s}));')
    prog.append("return 0;}")
    c = tmp_path / "sz.c"
    c.write_text("\n".join(prog))
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    sizes = dict((ln.split()[0], int(ln.split()[1])) for ln in out if ln.strip())
    for s in structs:
        assert hasattr(_lib, s), f"ctypes binding lacks struct {s}"
        assert C.sizeof(getattr(_lib, s)) == sizes[s], f"{s}: ctypes {C.sizeof(getattr(_lib, s))} != C {sizes[s]}"


def test_argument_validation_without_gpu():
    """Bad arguments are rejected before any launch with the reference's exception classes."""
    from sparse_vllm_amd import _lib
    lib = _lib.load()
    a = _lib.SvkH2oSelectArgs(rows=1, kv_len=8, budget=0, recent_count=1)
    with pytest.raises(ValueError, match="H2O budget must be positive"):
        _lib.check(lib.svk_h2o_select_indices(C.byref(a), None), lib)
    s1 = _lib.SvkFlashDecodeStage1Args(head_dim=128, block_seq=24, num_kv_heads=4, num_q_heads=28, batch=1,
                                       max_len_in_batch=8)
    with pytest.raises(AssertionError, match="block_seq"):
        _lib.check(lib.svk_flash_decode_stage1(C.byref(s1), None), lib)
    c = _lib.SvkCompactRowsArgs(keep_len=0, cur_len=4, n_layers=1, n_lanes=1)
    with pytest.raises(RuntimeError, match="empty keep_indices"):
        _lib.check(lib.svk_compact_rows(C.byref(c), None), lib)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from sparse_vllm_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SvkLibraryError, match="no CPU fallback"):
        _lib.load()
