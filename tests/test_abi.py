"""The C-ABI library loads, exports every declared symbol, and the ctypes structs match the header."""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "hymeta_hip.h")


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hymeta_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from vllm_hymeta import _C

    lib = _C.load()
    declared = _declared_functions()
    assert declared, "no functions found in the header"
    assert sorted(_C.exported_symbols()) == declared
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported"
    assert lib.hymeta_abi_version() == _C.ABI_VERSION
    assert lib.hymeta_strerror(0) == b"ok"
    assert b"unsupported" in lib.hymeta_strerror(-2)


def test_ctypes_structs_match_header_layout():
    """Compile the header with gcc and compare sizeof/offsetof with the ctypes mirrors."""
    from vllm_hymeta import _C

    lines = ["#include <stdio.h>", "#include <stddef.h>", f'#include "{HEADER}"', "int main(void){"]
    for cname, cls in _C.STRUCTS.items():
        lines.append(f'printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines.append("return 0;}")
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "abi.c"), os.path.join(td, "abi")
        open(src, "w").write("\n".join(lines))
        subprocess.run(["gcc", "-std=c11", "-o", exe, src], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    got = dict(line.split() for line in out.strip().splitlines())
    for cname, cls in _C.STRUCTS.items():
        assert int(got[cname]) == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    """The ops have no fallback: a missing .so is an error, not a silent CPU path."""
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['HYMETA_HIP_LIB'] = %r\n"
            "from vllm_hymeta import _C\n"
            "try:\n    _C.load()\nexcept RuntimeError as e:\n    print('RAISED', e)\n"
            % (os.path.join(ROOT, "vllm-hymeta_amd"), str(tmp_path / "nope.so")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout
    assert "RAISED" in out and "no non-HIP fallback" in out


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "vllm-hymeta_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), os.path.join(dirpath, f)
