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


def test_every_header_struct_has_a_ctypes_mirror_that_starts_with_its_size():
    from vllm_hymeta import _C

    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    declared = sorted(set(re.findall(r"typedef struct (hymeta_\w+) \{", text)))
    assert declared == sorted(_C.STRUCTS), "a struct of the header has no mirror in _C.STRUCTS (or the reverse)"
    for cname, cls in _C.STRUCTS.items():
        assert cls._fields_[0][0] == "struct_size" and cls.struct_size.offset == 0, cname
        assert cls().struct_size == C.sizeof(cls), cname                # filled in on construction
        assert cls(struct_size=7).struct_size == 7, cname               # unless the caller says otherwise


def test_integration_document_struct_equals_the_binding():
    """The ctypes struct INTEGRATION.md tells a maintainer to declare is the one the library reads: same fields,
    same types, same size (a stale document would hand the kernels a short struct)."""
    from vllm_hymeta import _C

    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"^(P, I32, I64, F32, U32 = [^\n]*\n)class GlaDecodeArgs\(C\.Structure\):[^\n]*\n(    _fields_ = \[.*?\)\])\n",
                  doc, flags=re.S | re.M)
    assert m, "INTEGRATION.md no longer holds the GlaDecodeArgs snippet"
    ns = {"C": C}
    exec(m.group(1) + "class GlaDecodeArgs(C.Structure):\n" + m.group(2) + "\n", ns)
    doc_cls = ns["GlaDecodeArgs"]
    assert [(n, t) for n, t in doc_cls._fields_] == [(n, t) for n, t in _C.GlaDecodeArgs._fields_]
    assert C.sizeof(doc_cls) == C.sizeof(_C.GlaDecodeArgs)
    ver = re.search(r"hymeta_abi_version\(\) == (\d+)", doc)
    assert ver and int(ver.group(1)) == _C.ABI_VERSION
    assert f"#define HYMETA_ABI_VERSION {_C.ABI_VERSION}\n" in open(HEADER).read()


def test_wrong_struct_size_is_refused_by_every_entry_point():
    """Each struct-taking entry point compares struct_size with its own sizeof first and returns
    HYMETA_ERR_INVALID_ARG (-1); nothing is launched, so this runs without a GPU."""
    from vllm_hymeta import _C

    lib = _C.load()

    def struct_of(argtype):
        t = getattr(argtype, "_type_", None)
        return t if isinstance(t, type) and issubclass(t, _C._Args) else None

    seen = set()
    for name, (_, argtypes) in _C._SIGNATURES.items():
        if not any(struct_of(a) for a in argtypes):
            continue
        for delta in (-4, 8):
            args = []
            for a in argtypes:
                st = struct_of(a)
                if st is not None:
                    args.append(C.byref(st(struct_size=C.sizeof(st) + delta)))
                    seen.add(st)
                elif a is C.c_void_p:
                    args.append(None)
                else:
                    args.append(a())
            assert getattr(lib, name)(*args) == -1, (name, delta)
    assert seen == set(_C.STRUCTS.values())      # every struct is taken by some entry point and was exercised


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


def test_shapes_the_bf16_entry_point_takes_above_128_rows():
    from vllm_hymeta import _C

    takes = _C.load().hymeta_bf16_gemm_takes
    assert takes(256, 18944, 3584, 0) == 1 and takes(129, 18944, 3584, 0) == 1        # deep, narrow: K slices
    assert takes(256, 3584, 3584, 0) == 1 and takes(256, 3584, 37888, 1) == 1         # K slices; whole tiles with the activation
    assert takes(256, 3584, 9472, 1) == 0 and takes(256, 256, 1024, 0) == 0 and takes(256, 3584, 152064, 2) == 1 and takes(256, 3584, 3584, 2) == 0              # too few workgroups: a library GEMM
    assert takes(128, 3584, 3584, 0) == 1 and takes(257, 18944, 3584, 0) == 0 and takes(64, 96, 64, 0) == 1
