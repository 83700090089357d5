"""CPU: the C-ABI library loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

from conftest import ROOT

import recsys_benchmark_amd as pkg
from recsys_benchmark_amd import _lib

DECL = re.compile(r"MI_API\s+(?:const\s+)?[\w\s\*]+?\b(mi_\w+)\s*\(([^;]*?)\)\s*;", re.S)


def declared():
    out = {}
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        for name, args in DECL.findall(open(h).read()):
            args = re.sub(r"/\*.*?\*/", "", args, flags=re.S).strip()
            n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
            out[name] = n
    return out


PROTO = re.compile(r"MI_API\s+((?:const\s+)?[\w\s\*]+?)\b(mi_\w+)\s*\(([^;]*?)\)\s*;", re.S)


def prototypes():
    """name -> normalised prototype text (return type + parameter TYPES, names and comments dropped)."""
    out = {}
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        for ret, name, args in PROTO.findall(open(h).read()):
            args = re.sub(r"/\*.*?\*/", "", args, flags=re.S)
            norm = []
            for a in args.split(","):
                a = " ".join(a.split())
                a = re.sub(r"\b\w+$", "", a).strip() if not a.endswith("*") and a not in ("void", "") else a
                norm.append(a.replace(" *", "*").replace("* ", "*"))
            out[name] = " ".join(ret.split()) + "(" + ",".join(norm) + ")"
    return out


def test_changing_an_existing_prototype_requires_an_abi_version_bump():
    """tests/golden/abi_prototypes.json is the prototype set of the ABI version it names.  ADDING entry points is free;
    an existing prototype that differs from the snapshot must come with a higher MI_ABI_VERSION (and a refreshed
    snapshot): a C or ctypes caller built against the old header would otherwise misbind its arguments silently."""
    import json

    snap = json.load(open(os.path.join(ROOT, "tests", "golden", "abi_prototypes.json")))
    header = open(os.path.join(ROOT, "include", "mi355x_recsys.h")).read()
    version = int(re.search(r"#define MI_ABI_VERSION (\d+)", header).group(1))
    now = prototypes()
    changed = [n for n, p in snap["prototypes"].items() if n in now and now[n] != p]
    removed = [n for n in snap["prototypes"] if n not in now]
    if changed or removed:
        assert version > snap["abi_version"], f"prototypes changed without a version bump: {changed + removed}"
    else:
        assert version == snap["abi_version"], "MI_ABI_VERSION moved: refresh tests/golden/abi_prototypes.json"


def test_header_declares_entry_points():
    d = declared()
    assert "mi_gather_fm_fwd" in d and d["mi_gather_fm_fwd"] == 14
    assert len(d) >= 10


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared():
        assert hasattr(lib, name), f"{name} declared in include/ but not exported"


def test_binding_table_matches_header():
    d = declared()
    assert set(d) == set(_lib.SIGNATURES), set(d) ^ set(_lib.SIGNATURES)
    for name, n in d.items():
        assert len(_lib.SIGNATURES[name]) == n, name


def test_load_and_version():
    lib = _lib.load()
    assert lib.mi_abi_version() == _lib.ABI_VERSION == 3
    assert lib.mi_strerror(0) == b"ok"
    assert b"invalid" in lib.mi_strerror(-1)


def test_product_fails_loudly_without_gpu_tensor():
    import pytest
    import torch

    m = pkg.DeepFM([3, 4], 4, [8])
    with pytest.raises(pkg.MI355XLibraryError):
        m(torch.tensor([[0, 1]]))


def test_header_compiles_as_plain_c_and_struct_layouts_match_the_bindings(tmp_path):
    """include/mi355x_recsys.h is a C header (gcc -std=c99 takes it as is), and the one struct that crosses the boundary by
    pointer — mi_gemm_problem — has the size and field offsets the ctypes mirror in _kernels.py assumes."""
    import shutil
    import subprocess

    from recsys_benchmark_amd import _kernels

    gcc = shutil.which("gcc")
    if gcc is None:
        import pytest

        pytest.skip("no gcc in this environment")
    from recsys_benchmark_amd import tail as _tailmod

    for cname, ct in (("mi_tail_bn_fwd", _tailmod._BnFwd), ("mi_tail_bn_bwd", _tailmod._BnBwd),
                      ("mi_tail_mask_ride", _tailmod._MaskRide)):
        src2 = tmp_path / f"{cname}.c"
        lines2 = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{os.path.join(ROOT, "include", "mi355x_recsys.h")}"',
                  'int main(void) {', f'  printf("%zu\\n", sizeof({cname}));']
        lines2 += [f'  printf("%zu\\n", offsetof({cname}, {f}));' for f, _ in ct._fields_]
        lines2 += ['  return 0;', '}']
        src2.write_text("\n".join(lines2))
        exe2 = tmp_path / cname
        subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", str(src2), "-o", str(exe2)])
        got = [int(v) for v in subprocess.check_output([str(exe2)]).split()]
        assert got[0] == ctypes.sizeof(ct), cname
        assert got[1:] == [getattr(ct, f).offset for f, _ in ct._fields_], cname
    fields = [name for name, _ in _kernels._GemmProblem._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{os.path.join(ROOT, "include", "mi355x_recsys.h")}"',
             'int main(void) {', '  printf("%zu\\n", sizeof(mi_gemm_problem));']
    lines += [f'  printf("%zu\\n", offsetof(mi_gemm_problem, {f}));' for f in fields]
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call([gcc, "-std=c99", "-Wall", "-Werror", "-o", str(exe), str(src)])
    out = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert out[0] == ctypes.sizeof(_kernels._GemmProblem)
    for f, off in zip(fields, out[1:]):
        assert getattr(_kernels._GemmProblem, f).offset == off, f


def test_binding_argument_types_match_header():
    """Every argument of every entry point has the ctypes kind its C declaration asks for (pointer / int32 / int64 / float /
    double) — a drifted width would pass the arity check and corrupt the call."""
    sizes = {"ptr": ctypes.sizeof(ctypes.c_void_p)}

    def c_kind(arg: str) -> str:
        arg = re.sub(r"/\*.*?\*/", "", arg, flags=re.S).strip()
        if "*" in arg:
            return "ptr"
        for key, kind in (("int64_t", "i64"), ("int32_t", "i32"), ("double", "f64"), ("float", "f32"), ("int ", "i32")):
            if key in arg + " ":
                return kind
        raise AssertionError(f"unrecognised C argument: {arg!r}")

    def py_kind(t) -> str:
        if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents") or getattr(t, "_type_", None) == "P":
            return "ptr"
        if isinstance(t, type) and issubclass(t, ctypes._Pointer):
            return "ptr"
        return {ctypes.c_int32: "i32", ctypes.c_int64: "i64", ctypes.c_float: "f32", ctypes.c_double: "f64",
                ctypes.c_int: "i32", ctypes.c_longlong: "i64"}[t]

    text = ""
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        text += open(h).read()
    checked = 0
    for name, args in DECL.findall(text):
        args = re.sub(r"/\*.*?\*/", "", args, flags=re.S).strip()
        want = [] if args in ("", "void") else [c_kind(a) for a in args.split(",") if a.strip()]
        got = [py_kind(t) for t in _lib.SIGNATURES[name]]
        assert got == want, f"{name}: bindings {got} vs header {want}"
        checked += len(want)
    assert checked > 500 and sizes["ptr"] == 8
