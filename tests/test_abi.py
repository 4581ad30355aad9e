"""The C-ABI library loads on a CPU-only machine, exports every symbol include/mmdyn_hip.h declares, and the
ctypes signature table agrees with the header (argument count and pointer/int/float kinds)."""
import os
import re

from mmdyn_hip import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_decls():
    txt = open(os.path.join(ROOT, "include", "mmdyn_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(mmdyn_\w+)\s*\(([^)]*)\)\s*;", txt):
        name, args = m.group(1), m.group(2).strip()
        kinds = ""
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    kinds += "p"
                elif a.startswith("int64_t"):
                    kinds += "l"
                elif a.startswith("uint64_t"):
                    kinds += "Q"
                elif a.startswith("float"):
                    kinds += "f"
                elif a.startswith("int"):
                    kinds += "i"
                else:
                    raise AssertionError(f"unparsed argument {a!r} of {name}")
        decls[name] = kinds
    return decls


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    decls = header_decls()
    assert len(decls) >= 38
    for name in decls:
        assert hasattr(lib, name), name
    assert lib.mmdyn_version().startswith(b"mmdyn_hip")


def test_ctypes_signatures_match_header():
    decls = header_decls()
    for name, sig in _lib._SIGNATURES.items():
        assert name in decls, f"{name} not declared in include/mmdyn_hip.h"
        assert decls[name] == sig, (name, decls[name], sig)
    assert set(decls) - {"mmdyn_version"} == set(_lib._SIGNATURES)


def test_argument_errors_are_reported_without_a_gpu():
    lib = _lib.load()
    # null pointers / bad shapes are rejected on the host before any launch
    assert lib.mmdyn_igemm_nt(None, None, None, None, None, None, None, 0, 1, 1, 1, 1, 32, 1, 1, 32, 32, 1, 0, 0, 1, None) == -2
    assert lib.mmdyn_wgrad_chunks(0, 128, 33, 32) == -1
    assert lib.mmdyn_igemm_stat_tiles(2, 4, 256, 8, 8, 128, 16, 16, 64) > 0
