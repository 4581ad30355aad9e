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


def test_split_arithmetic_host_queries_without_a_gpu():
    """Flag bit 7 ("this fp32 launch may take the three-term split") at the shape queries: host arithmetic only.
    The weight-gradient cut aims at 512 blocks in flight instead of 768 (the split kernels' LDS planes: two 128x128 blocks per CU);
    the partial-sum tile count of a launch follows the kernel the flag selects, and a launch the split does not serve answers as
    the plain fp32 query does."""
    lib = _lib.load()
    CONV, TCONV_S2P1, DENSE = 1, 2, 0
    # conv 128 -> 256 channels on 1024 x 5x5 rows: 2 x 1 tiles x 16 taps = 32 tiles -> 24 slabs natively, 16 with the split
    assert lib.mmdyn_wgrad_chunks_mx(CONV, 1024 * 25, 256, 128, 0) == 24
    assert lib.mmdyn_wgrad_chunks_mx(CONV, 1024 * 25, 256, 128, 128) == 16
    assert lib.mmdyn_wgrad_chunks_mx(CONV, 1024 * 25, 256, 128, 7) == lib.mmdyn_wgrad_chunks_mx(CONV, 1024 * 25, 256, 128, 6)
    # the persistent kernel keeps its launches (and its tile count) in the split arithmetic
    native = lib.mmdyn_igemm_stat_tiles(CONV, 4, 256, 16, 16, 64, 8, 8, 128)
    assert lib.mmdyn_igemm_stat_tiles_mx(CONV, 4, 256, 16, 16, 64, 8, 8, 128, 128) == native
    assert lib.mmdyn_igemm_slab_floats_mx(CONV, 4, 256, 16, 16, 64, 8, 8, 128, 128) == lib.mmdyn_igemm_slab_floats(CONV, 4, 256, 16, 16, 64, 8, 8, 128)
    # a large N = 64 launch: the register-staged split kernel on 128x64 tiles (one partial-sum tile per 128 rows and class), no slabs
    assert lib.mmdyn_igemm_stat_tiles_mx(TCONV_S2P1, 4, 256, 8, 8, 128, 16, 16, 64, 128) == 4 * (256 * 64 // 128)
    assert lib.mmdyn_igemm_slab_floats_mx(TCONV_S2P1, 4, 256, 8, 8, 128, 16, 16, 64, 128) == 0
    # too small for the split's launch rule: the plain answer
    assert lib.mmdyn_igemm_stat_tiles_mx(DENSE, 1, 256, 1, 1, 512, 1, 1, 512, 128) == lib.mmdyn_igemm_stat_tiles(DENSE, 1, 256, 1, 1, 512, 1, 1, 512)


def test_plane_operand_host_queries_without_a_gpu():
    """Round 5: operands that arrive split (flag bits 7 + 8 / 9).  Host arithmetic only: which launches a plane kernel serves, the
    partial-sum tile count and slab workspace of such a launch, the cut of the plane weight gradient, the ABI version."""
    lib = _lib.load()
    CONV, TCONV_S2P1, TCONV_S1P0, DENSE = 1, 2, 4, 0
    assert lib.mmdyn_abi_version() == _lib.ABI_VERSION == 6
    # convolution-level launches of the bs-256 step: N % 128 == 0 and N == 64 on the plane-ring kernel, the one-group k4 s1 p0 launch too
    assert lib.mmdyn_igemm_planes_served(CONV, 4, 256, 16, 16, 64, 8, 8, 128) == 1
    assert lib.mmdyn_igemm_planes_served(CONV, 4, 256, 32, 32, 32, 16, 16, 64) == 1
    assert lib.mmdyn_igemm_planes_served(TCONV_S1P0, 1, 256, 5, 5, 256, 8, 8, 128) == 1
    # the 32-channel up-sampling layers: the patch-resident kernel's plane form, at any batch size; one partial-sum tile per image
    assert lib.mmdyn_igemm_planes_served(TCONV_S2P1, 4, 2, 16, 16, 64, 32, 32, 32) == 1
    assert lib.mmdyn_igemm_stat_tiles_mx(TCONV_S2P1, 4, 256, 16, 16, 64, 32, 32, 32, 384) == 256
    assert lib.mmdyn_igemm_slab_floats_mx(TCONV_S2P1, 4, 256, 16, 16, 64, 32, 32, 32, 384) == 0
    # not served: FC-level GEMMs, too little work, channel counts the ring's 32-channel K-step does not divide
    assert lib.mmdyn_igemm_planes_served(DENSE, 1, 1024, 1, 1, 512, 1, 1, 512) == 0
    assert lib.mmdyn_igemm_planes_served(CONV, 1, 4, 8, 8, 64, 4, 4, 64) == 0
    assert lib.mmdyn_igemm_planes_served(CONV, 4, 256, 16, 16, 48, 8, 8, 128) == 0
    # a served plane launch writes one partial-sum tile per wave row of its 128-row tiles; the k4 s1 p0 layer's stream-K cut always
    # splits tiles and so wants slabs
    T = lib.mmdyn_igemm_stat_tiles_mx(CONV, 4, 256, 16, 16, 64, 8, 8, 128, 384)
    assert T == 2 * (256 * 64 // 128)
    assert lib.mmdyn_igemm_slab_floats_mx(TCONV_S1P0, 4, 256, 5, 5, 256, 8, 8, 128, 384) > 0
    # weight gradient with both operands split: ~256 blocks (one per CU), a multiple of four slabs; one operand split keeps the
    # register-staged kernels' cut
    assert lib.mmdyn_wgrad_chunks_mx(CONV, 1024 * 25, 256, 128, 128 | 256 | 512) == 8
    assert lib.mmdyn_wgrad_chunks_mx(CONV, 1024 * 64, 128, 64, 128 | 256 | 512) == 32
    assert lib.mmdyn_wgrad_chunks_mx(CONV, 1024 * 256, 64, 32, 128 | 256 | 512) == 64
    assert lib.mmdyn_wgrad_chunks_mx(CONV, 1024 * 25, 256, 128, 128 | 256) == lib.mmdyn_wgrad_chunks_mx(CONV, 1024 * 25, 256, 128, 128)
