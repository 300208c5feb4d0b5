"""CPU-only checks of the drop-in boundary: libgdx.so loads, exports every symbol the headers declare,
and the product fails loudly (no CPU fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("gdx.h", "gdx_experimental.h", "gdx_bench.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(gdx_[a-z0-9_]+)\s*\(", src))
    return names


def test_library_exports_every_declared_symbol():
    from genedex_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 35
    for name in sorted(decl):
        assert hasattr(lib, name), f"{name} is declared in include/*.h but not exported by libgdx.so"
    # and the ctypes stub covers the whole ABI
    assert decl == set(_lib.SIGNATURES), decl ^ set(_lib.SIGNATURES)


def test_the_core_header_is_the_reference_api_and_little_else():
    """include/gdx.h is what a binding of the reference's API needs; the steps of earlier rounds that the
    core calls superseded live in gdx_experimental.h (round 6).  The split must not creep back."""
    def names(h):
        src = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", h)).read(), flags=re.S)
        return set(re.findall(r"\b(gdx_[a-z0-9_]+)\s*\(", src))

    core, exp = names("gdx.h"), names("gdx_experimental.h")
    assert not core & exp
    assert len(core) <= 95 and len(exp) >= 30
    for n in ("gdx_index_build", "gdx_index_from_parts", "gdx_count_many", "gdx_cursors_for_many_queries", "gdx_locate_many",
              "gdx_cursor_empty", "gdx_cursor_extend_front_many", "gdx_cursor_locate_many", "gdx_rank_many", "gdx_symbol_at_many",
              "gdx_index_free", "gdx_last_error", "gdx_index_info", "gdx_locate_many_step_compact_layout_dev"):
        assert n in core, n
    assert not [n for n in core if "_packed" in n and n not in ("gdx_packed_bytes",)] and not [n for n in core if "_hint" in n]


def test_no_cpu_fallback_without_a_gpu():
    from genedex_amd import FmIndexConfig, GdxError, _lib, alphabet

    if _lib.load().gdx_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(GdxError) as e:
        FmIndexConfig("i32").construct_index([b"ACGT"], alphabet.ascii_dna())
    assert e.value.status == _lib.GDX_ERR_DEVICE


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "genedex_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"
