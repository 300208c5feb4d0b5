"""genedex_amd -- MI355X-native FM-index query engine with the query API of feldroop/genedex.

Only what the query hot path needs lives here: csrc/ (HIP kernels, host C++ and the C ABI of
include/gdx.h, built into libgdx.so), and the host-side mirror of the reference's public interface
(alphabet, FmIndexConfig, FmIndex, Cursor, Hit).
"""
from . import alphabet
from .alphabet import Alphabet

__all__ = ["alphabet", "Alphabet", "FmIndexConfig", "FmIndex", "PartitionedFmIndex", "Cursor", "Hit", "GdxError"]


def __getattr__(name):
    # the query API needs libgdx.so; importing the alphabet tables alone does not
    if name in ("FmIndexConfig", "FmIndex", "PartitionedFmIndex", "Cursor", "Hit", "pack_queries"):
        from . import index

        return getattr(index, name)
    if name == "GdxError":
        from ._lib import GdxError

        return GdxError
    raise AttributeError(name)
