"""``cna_amd.tl`` -- same names as the reference's ``cna.tl`` (cna/tools/__init__.py:1-10)."""
from ._nam import nam, svd_nam, diffuse, diffuse_stepwise
from ._association import association

__all__ = [
    'association',
    'nam',
    'svd_nam',
    'diffuse',
    'diffuse_stepwise',
]
