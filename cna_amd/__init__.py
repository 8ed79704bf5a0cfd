"""cna_amd -- MI355X-native hot path of covarying neighborhood analysis (CNA).

Same call surface as the reference package (``cna.tl.association``, ``cna.tl.nam``,
``cna.tl.svd_nam``, ``cna.tl.diffuse``, ``cna.tl.diffuse_stepwise``,
``cna.ut.obs_to_sample``; /root/reference/src/cna/__init__.py:1-3) so that
``import cna_amd as cna`` is a drop-in for that path.  All O(cells) arithmetic runs
in hand-written HIP kernels for gfx950 reached through the C ABI declared in
``include/cna_hip.h``; there is no CPU fallback -- the calls raise if the
library or a GPU is missing.
"""
from . import tools as tl
from . import utils as ut

__version__ = '0.1.0'
__all__ = ['tl', 'ut']
