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
__all__ = ['tl', 'ut', 'tune_host_allocator']


def tune_host_allocator(mmap_threshold=32 << 20, trim_threshold=1 << 30):
    """Optional host tuning (glibc only): keep cell-sized numpy temporaries on the heap instead of
    mmap/munmap-ing them on every call.  Each call of ``association`` creates a few dozen MB of
    per-cell temporaries (sample codes, masks, the two ``data.obs`` columns); on large hosts the
    page faults of fresh mappings cost more than the GPU work (measured on a 256-CPU box at 1M
    cells: 21 ms for one 8 MB ``take``).  Returns True if the allocator accepted the settings."""
    import ctypes
    try:
        libc = ctypes.CDLL('libc.so.6')
        ok1 = libc.mallopt(-3, int(mmap_threshold))      # M_MMAP_THRESHOLD
        ok2 = libc.mallopt(-1, int(trim_threshold))      # M_TRIM_THRESHOLD
        return bool(ok1 and ok2)
    except Exception:
        return False
