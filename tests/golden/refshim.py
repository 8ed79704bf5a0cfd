"""Import the *reference* CNA package from /root/reference in this container only.

Used by tests/golden/make_golden.py to capture golden vectors.  The reference
never travels to the GPU box; nothing under -m gpu, smoke() or bench.py imports
this module.  Recipe:
  * stub ``anndata`` / ``scanpy`` (imported at module scope by the reference,
    never called on the path we exercise),
  * make ``importlib.metadata.version('anndata')`` answer a modern version so
    ``get_connectivity`` reads ``data.obsp`` (_nam.py:12-19),
  * restore ``np.NaN`` which _association.py:230 still uses (removed in NumPy 2).
"""
import sys
import types
import importlib.metadata as _md

REFERENCE_SRC = '/root/reference/src'


def load_reference():
    import numpy as np
    if not hasattr(np, 'NaN'):
        np.NaN = np.nan
    for name in ('anndata', 'scanpy'):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    real_version = _md.version

    def version(pkg):
        if pkg == 'anndata':
            return '0.10.9'
        return real_version(pkg)

    _md.version = version
    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
    import cna  # noqa: the reference package
    return cna
