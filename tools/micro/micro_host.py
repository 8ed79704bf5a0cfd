"""Python side of tools/micro/host_walk.c: the planners of the walk-step probes (gather_lds / walk_lds / walk_tiles), which
left the product library in round 6 (the kernels they serve were measured and dropped: HISTORY.md 5).  Builds
tools/micro/libmicro_host.so with gcc on first use."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cna_amd._order import usable_cpus, DEFAULT_CLUSTER    # noqa: E402

_lib = None


def load():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, 'libmicro_host.so')
        src = os.path.join(HERE, 'host_walk.c')
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.run(['gcc', '-O3', '-fPIC', '-shared', '-Wall', src, '-o', so, '-lpthread'], check=True)
        _lib = C.CDLL(so)
        for name in ('micro_block_sources', 'micro_walk_blocks', 'micro_walk_tiles'):
            getattr(_lib, name).restype = C.c_int64
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def block_sources(indptr, indices, n_cols, B, cap):
    """(src_ptr, src, slot) of micro_block_sources for the device-ordered CSR rows (indptr int64,
    indices int32): per block of B rows the distinct columns, per edge the column's position in it."""
    n_local = len(indptr) - 1
    nblocks = (n_local + B - 1) // B
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    src_ptr = np.zeros(nblocks + 1, dtype=np.int64)
    src = np.empty(max(len(indices), 1), dtype=np.int32)
    slot = np.empty(max(len(indices), 1), dtype=np.uint16)
    tot = load().micro_block_sources(C.c_int64(n_local), C.c_int64(int(n_cols)), _p(indptr), _p(indices), int(B), int(cap),
                                             _p(src_ptr), _p(src), _p(slot))
    if tot < 0:
        raise MemoryError('micro_block_sources')
    return src_ptr, src[:tot].copy(), slot[:len(indices)]


def walk_blocks(indptr, indices, n_cols, bmax=64, cap=960, super_rows=DEFAULT_CLUSTER):
    """(blk_row, src_ptr, src, slot) of micro_walk_blocks for the device-ordered CSR rows: blocks of
    at most `bmax` consecutive rows with at most `cap` distinct columns, the sorted columns per block and
    the position of every edge's column in its block's list."""
    n_local = len(indptr) - 1
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    blk_row = np.empty(n_local + 1, dtype=np.int64)
    src_ptr = np.empty(n_local + 1, dtype=np.int64)
    src = np.empty(max(len(indices), 1), dtype=np.int32)
    slot = np.empty(max(len(indices), 1), dtype=np.uint16)
    nb = load().micro_walk_blocks(C.c_int64(n_local), C.c_int64(int(n_cols)), _p(indptr), _p(indices), int(bmax), int(cap),
                                          int(super_rows), usable_cpus(16), _p(blk_row), _p(src_ptr),
                                          _p(src), _p(slot))
    if nb < 0:
        raise MemoryError('micro_walk_blocks')
    return blk_row[:nb + 1].copy(), src_ptr[:nb + 1].copy(), src[:int(src_ptr[nb])].copy(), slot[:len(indices)]


def walk_tiles(indptr, indices, key, nw=16, rpw=8, S=46):
    """Tile program of the LDS-tiled walk step (micro_walk_tiles) for the device-ordered CSR rows:
    dict(blk_tile, tile_src0, tile_src, seg, rec_pos, rec_slot, rec_row), or None when the rows do not list
    their columns in ascending caller's index (key[c] = caller's index of device column c)."""
    lib = load()
    n_local = len(indptr) - 1
    B = nw * rpw
    nb = (n_local + B - 1) // B
    indptr = np.ascontiguousarray(indptr, dtype=np.int64)
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    key = np.ascontiguousarray(key, dtype=np.int64)
    blk_tile = np.zeros(nb + 1, dtype=np.int64)
    args = (C.c_int64(n_local), _p(indptr), _p(indices), _p(key), int(nw), int(rpw), int(S), usable_cpus(16))
    nt = lib.micro_walk_tiles(*args, _p(blk_tile), None, None, None, None, None, None)
    if nt == -2:
        return None
    if nt < 0:
        raise MemoryError('micro_walk_tiles')
    nnz = len(indices)
    out = dict(blk_tile=blk_tile, tile_src0=np.zeros(nt + 1, dtype=np.int64), tile_src=np.zeros(max(nnz, 1), dtype=np.int32),
               seg=np.zeros(nt * nw + 1, dtype=np.int64), rec_pos=np.zeros(max(nnz, 1), dtype=np.int64),
               rec_slot=np.zeros(max(nnz, 1), dtype=np.uint16), rec_row=np.zeros(max(nnz, 1), dtype=np.uint8))
    got = lib.micro_walk_tiles(*args, _p(out['blk_tile']), _p(out['tile_src0']), _p(out['tile_src']),
                                  _p(out['seg']), _p(out['rec_pos']), _p(out['rec_slot']),
                                  _p(out['rec_row']))
    if got != nt:
        raise RuntimeError('micro_walk_tiles: %d' % got)
    out['tile_src'] = out['tile_src'][:int(out['tile_src0'][nt])].copy()
    return out


