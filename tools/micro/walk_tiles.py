"""Inputs of tools/micro/walk_tiles.hip: a synthetic kNN graph in the device's cluster order with the tile
program of micro_walk_tiles (tools/micro/host_walk.c).  Run on the GPU box:
    python tools/micro/walk_tiles.py /tmp/wt 500000 16 8 46 [cluster] && ./walk_tiles /tmp/wt 200 16 8 46"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cna_amd import synth, _order  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import micro_host  # noqa: E402


def main():
    d, n = sys.argv[1], int(sys.argv[2])
    nw, rpw, S = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    cluster = int(sys.argv[6]) if len(sys.argv) > 6 else 512
    os.makedirs(d, exist_ok=True)
    t = time.time()
    X, _ = synth.mixture_points(n)
    A = synth.fuzzy_knn_graph(X, k=30)
    print('graph %.1fs nnz/row %.1f' % (time.time() - t, A.nnz / n), flush=True)
    t = time.time()
    order = _order.cluster_order(A, cluster)
    t_order = time.time() - t
    indptr, indices, data = _order.permuted_rows(A, order, 0, n)
    t = time.time()
    tp = micro_host.walk_tiles(indptr, indices, order, nw, rpw, S)
    t_tiles = time.time() - t
    nt = len(tp['tile_src0']) - 1
    nb = len(tp['blk_tile']) - 1
    segl = np.diff(tp['seg']).reshape(nt, nw)
    print('order(%d) %.2fs, tile program %.2fs: %d blocks of %d rows, %.1f tiles of <= %d sources per block, edges/sources '
          '%.2f, records per (tile, wave) mean %.1f max %d, per-tile max/mean %.2f' % (
              cluster, t_order, t_tiles, nb, nw * rpw, nt / nb, S, len(indices) / len(tp['tile_src']), segl.mean(),
              segl.max(), segl.max(1).mean() / segl.mean()), flush=True)
    indptr.astype(np.int64).tofile(d + '/indptr.bin')
    indices.astype(np.int32).tofile(d + '/idx.bin')
    data.astype(np.float32).tofile(d + '/val.bin')
    rec = np.zeros(len(indices), dtype=[('w', np.float32), ('slot', np.uint16), ('row', np.uint8), ('pad', np.uint8)])
    rec['w'][tp['rec_pos']] = data.astype(np.float32)
    rec['slot'] = tp['rec_slot']
    rec['row'] = tp['rec_row']
    rec.tofile(d + '/rec.bin')
    tp['blk_tile'].tofile(d + '/blktile.bin')
    tp['tile_src0'].tofile(d + '/tilesrc0.bin')
    tp['tile_src'].tofile(d + '/tilesrc.bin')
    tp['seg'].tofile(d + '/seg.bin')
    # producer / consumer variant: one self-describing blob of BLOB bytes per tile -- nw + 1 uint16 record offsets
    # (the waves' segments inside the tile), then the tile's records
    BLOB, HDR = 4096, 64
    seg = tp['seg']
    t0 = seg[0:nt * nw:nw]
    cnt = np.diff(np.append(t0, seg[nt * nw]))
    cap = (BLOB - HDR) // 8
    print('records per tile: mean %.1f max %d (blob holds %d)' % (cnt.mean(), cnt.max(), cap), flush=True)
    assert cnt.max() <= cap and nw + 1 <= HDR // 2
    blob = np.zeros((nt, BLOB), dtype=np.uint8)
    offs = (seg[:nt * nw].reshape(nt, nw) - t0[:, None]).astype(np.uint16)
    hdr = np.zeros((nt, HDR // 2), dtype=np.uint16)
    hdr[:, :nw] = offs
    hdr[:, nw] = cnt.astype(np.uint16)
    blob[:, :HDR] = hdr.view(np.uint8).reshape(nt, HDR)
    recb = rec.view(np.uint8).reshape(-1, 8)
    tile_of = np.repeat(np.arange(nt), cnt)
    pos_in = np.arange(len(indices)) - np.repeat(t0, cnt)
    flat = blob.reshape(-1)
    base = tile_of * BLOB + HDR + pos_in * 8
    for k in range(8):
        flat[base + k] = recb[:, k]
    blob.tofile(d + '/blob.bin')


if __name__ == '__main__':
    main()
