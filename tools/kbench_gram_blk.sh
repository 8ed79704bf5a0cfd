#!/bin/bash
# A/B of the 3x3-block Gram kernel on 5..10 tiles per side (CNA_GRAM_BLK_SMALL=0: the tile-per-wave kernel of rounds 1-5)
for shape in "1000000 100" "1000000 112" "500000 80" "500000 96" "500000 128" "500000 144" "500000 160" "200000 50" "250000 200"; do
  for sw in 0 1; do
    echo -n "$shape small_blk=$sw: "; CNA_GRAM_BLK_SMALL=$sw python tools/kbench_gram.py $shape 2>&1 | tail -1
  done
done
