# round 4: ring depth and tile size of the staging-rate probe (tools/micro/stage_rate.hip); run on the GPU box
cd /root/repo
for cfg in "8 16 192" "8 16 96" "8 16 64"; do
  set -- $cfg
  d=/tmp/wt_$1_$2_$3
  python tools/micro/walk_tiles.py $d 500000 $1 $2 $3 2>&1 | grep order
  for nb in 2 3 4 6; do NBUF=$nb tools/micro/stage_rate $d 200 $1 $3 256 2>&1 | grep -E "^n =|xcd_chunk=4 |G=32 |G=16 |exceeds"; done
done
