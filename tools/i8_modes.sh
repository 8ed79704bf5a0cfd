R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02/i8modes; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for M in 0 1 2 3; do
  CNA_I8_MODE=$M timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o m$M -- python $R/tools/kbench_null.py 2000000x200 200000x50 > $OUT/m$M.log 2>&1
  f=$(find $OUT -name "m${M}_kernel_stats.csv"); echo "mode $M"; grep "k_null_i8" $f | sed 's/"void k_null_i8<\([0-9]\), *\([0-9]\)>[^"]*"/KS\1 M\2/; s/"void k_null_i8<\([0-9]\)>[^"]*"/KS\1/' | cut -c1-60
done
