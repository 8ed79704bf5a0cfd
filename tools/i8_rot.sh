R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02/i8rot; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for ROT in 0 1; do
  CNA_I8_ROT=$ROT timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r$ROT -- python $R/tools/kbench_null.py 2000000x200 1000000x100 200000x50 > $OUT/r$ROT.log 2>&1
  f=$(find $OUT -name "r${ROT}_kernel_stats.csv"); echo "rot $ROT"; grep "k_null_i8\|k_quant_x\|k_null_recheck" $f | sed 's/"void k_null_i8<\([0-9]\), *\([0-9]\), *\([0-9]\)>[^"]*"/KS\1 G\2/; s/"\(k_[a-z_]*\)[^"]*"/\1/' | cut -c1-70
done
