# PMC view of the integer local-null kernel (tools/kbench_null.py at 2M x 200); counters only, one pass each
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02/i8pmc; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
for M in 0; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o sq1_m$M --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -- python $R/tools/kbench_null.py ${SIZES:-2000000x200} > $OUT/sq1_m$M.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o sq2_m$M --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE -- python $R/tools/kbench_null.py ${SIZES:-2000000x200} > $OUT/sq2_m$M.log 2>&1
done
python - <<P
import csv, glob, collections
for f in sorted(glob.glob("$OUT/*_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_null_i8' not in k and 'k_quant_x' not in k and 'k_null_recheck' not in k: continue
        k = k.split('(')[0][-40:]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] in ('SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE'): n[k] += 1
    print(f.split('/')[-1])
    for k, d in acc.items():
        print('  ', k, 'dispatches', n[k], ' '.join(f'{c}={v / max(n[k], 1):.4g}' for c, v in sorted(d.items())))
P
