cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc2; mkdir -p $R/gpurun_out/pmc2
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 180 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc2/$tag -- python3 $R/tools/mesh_pmc_driver.py > $R/gpurun_out/pmc2/$tag.log 2>&1 || echo "group failed: $grp"
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
rows=collections.defaultdict(dict)
for f in glob.glob(R+"/gpurun_out/pmc2/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "pathtrace" not in row["Kernel_Name"]: continue
        rows[row["Counter_Name"]][int(row["Dispatch_Id"])]=float(row["Counter_Value"])
for c in sorted(rows):
    ids=sorted(rows[c]); print("%-26s"%c, " ".join("%11.4g"%rows[c][i] for i in ids))
PY
