# tools/dev/pmc_spec_lds.sh [nfft...] — LDS counters of the spectrum kernel the installed libsdrfm.so runs (on the GPU box)
export TMPDIR=/tmp
for n in ${@:-1024}; do
  rm -rf /tmp/pmc_$n
  rocprofv3 --output-format csv --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -d /tmp/pmc_$n -o pmc -- python3 $GRAFT_REPO_ROOT/bench.py --workload spectrum --nfft $n --steps 30 --warmup 5 > /tmp/pmc_$n.log 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmc_$n/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_spectrum" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k: sum(v)/len(v) for k,v in acc.items()}
print("N=$n LDS instr %.2fM  array cycles per instr %.2f  LDS active / kernel cycles per CU %.2f" % (m["SQ_INSTS_LDS"]/1e6, m["SQ_LDS_IDX_ACTIVE"]/m["SQ_INSTS_LDS"], m["SQ_LDS_IDX_ACTIVE"]/256/(m["SQ_BUSY_CYCLES"]/32)))
PY
done
