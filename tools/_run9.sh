O=gpurun_out/r05h; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "rc $?" >> $O/pytest.log; tail -3 $O/pytest.log
python bench.py --no-extras --no-cpu-baseline --steps 20 > $O/bench_noextras.json 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05h/bench_noextras.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['stage_ms'], d['repeats'])
PY
export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$C -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-extras > $O/pmc_$C.log 2>&1
done
python - <<'PY'
import csv,glob,collections
for C in ("FETCH_SIZE","WRITE_SIZE"):
    acc=collections.defaultdict(list)
    for path in glob.glob(f"gpurun_out/r05h/pmc_{C}/**/*counter_collection.csv", recursive=True):
        per=collections.defaultdict(float); names={}
        for row in csv.DictReader(open(path)):
            if row["Counter_Name"]!=C: continue
            per[row["Dispatch_Id"]]+=float(row["Counter_Value"]); names[row["Dispatch_Id"]]=row["Kernel_Name"][:40]
        for d,v in per.items(): acc[names[d]].append(v)
    for k,v in acc.items():
        if "k_seed" in k or "k_align" in k: print(C,k,sum(v)/len(v)*1024/1e6,"MB (raw KiB*1024)")
PY
