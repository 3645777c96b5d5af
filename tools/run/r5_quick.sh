# quick A/B: TAM branch latency + the bench line (no Swin / SGD legs)   usage: bash tools/run/r5_quick.sh <tag>
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=${1:-q}
timeout 300 python tools/bench_tam.py --out $O/${T}_tam.json > $O/${T}_tam.txt 2>&1
timeout 600 python bench.py --no-swin --no-sgd-all --no-cpu-baseline --no-streaming > $O/${T}_bench.json 2> $O/${T}_bench.err
cat $O/${T}_tam.txt | tail -5
python - <<PY
import json
d=json.loads(open("$O/${T}_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("$T", round(d["value"],2), round(d["ms_per_step"],3), d.get("adapt_only_ms"), r.get("frac_of_fp32_matrix_peak"))
PY
