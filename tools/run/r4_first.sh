# round 4, first GPU call: the new / changed tests, then the baseline bench line with the per-shape composite roofline
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1500 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "wide_dynamic or large_mean or non_finite" > $O/r4a_tests_conv.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_trunk.py -x -q -m gpu -k "producer" > $O/r4a_tests_trunk.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_tta.py -q -m gpu -k "three_tta_steps or batch_of_two" > $O/r4a_tests_tta.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_entrypoints.py -q -m gpu -k "swin_buckets or bucketed_exchange or rccl" > $O/r4a_tests_entry.txt 2>&1
timeout 600 python bench.py > $O/r4a_bench.json 2> $O/r4a_bench.err
tail -3 $O/r4a_tests_conv.txt $O/r4a_tests_trunk.txt $O/r4a_tests_tta.txt $O/r4a_tests_entry.txt
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r4a_bench.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["ms_per_step"], d.get("adapt_only_ms"), r["frac"], r.get("frac_of_fp32_matrix_peak"), r.get("frac_composite"), r.get("algorithmic_bytes_per_step"), r.get("hbm_bound_launches_per_step"))
print((d.get("sgd_all") or {}).get("value"), (d.get("swin") or {}).get("ms_per_step"), (d.get("swin_c5_bf16") or {}).get("ms_per_step"), d.get("dp_graph"))
for s in r["by_shape"]: print(s)
PY
