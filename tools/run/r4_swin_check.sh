export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_kernels.py -x -q -m gpu -k "bf16 or layernorm or bfloat16" > $O/r4e_tests_units.txt 2>&1; tail -15 $O/r4e_tests_units.txt
timeout 1500 python -m pytest tests/test_gpu_tta.py tests/test_gpu_fullsize.py -x -q -m gpu -k "swin" > $O/r4e_tests_swin.txt 2>&1; tail -15 $O/r4e_tests_swin.txt
timeout 600 python - > $O/r4e_swin_legs.json 2> $O/r4e_swin_legs.err <<'PY'
import json, sys, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda:0")
out = {}
for key, cfg in (("swin", dict(views=2, frames=16, window_depth=8, classes=101, bf16=False, steps=8)),
                 ("swin_c5_bf16", dict(views=4, frames=32, window_depth=16, classes=174, bf16=True, steps=4))):
    out[key] = bench.swin_leg(dev, **cfg)
    r = out[key]
    print(key, round(r["ms_per_step"], 2), {k: (round(v["achieved"], 1), round(v["ms_per_step"], 2)) for k, v in r["roofline"].items()}, file=sys.stderr)
print(json.dumps(out))
PY
tail -4 $O/r4e_swin_legs.err
