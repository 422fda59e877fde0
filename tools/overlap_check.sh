mkdir -p gpurun_out/r5j; rm -f gpurun_out/r5j/*.json
python -m pytest tests/test_gpu_torch_side.py tests/test_gpu_bench.py -m gpu -q -x > gpurun_out/r5j/t.log 2>&1; echo rc=$? >> gpurun_out/r5j/t.log; tail -n 15 gpurun_out/r5j/t.log
for f in "" "--overlap-gather"; do
  python bench.py --force-comm --graph on $f --no-c5 --no-cpu-baseline --envs 8192 --steps 960 --warmup 96 > gpurun_out/r5j/graph_8k$f.json 2>/dev/null
  python bench.py --force-comm --graph on $f --no-c5 --no-cpu-baseline --steps 960 --warmup 96 > gpurun_out/r5j/graph_64k$f.json 2>/dev/null
  python bench.py --force-comm --graph off $f --no-c5 --no-cpu-baseline --envs 8192 --steps 960 --warmup 96 > gpurun_out/r5j/eager_8k$f.json 2>/dev/null
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/r5j/*.json")):
    t=open(f).read().strip().splitlines()
    if not t: print(f, "EMPTY"); continue
    d=json.loads(t[-1]); p=d["phases"]
    print(f, "ms/step %.4f"%d["ms_per_step"], "expected %.4f"%p["expected_ms_per_step"], p["expected_bound"], {k: round(v,2) for k,v in p["per_rank"][0].items()})
PY
