# pair kernel: parity + phase clocks + A/B bench (round 4)
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/${1:-r4b}
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_lstm_cluster_gpu.py -x -q -k "two_layer or switch_parity" > $O/pytest.txt 2>&1 || { tail -20 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
python tools/prof_rs.py > $O/prof_pair.txt 2>&1 || exit 1
python bench.py --steps 20 --warmup 5 > $O/bench_pair.json 2> $O/bench_pair.err || exit 1
FHVAE_NO_RS_PAIR=1 python bench.py --steps 20 --warmup 5 > $O/bench_layer.json 2> $O/bench_layer.err || exit 1
python - <<PY
import json
for f in ("bench_pair.json", "bench_layer.json"):
    d = json.loads(open("$O/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
