cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
timeout -k 10 500 python -m pytest tests/test_lstm_cluster_gpu.py -x -q -k "two_layer or switch_parity or partial_dh or matches_step" > gpurun_out/r4a/pytest.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/r4a/pytest.txt
tail -5 gpurun_out/r4a/pytest.txt
