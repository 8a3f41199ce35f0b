#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r1
( time timeout 1500 python -m pytest tests/test_gpu_switches.py tests/test_gpu_frontend_chain.py "tests/test_gpu_parity.py" -x -q -k "twin or one_record_buffer or wrap or dropin or adaptor or lazy or captured or bf16_convolutions_tracks or two_side_chains" ) > gpurun_out/r1/pytest_a.log 2>&1
echo "pytest_a rc=$?" >> gpurun_out/r1/pytest_a.log
( time timeout 1800 python -m pytest tests/test_gpu_bench_two_ranks.py -x -q ) > gpurun_out/r1/pytest_b.log 2>&1
echo "pytest_b rc=$?" >> gpurun_out/r1/pytest_b.log
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r1/bench.json 2> gpurun_out/r1/bench.err
echo "bench rc=$?" >> gpurun_out/r1/bench.err
tail -3 gpurun_out/r1/pytest_a.log gpurun_out/r1/pytest_b.log
wc -c gpurun_out/r1/bench.json
