#!/bin/bash
# round 6, job 28: long randomised stress of the final tree: the synchronous calls in parts with caller map buffers, the pipelines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r28; mkdir -p $out
( time timeout 1500 python tools/stress_sync_parts.py 60 7 ) > $out/sync_parts.log 2>&1; echo "rc=$?" >> $out/sync_parts.log
( time timeout 1500 python tools/stress_pipeline.py 40 5 ) > $out/pipeline.log 2>&1; echo "rc=$?" >> $out/pipeline.log
( time timeout 900 python -m pytest tests/test_gpu_stress.py -x -q -m gpu ) > $out/pytest.log 2>&1; echo "rc=$?" >> $out/pytest.log
tail -4 $out/sync_parts.log; tail -4 $out/pipeline.log; tail -4 $out/pytest.log
