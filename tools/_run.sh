cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_decode_persist_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/microbench_decode.py 32 64 552 32 2>&1 | grep -v amdgpu.ids
