cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_train_gpu.py -q -x -m gpu 2>&1 | tail -2
timeout -k 10 400 python tools/train_split.py 2>&1 | tail -4
