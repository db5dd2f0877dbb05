cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/r2g/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2g/pytest.log
tail -3 gpurun_out/r2g/pytest.log
bash tools/refresh_profiles.sh r02
