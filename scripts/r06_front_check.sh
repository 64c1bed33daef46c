cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_models.py tests/test_u8.py tests/test_gpu_race.py -x -q 2>&1 | tail -15
python scripts/time_kernels.py 2>&1 | tail -12
MF_DEV=1 MF_NO_PAIR_FRONT=1 python scripts/time_kernels.py 2>&1 | tail -4
