cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_gpu_nn_ops.py -m gpu -q -x -p no:cacheprovider -k "bn or batchnorm" 2>&1 | tail -4
ROUNDS=2 STEPS=30 timeout 600 bash tools/ab.sh "" "SNIPER_BN_APPLY_FLAT=1"
bash tools/kernel_ab.sh "bn_apply" "" "SNIPER_BN_APPLY_FLAT=1" 2>&1 | tail -8
