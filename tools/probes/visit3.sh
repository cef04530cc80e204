cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests/test_gpu_nn_ops.py -m gpu -q -x -p no:cacheprovider -k "finalize_fused" 2>&1 | tail -4
ROUNDS=1 STEPS=30 timeout 600 bash tools/ab.sh "" "SNIPER_BN_FUSED_FINALIZE=2"
bash tools/kernel_ab.sh "bn_" "SNIPER_BN_FUSED_FINALIZE=2" 2>&1 | tail -12
