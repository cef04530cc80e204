cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
for L in "" "SNIPER_HIP_LIB=sniper_amd/lib/libsniper_hip_fl.so"; do echo "## $L"; env $L python tools/conv_trace.py --cfgs 18 --only 's3 ' 2>&1 | grep -v amdgpu.ids | grep warm; done
timeout 300 env SNIPER_HIP_LIB=sniper_amd/lib/libsniper_hip_fl.so python -m pytest tests/test_gpu_baseline_shapes.py -m gpu -q -x -p no:cacheprovider -k "conv or shape" 2>&1 | tail -3
ROUNDS=2 STEPS=30 bash tools/ab.sh "" "SNIPER_HIP_LIB=sniper_amd/lib/libsniper_hip_fl.so"
