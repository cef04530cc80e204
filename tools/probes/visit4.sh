cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
for P in 0 8 16 24 56; do echo "## flags build, SNIPER_CONV_PROBE_SKIP_A=$P"; SNIPER_CONV_PROBE_SKIP_A=$P SNIPER_HIP_LIB=sniper_amd/lib/libsniper_hip_fl.so timeout 120 python tools/conv_trace.py --cfgs 18,19 --only 's3 3x3' 2>&1 | grep -v amdgpu.ids | grep "fwd.*warm"; done
