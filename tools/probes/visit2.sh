cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
for P in 0 3; do echo "## SNIPER_CONV_PROBE_SKIP_A=$P"; SNIPER_CONV_PROBE_SKIP_A=$P python tools/conv_trace.py --cfgs 18,14 --only 's3 ' 2>&1 | grep -v amdgpu.ids | grep warm; done
