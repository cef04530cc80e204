# conv_dma timing probes (profiles/r06_conv_probe_no_dma_clock.txt, r06_conv_probe_barrier.txt): the specialised kernel's K loop without
# operand traffic (SNIPER_CONV_PROBE_SKIP_A bits 0 / 1) and without its step barrier (bit 2) -- wrong results, phase stamps + shader clock only.
#   gpurun -- 'bash tools/probes/conv_probe_visit.sh'
cd $GRAFT_REPO_ROOT; export PYTHONDONTWRITEBYTECODE=1
for P in 0 3 4 7; do echo "## SNIPER_CONV_PROBE_SKIP_A=$P"; SNIPER_CONV_PROBE_SKIP_A=$P python tools/conv_trace.py --cfgs 18,14 --only 's3 ' 2>&1 | grep -v amdgpu.ids | grep warm; done
