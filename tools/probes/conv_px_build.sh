#!/bin/bash
# Builds sniper_amd/lib/libsniper_hip_px.so: the shipped kernel sources + the pixel-stationary 1x1 experiment (conv_px_experiment.hip as
# conv_px.hip, the hook of conv_px_hook.patch in conv.hip / conv_common.h / conv_dma.hip), from a scratch COPY of csrc (sniper_amd/csrc_px,
# git-ignored).  The shipped library and its sources are not touched.  Run the tools with SNIPER_HIP_LIB=sniper_amd/lib/libsniper_hip_px.so.
set -e
cd "$(dirname "$0")/../.."
rm -rf sniper_amd/csrc_px && cp -r sniper_amd/csrc sniper_amd/csrc_px
( cd sniper_amd/csrc_px && patch -p3 -s < ../../tools/probes/conv_px_hook.patch )
sed -n '/^\/\/ conv_px.hip --/,$p' tools/probes/conv_px_experiment.hip > sniper_amd/csrc_px/conv_px.hip
# mode 3 of sn_conv_px / SNIPER_CONV_PX: staggered, FORWARD layers only (the BatchNorm-backward variant stays on the tile kernel)
sed -i -e 's/(px \& 3) != 0 \&\& conv_px_ok(q)/(px \& 3) != 0 \&\& !((px \& 3) == 3 \&\& q.bn_x) \&\& conv_px_ok(q)/' -e 's/(px \& 3) == 2, px >> 4, s)/(px \& 3) >= 2, px >> 4, s)/' sniper_amd/csrc_px/conv.hip
grep -c "(px & 3) == 3 && q.bn_x" sniper_amd/csrc_px/conv.hip
SNIPER_BUILD_CSRC="$(pwd)/sniper_amd/csrc_px" SNIPER_BUILD_SUFFIX=_px python -m sniper_amd.build | grep -v "warning\|unused variable" | tail -3
grep -E "Function Name|VGPRs:|ScratchSize" sniper_amd/lib/obj_px/conv_px.remarks | paste - - - | sed -E 's/.*conv_px_kernel(I[A-Za-z0-9]+E)Ev.*VGPRs: ([0-9]+).*lane\]: ([0-9]+).*/\1 vgpr \2 scratch \3/'
