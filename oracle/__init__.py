"""oracle -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's algorithms for the SNIPER hot path, used exclusively as the
*checker*: by ``tests/``, by ``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline``
leg.  Nothing under ``sniper_amd/`` imports this package; the product path fails loudly when the
HIP library is missing instead of falling back here.

Pinning status (see DESIGN.md section "Oracle"):

* chips / IoU / anchor labelling / chip extraction / box assignment / hard NMS: pinned against the
  reference's own native + Python code run in the build container (``oracle/_ref``,
  ``oracle/ref_py.py``) and against committed golden vectors (``tests/golden``).
* soft-NMS, cpu_nms: restated line by line from ``lib/nms/cpu_nms.pyx`` (which does not build
  under Cython 3); pinned by hand-derived known-answer cases only.
* cv2.resize(uint8, INTER_LINEAR) (im_worker): ``oracle/cv_resize.py`` + ``sniper_oracle.c::orc_cv_resize_linear_u8c3``, two
  independent restatements of OpenCV's published fixed-point algorithm (OpenCV is a third-party dependency absent from
  the reference tree and from this image): pinned to the published algorithm, no cv2-minted vectors.
* cv2.dilate / cv2.findContours(RETR_LIST) / cv2.boundingRect (gmask, FocusChip generation): ``oracle/cv_contours.py`` -- the
  published algorithms (Suzuki & Abe border following), held against the product's border following AND its connected-component
  form: pinned to the published algorithm, no cv2-minted vectors.
* network ops whose source lives in the un-vendored SNIPER-mxnet submodule (Convolution,
  BatchNorm, MultiProposalTarget, DeformablePSROIPooling, ...): **parity unpinned** -- the
  restatements in ``oracle/nn.py`` follow the published definitions and are cross-checked
  against torch-CPU fp32 for the standard ops.
"""
from .capi import (  # noqa: F401
    candidate_chips,
    shuffle_perm,
    chips_generate,
    bbox_overlaps,
    ignore_overlaps,
    nms_sorted,
    cpu_nms,
    soft_nms,
)
