"""CPU oracle of the FuseTrack inference path — TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU / numpy fp32 restatement of `PanopticFuseTrack.simple_test`
(/root/reference/mmdet/models/detectors/panoptic_fusetrack.py:502-606) and of every operator below it.
Each function cites the reference file:line it follows.

Rules (see the task statement): only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package, and only as the checker / the timed CPU baseline. Nothing under `vps_amd/` imports it.

Pinning status
  * The reference ships NO tests, golden vectors or fixtures for this path.
  * Python-level modules (ResNet, FPN, BFPTcea, TCEA_Fusion, FlowNet2/C/S/SD/Fusion wiring, UPSNetFPN wiring,
    RPNHead.get_bboxes, delta2bbox, SingleRoIExtractor level mapping, SharedFCBBoxHead, MaskROI, TrackHead,
    FCNMaskHead, MaskRemoval, SegTerm, the tracking block of simple_test_bboxes) are PINNED against the
    reference's own code imported in the build container with import shims (tests/golden/make_golden.py);
    the resulting vectors are committed under tests/golden/ and checked by tests/test_oracle_golden.py.
  * The reference's CUDA-only operators (correlation, resample2d, channelnorm, roi_align, deform_conv, nms,
    gpu_nms) and cv2.resize cannot run here (no CUDA, binaries stripped, cv2 absent): their restatements in
    `oracle.ops` follow the .cu sources line by line but are **parity unpinned** against executed reference
    output. DESIGN.md repeats this.
"""
