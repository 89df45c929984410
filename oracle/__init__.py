"""CPU oracle for the Dedark-YOLO hot path -- TEST INFRASTRUCTURE ONLY.

This package is a from-scratch, functional (state_dict in, tensors out) fp32
PyTorch-CPU restatement of the reference algorithm for the path named by
BASELINE.json `north_star`.  It exists to CHECK the HIP product path:

  * only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline`
    leg may import it -- the product package `dedark_yolo_amd` never does;
  * it is never the thing measured as the product, never shipped;
  * it is pinned against golden vectors captured from the reference itself
    (tests/golden/*.npz, generator tests/golden/make_golden.py, which imports
    /root/reference in the build container with arithmetic-free stand-ins for
    the import-only dependencies cv2 / easydict / torchvision) and against the
    known-answer values of SURVEY.md Appendix A (KA1..KA5).

Parity status: PINNED for rows A1..A18 (goldens from the reference);
`nms` (A19, torchvision.ops.nms -- un-vendored third-party, version unpinned
by the reference) is "parity unpinned": restated from torchvision's documented
semantics only.

Every function cites the reference file:line it follows (U/ = ultralytics/).
"""
