"""CPU oracle for the ACT masked-point-modeling hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy / pure
PyTorch-CPU / plain C) of the reference algorithm for the path named in
BASELINE.json:north_star.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import, call, link or execute anything in here.  The product path
(``act_amd/``) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * mini-PointNet Encoder, Transformer Block/Attention/Mlp, encoder/decoder
    stacks, DGCNN, FoldingNet decoder, teacher prompt path, Stage-II forward,
    Stage-I forward + losses, Chamfer reductions, optimizer param groups:
    PINNED against golden vectors produced by importing the reference's own
    Python modules in the build container (tests/golden/make_golden.py).
  * FPS: pinned against the reference's in-tree pure-PyTorch
    ``farthest_point_sample`` (part_segmentation/models/pointnet2_utils.py:60-81)
    run with the start index forced to 0.  The CUDA wheel the training path
    actually calls (pointnet2_ops, un-vendored, unpinned HEAD) is absent:
    parity with *that binary* is UNPINNED.
  * kNN: direct-difference distances, ascending (dist, idx).  The wheel
    (KNN_CUDA 0.2, un-vendored) is absent: parity with that binary is
    UNPINNED; set agreement with the in-tree expansion-form ``knn_point``
    (models/dvae.py:120-152) is measured and stored in the golden file.
  * Chamfer kernels: restated from extensions/chamfer_dist/chamfer.cu (CUDA,
    unbuildable here: needs nvcc + torch CUDA headers); pinned by the
    reference's only test, gradcheck in float64 (extensions/chamfer_dist/test.py:23-29).
"""
