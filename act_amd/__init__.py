"""act_amd -- MI355X-native (gfx950) implementation of ACT's masked-point-modeling hot path.

Layout mirrors the reference's import surface for this path so it is a drop-in:
  act_amd.models            build_model_from_cfg, MODELS, dvae.{Group,Encoder,DGCNN,...}, act.{Block,...}
  act_amd.utils.misc        fps
  act_amd.pointnet2_ops     pointnet2_utils.{furthest_point_sample, gather_operation}
  act_amd.knn_cuda          KNN
  act_amd.extensions.chamfer_dist   ChamferFunction, ChamferDistanceL1/L2/L2_split
  act_amd.tools             runner_pretrain.run_net, runner_autoencoder.run_net, builder
All arithmetic on the path runs in hand-written HIP kernels (act_amd/csrc, C ABI in include/act_hip.h).
"""
__version__ = "0.1.0"
