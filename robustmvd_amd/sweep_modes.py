"""Other consumers of the plane sweep on the engine's generic reduction kernel
(mvd_sweep_reduce_f32), with the reference's call shapes:

  cvp_proj_cost            proj_cost, rmvd/models/blocks/cvp_mvsnet_components.py:375-456 (per-pixel depth hypotheses) and
                           the coarse level of CVPMVSNet.forward, rmvd/models/cvp_mvsnet.py:125-160 (depth_hypos (B,D))
  vis_cost_volumes         SingleStage.build_cost_volume + groupwise_correlation,
                           rmvd/models/blocks/vis_mvsnet_singlestage.py:86-122,242 and blocks/utils.py:71-152

Both are CUDA-only in the reference (hard `.cuda()` calls); the arithmetic is the same bilinear sweep as Path B.
"""
import torch

from . import _lib as L
from . import ops


@ops.inference_only
def sweep_reduce(key_feat, src_feats, Ms, depth, mode, groups=1, pix_offset=0.0, stretch=True):
    """key_feat (B,C,h,w); src_feats V x (B,C,h,w); Ms V x (B,3,4) [R|t]; depth (B,D) or (B,D,h,w).
    stretch=True: homo_warp's convention (index = X/Z * W/(W-1) - 0.5 from integer pixel positions);
    stretch=False with pix_offset=0.5: homography_warping's (index = X/Z - 0.5 from positions x + 0.5).
    Returns (B,C,D,h,w) for the variance modes, a list of V (B,groups,D,h,w) volumes for REDUCE_GROUPCORR."""
    lib = L.load()
    kf = L.as_f32(key_feat, "key_feat")
    if kf.dim() != 4:
        raise ValueError("key_feat must be (B,C,h,w)")
    B, C, h, w = kf.shape
    dev = kf.device
    srcs = [L.as_f32(s, f"src_feats[{i}]", (B, C, h, w), dev) for i, s in enumerate(ops._views(src_feats, "src_feats"))]
    V = len(srcs)
    Ms = [L.as_f32(m, f"Ms[{i}]", (B, 3, 4), dev) for i, m in enumerate(ops._views(Ms, "Ms", V))]
    dv = L.as_f32(depth, "depth", device=dev)
    if dv.dim() == 2 and dv.shape[0] == B:
        per_pixel, D = 0, dv.shape[1]
    elif dv.dim() == 4 and dv.shape[0] == B and tuple(dv.shape[2:]) == (h, w):
        per_pixel, D = 1, dv.shape[1]
    else:
        raise ValueError(f"depth must be (B,D) or (B,D,h,w), got {tuple(dv.shape)}")
    if mode == L.REDUCE_GROUPCORR:
        if C % groups or (C // groups) % 4:
            raise ValueError(f"group correlation needs C/groups a multiple of 4, got {C}/{groups}")
        outs = [torch.empty((B, groups, D, h, w), dtype=torch.float32, device=dev) for _ in range(V)]
    elif mode in (L.REDUCE_VARIANCE, L.REDUCE_VARIANCE_KEYSQ):
        if C % 4:
            raise ValueError(f"C={C} must be a multiple of 4")
        outs = [torch.empty((B, C, D, h, w), dtype=torch.float32, device=dev)]
    else:
        raise ValueError(f"mode {mode}")
    sx, sy = (w / (w - 1), h / (h - 1)) if stretch else (1.0, 1.0)
    wsb = lib.mvd_sweep_reduce_workspace_bytes(B, C, h, w, V)
    wsp = ops._workspace(wsb, dev)
    a_s, k1 = L.ptr_array(srcs)
    a_m, k2 = L.ptr_array(Ms)
    a_o, k3 = L.ptr_array(outs)
    with torch.cuda.device(dev):
        rc = lib.mvd_sweep_reduce_f32(L.ptr(kf), a_s, a_m, L.ptr(dv), per_pixel, float(pix_offset), float(sx), float(sy), -0.5,
                                      mode, groups, B, C, D, h, w, V, a_o, L.ptr(wsp), wsb, L.stream_of(kf))
    L.check(rc, "mvd_sweep_reduce_f32")
    return outs if mode == L.REDUCE_GROUPCORR else outs[0]


def _cvp_transform(ref_in, src_in, ref_ex, src_ex):
    """cvp_mvsnet_components.py:201-210: proj = [K_s E_s[:3]; 0 0 0 1] @ inverse([K_r E_r[:3]; 0 0 0 1]) -> (B,3,4)."""
    last = torch.tensor([[[0.0, 0.0, 0.0, 1.0]]], dtype=torch.float32, device=ref_in.device).repeat(ref_in.shape[0], 1, 1)
    src_proj = torch.cat((torch.matmul(src_in.float(), src_ex.float()[:, 0:3, :]), last), 1)
    ref_proj = torch.cat((torch.matmul(ref_in.float(), ref_ex.float()[:, 0:3, :]), last), 1)
    return torch.matmul(src_proj, torch.inverse(ref_proj))[:, :3, :4].contiguous()


def cvp_proj_cost(ref_feature, src_features, ref_in, src_in, ref_ex, src_ex, depth_hypos, reproduce_alias_bug=True):
    """CVP-MVSNet cost volume.  ref_feature (B,C,h,w); src_features: list of V (B,C,h,w) (the reference indexes
    src_feature[src][level]: pass the level's maps); ref_in (B,3,3), src_in (B,V,3,3), ref_ex (B,4,4), src_ex (B,V,4,4);
    depth_hypos (B,D) (coarse level, cvp_mvsnet.py:116-160) or (B,D,h,w) (proj_cost, refinement levels).
    reproduce_alias_bug=True gives what the reference computes (its running sum starts from the SQUARED key volume,
    cvp_mvsnet.py:129-130 / cvp_mvsnet_components.py:393-394); False gives the variance it meant."""
    Ms = [_cvp_transform(ref_in, src_in[:, v], ref_ex, src_ex[:, v]) for v in range(len(src_features))]
    mode = L.REDUCE_VARIANCE_KEYSQ if reproduce_alias_bug else L.REDUCE_VARIANCE
    return sweep_reduce(ref_feature, src_features, Ms, depth_hypos, mode)


def _vis_transform(ref_cam, src_cam):
    """get_homographies (blocks/utils.py:95-152) in projective form: H(d) x = A x + b / d with
    A = K_r R_r R_l^T K_l^-1 and b = -K_r R_r (c_r - c_l) (the fronto-parallel normal n = R_l[2] gives n^T R_l^T K_l^-1 x = 1
    for homogeneous pixel coordinates), i.e. d H(d) x = A x d + b: the [R | t] form of the sweep kernel."""
    Rl, Rr = ref_cam[:, 0, :3, :3].float(), src_cam[:, 0, :3, :3].float()
    tl, tr = ref_cam[:, 0, :3, 3:4].float(), src_cam[:, 0, :3, 3:4].float()
    Kl, Kr = ref_cam[:, 1, :3, :3].float(), src_cam[:, 1, :3, :3].float()
    c_rel = (-Rr.transpose(-2, -1) @ tr) - (-Rl.transpose(-2, -1) @ tl)
    A = Kr @ Rr @ Rl.transpose(-2, -1) @ torch.inverse(Kl)
    b = -(Kr @ Rr @ c_rel)
    return torch.cat((A, b), 2).contiguous()


def vis_cost_volumes(ref_feat, ref_cam, srcs_feat, srcs_cam, depth_num, depth_start, depth_interval, groups=8):
    """Vis-MVSNet pair-wise cost volumes: for every source view the group-wise correlation (8 groups, channel SUM) of the
    key features with the source features warped by the fronto-parallel plane homographies at
    depth_start + depth_interval * k, k = 0..depth_num-1 (cameras already scaled to the feature resolution).
    ref_cam / srcs_cam[v]: (B,2,4,4) [extrinsic; intrinsic]; depth_start, depth_interval: (B,1,1,1) or (B,1,h,w).
    Returns a list of V volumes (B,groups,depth_num,h,w)."""
    B, C, h, w = ref_feat.shape
    k = torch.arange(depth_num, dtype=torch.float32, device=ref_feat.device).view(1, depth_num, 1, 1)
    depth = depth_start.float() + depth_interval.float() * k  # (B,D,1,1) or (B,D,h,w)
    depth = depth.reshape(B, depth_num) if depth.shape[2:] == (1, 1) else depth.expand(B, depth_num, h, w).contiguous()
    Ms = [_vis_transform(ref_cam, sc) for sc in srcs_cam]
    return sweep_reduce(ref_feat, srcs_feat, Ms, depth, L.REDUCE_GROUPCORR, groups=groups, pix_offset=0.5, stretch=False)
