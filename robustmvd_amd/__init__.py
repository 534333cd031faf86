"""robustmvd_amd — MI355X (gfx950) plane-sweep cost-volume engine behind the robustmvd model protocol.

    from robustmvd_amd import create_model
    model = create_model("robust_mvd", pretrained=False, num_gpus=1)
    pred, aux = model.run(images=..., keyview_idx=0, poses=..., intrinsics=...)

The compute path is libmvd_hip.so (robustmvd_amd/csrc, C ABI in include/mvd.h); importing the ops without
the built library raises — there is no CPU fallback.
"""
from .registry import (create_model, prepare_custom_model, register_model, list_models, has_model,  # noqa: F401
                       add_run_function)
from . import models  # noqa: F401  (registers robust_mvd, robust_mvd_5M, mvsnet_train)
from .blocks import (PlanesweepCorrelation, LearnedFusion, CostRegNet, homo_warp, depth_regression,  # noqa: F401
                     compute_sampling_invdepths)
from .models import RobustMVD, MVSNet  # noqa: F401
from .serving import FramePipeline, PinnedUploader  # noqa: F401
from .sweep_modes import cvp_proj_cost, vis_cost_volumes, sweep_reduce  # noqa: F401
