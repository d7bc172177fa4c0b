"""Model hyper-parameters that fix tensor shapes — plain-dict equivalent of experiments/lcrnet/config_model.py:33-93
(same names; no EasyDict, no mkdir side effects)."""


def make_cfg():
    init_voxel = 0.3
    return {
        "seed": 7351,
        "backbone": {"num_stages": 4, "init_voxel_size": init_voxel, "kernel_size": 15, "base_radius": 4.25, "base_sigma": 2.0,
                     "init_radius": 4.25 * init_voxel, "init_sigma": 2.0 * init_voxel, "group_norm": 32, "input_dim": 1,
                     "init_dim": 64, "output_dim": 256},
        "GAT": {"input_dim": 1024, "hidden_dim": 128, "output_dim": 256, "num_heads": 4, "num_layers": 4, "k": None},
        "model": {"num_points_in_patch": 128, "num_sinkhorn_iterations": 100, "ground_truth_matching_radius": 0.45},
        "Vote": {"MAX_TRANSLATE_RANGE": 4.2, "MLPS": [512, 256], "NMS_radius": 2.4},
        "fine_matching": {"acceptance_radius": 0.45, "mutual": False, "topk": 1, "confidence_threshold": 0, "use_dustbin": True,
                          "use_global_score": False, "correspondence_threshold": 3, "correspondence_limit": None, "num_refinement_steps": 5},
        "neighbor_limits": [64, 65, 74, 80],     # dataset_loop_detection.py:25,80 (training default)
    }
