"""Ground-truth node correspondences for the registration harness (modules/registration/matching.py:252-349 in the reference):
which (ref node, src node) patch pairs overlap under the ground-truth transform, and by how much.  Evaluation metadata of
`LCRNet_Matching.forward` (LCRNet_Matching.py:189-204) — label generation, not part of the timed path; device tensors in, device
tensors out, patch pairs processed in bounded chunks (the reference materialises all (B, K, K) distance blocks at once)."""
import torch


def _apply(points, transform):
    return points @ transform[:3, :3].t() + transform[:3, 3]


def _sq_dist(x, y):
    """|x|^2 - 2 x.y + |y|^2 clamped at 1e-12 — the reference's formula (ops/pairwise_distance.py:3-33), kept so that the labels
    agree with it at the radius threshold."""
    xy = x @ y.transpose(-1, -2)
    return ((x * x).sum(-1).unsqueeze(-1) - 2 * xy + (y * y).sum(-1).unsqueeze(-2)).clamp(min=1e-12)


@torch.no_grad()
def get_node_correspondences(ref_nodes, src_nodes, ref_knn_points, src_knn_points, transform, pos_radius, ref_masks=None, src_masks=None,
                             ref_knn_masks=None, src_knn_masks=None, chunk=2048):
    """-> corr_indices i64 (C, 2), corr_overlaps f32 (C,): node pairs whose patches share at least one point pair closer than
    `pos_radius` after `transform` is applied to the source side; overlap = mean of the two sides' fractions of covered points."""
    dev = ref_nodes.device
    M, N, K = ref_nodes.shape[0], src_nodes.shape[0], ref_knn_points.shape[1]
    src_nodes = _apply(src_nodes, transform)
    src_knn_points = _apply(src_knn_points.reshape(-1, 3), transform).reshape(N, -1, 3)
    ones = lambda *s: torch.ones(s, dtype=torch.bool, device=dev)
    ref_masks = ones(M) if ref_masks is None else ref_masks
    src_masks = ones(N) if src_masks is None else src_masks
    ref_knn_masks = ones(M, K) if ref_knn_masks is None else ref_knn_masks
    src_knn_masks = ones(N, src_knn_points.shape[1]) if src_knn_masks is None else src_knn_masks
    # patches whose enclosing spheres (+ radius) do not touch cannot overlap
    ref_r = torch.linalg.norm(ref_knn_points - ref_nodes[:, None], dim=-1).masked_fill(~ref_knn_masks, 0.0).max(1)[0]
    src_r = torch.linalg.norm(src_knn_points - src_nodes[:, None], dim=-1).masked_fill(~src_knn_masks, 0.0).max(1)[0]
    centre = torch.sqrt(_sq_dist(ref_nodes, src_nodes))
    touch = (ref_r[:, None] + src_r[None, :] + pos_radius - centre > 0) & ref_masks[:, None] & src_masks[None, :]
    ri, si = torch.nonzero(touch, as_tuple=True)
    overlaps = torch.empty(ri.shape[0], dtype=torch.float32, device=dev)
    for a in range(0, ri.shape[0], chunk):
        r, s = ri[a:a + chunk], si[a:a + chunk]
        rm, sm = ref_knn_masks[r], src_knn_masks[s]
        d = _sq_dist(ref_knn_points[r], src_knn_points[s]).masked_fill(~(rm[:, :, None] & sm[:, None, :]), 1e12)
        near = d < pos_radius ** 2
        ref_cov = near.any(-1).sum(-1).float() / rm.sum(-1).float()
        src_cov = near.any(-2).sum(-1).float() / sm.sum(-1).float()
        overlaps[a:a + chunk] = (ref_cov + src_cov) / 2
    keep = overlaps > 0
    return torch.stack([ri[keep], si[keep]], dim=1), overlaps[keep]
