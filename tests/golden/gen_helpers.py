"""Seeded synthetic batches shared by the fixture generator and the GPU tests (no reference imports here)."""
import torch


def nms_like_batch(n_graphs, n_body, seed, h_dim=1, chi_dim=3, e_dim=17, xi_dim=1, int_h=False, n_types=9):
    """Batch of fully-connected n-body graphs (block-diagonal collation as PyG does), seeded random features."""
    g = torch.Generator().manual_seed(seed)
    rows, cols, bidx = [], [], []
    for k in range(n_graphs):
        idx = torch.arange(n_body)
        r, c = torch.meshgrid(idx, idx, indexing="ij")
        keep = r != c
        rows.append(r[keep] + k * n_body)
        cols.append(c[keep] + k * n_body)
        bidx += [k] * n_body
    ei = torch.stack((torch.cat(rows), torch.cat(cols)))
    n, e = n_graphs * n_body, ei.shape[1]
    h = torch.randint(0, n_types, (n,), generator=g) if int_h else torch.randn(n, h_dim, generator=g)
    return dict(
        h=h, chi=torch.randn(n, chi_dim, 3, generator=g), e=torch.randn(e, e_dim, generator=g),
        xi=torch.randn(e, xi_dim, 3, generator=g), x=torch.randn(n, 3, generator=g) * 2 + 1.5,
        edge_index=ei, batch=torch.tensor(bidx),
    )
