"""Host-side plan of the fused diffusion kernel (was diffusion_net.batch.build_wg_plan)."""
from typing import Sequence

import numpy as np


def build_wg_plan(sizes: Sequence[int], n_cu: int):
    """Row ranges for the single-launch diffusion operator (dn_diffusion_fused.hip): ``n_cu`` workgroups, each a contiguous range of ONE
    mesh, the workgroups of a mesh consecutive, counts proportional to the vertex counts, boundaries on multiples of 32 rows.
    Returns an int32 [n_cu, 4] array {row0, nrows, mesh, first*1024 + count} or None when the batch does not fit the scheme (more meshes
    than CUs, or fewer than 64 rows per workgroup)."""
    sizes = [int(v) for v in sizes]
    vt, nm = sum(sizes), len(sizes)
    if nm == 0 or nm > n_cu or vt < 64 * n_cu or min(sizes) < 32:
        return None
    share = [v * n_cu / vt for v in sizes]
    cnt = [max(1, int(round(s_))) for s_ in share]
    cnt = [min(c, v // 32) for c, v in zip(cnt, sizes)]
    while sum(cnt) != n_cu:                                   # hand workgroups to / take them from the meshes with the largest error
        if sum(cnt) < n_cu:
            cand = [i for i in range(nm) if cnt[i] < sizes[i] // 32]
            if not cand:
                return None
            i = max(cand, key=lambda k: share[k] - cnt[k])
            cnt[i] += 1
        else:
            cand = [i for i in range(nm) if cnt[i] > 1]
            if not cand:
                return None
            i = min(cand, key=lambda k: share[k] - cnt[k])
            cnt[i] -= 1
    rows, first, row0 = [], 0, 0
    for m, (v, c) in enumerate(zip(sizes, cnt)):
        bounds = [min(v, 32 * int(round(i * v / c / 32.0))) for i in range(c)] + [v]
        for i in range(c):
            if bounds[i + 1] <= bounds[i]:
                return None
            rows.append((row0 + bounds[i], bounds[i + 1] - bounds[i], m, first * 1024 + c))
        first += c
        row0 += v
    return np.array(rows, dtype=np.int32).reshape(-1, 4)


