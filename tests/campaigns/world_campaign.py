"""Wide campaign of the uniform-grid neighbour query against the brute-force kernel (the O(N^2) definition on the
GPU) -- random world shapes and sizes, cell sizes from absurdly small to absurdly large, shards of the ensemble,
grids kept stale across changing worlds, non-finite positions: distances and indices must be identical for EVERY
query.   python tests/campaigns/world_campaign.py [worlds]"""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
afa = importlib.import_module("agri-fly_amd")


def make_world(rng, n):
    kind = rng.choice(["box", "flat", "clusters", "line", "duplicates", "lattice", "two_scales", "shell"])
    if kind == "box":
        p = rng.uniform(-1, 1, (3, n)) * rng.choice([0.5, 20.0, 3000.0])
    elif kind == "flat":
        p = rng.uniform(0, 200, (3, n)); p[int(rng.integers(0, 3))] = 1.5 + rng.normal(0, 0.01, n)
    elif kind == "clusters":
        k = int(rng.integers(1, 9)); c = rng.uniform(-500, 500, (3, k))
        p = c[:, rng.integers(0, k, n)] + rng.normal(0, rng.choice([0.01, 1.0, 30.0]), (3, n))
    elif kind == "line":
        t = rng.uniform(0, 1000, n); p = np.stack([t, 2 * t + rng.normal(0, 1e-3, n), 0 * t + 7])
    elif kind == "duplicates":
        p = rng.uniform(-5, 5, (3, n)); m = max(1, n // 3); p[:, :m] = p[:, rng.integers(0, n, m)]
    elif kind == "lattice":
        s = int(np.ceil(n ** (1 / 3))); g = np.stack(np.meshgrid(*[np.arange(s)] * 3, indexing="ij")).reshape(3, -1)[:, :n]
        p = g * 2.0
    elif kind == "two_scales":
        p = rng.uniform(-2000, 2000, (3, n)); m = n // 2; p[:, :m] = rng.normal(0, 0.05, (3, m))
    else:
        d = rng.normal(size=(3, n)); p = 100 * d / np.maximum(np.linalg.norm(d, axis=0), 1e-9)
    p = p.astype(np.float32)
    if rng.random() < 0.3 and n > 8:      # a few non-finite positions and one fly-away
        bad = rng.integers(0, n, 3); p[int(rng.integers(0, 3)), bad[0]] = np.nan; p[:, bad[1]] = np.inf; p[:, bad[2]] = 1e30
    return kind, p


def run_campaign(worlds=200, seed=1, verbose=True):
    master = np.random.default_rng(seed)
    total = bad = 0
    stale = None
    for w in range(worlds):
        rng = np.random.default_rng(master.integers(1 << 31))
        n = int(rng.choice([1, 2, 63, 64, 65, 1000, 5000, 20000, 60000]))
        kind, p = make_world(rng, n)
        first = int(rng.integers(0, n)) if rng.random() < 0.5 else 0
        n_self = int(rng.integers(1, n - first + 1)) if first or rng.random() < 0.3 else n
        cell = float(rng.choice([0.0, 0.0, 1e-3, 0.5, 10.0, 1e4]))
        refresh = int(rng.choice([1, 1, 1000]))
        xyz = torch.from_numpy(np.ascontiguousarray(p)).cuda()
        d2, idx = torch.empty(n_self, dtype=torch.float32, device="cuda"), torch.empty(n_self, dtype=torch.int32, device="cuda")
        bd2, bidx = torch.empty_like(d2), torch.empty_like(idx)
        q = torch.arange(n_self, dtype=torch.int32, device="cuda")
        with afa.Ensemble(n_self, first_global_index=first) as e:
            e.set_type_table([afa.params_from_type(5)])
            if refresh > 1 and stale is not None and stale.shape[1] == n:     # shape the grid on another world first
                sx = torch.from_numpy(np.ascontiguousarray(stale)).cuda()
                e.set_neighbour_grid_refresh(refresh)
                e.nearest_neighbour(sx.data_ptr(), n, d2.data_ptr(), idx.data_ptr(), cell_size=cell)
            e.nearest_neighbour(xyz.data_ptr(), n, d2.data_ptr(), idx.data_ptr(), cell_size=cell)
            e.nearest_neighbour_bruteforce(xyz.data_ptr(), n, q.data_ptr(), n_self, bd2.data_ptr(), bidx.data_ptr())
            e.sync()
            info = e.neighbour_grid_info()
        a, b = d2.cpu().numpy(), bd2.cpu().numpy()
        same = np.array_equal(idx.cpu().numpy(), bidx.cpu().numpy()) and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        total += n_self
        bad += 0 if same else 1
        stale = p
        if verbose or not same:
            print("world %3d %-10s n=%5d shard [%d, %d) cell %-7g refresh %4d grid %s brute-finished %d  %s"
                  % (w, kind, n, first, first + n_self, cell, refresh, info["dims"], info["n_bruteforce"], "ok" if same else "MISMATCH"), flush=True)
    print("campaign: %d worlds, %d queries, %d worlds with a mismatch" % (worlds, total, bad))
    return {"worlds": worlds, "queries": total, "mismatching_worlds": bad}


if __name__ == "__main__":
    r = run_campaign(int(sys.argv[1]) if len(sys.argv) > 1 else 200, verbose=len(sys.argv) > 2)
    sys.exit(1 if r["mismatching_worlds"] else 0)
