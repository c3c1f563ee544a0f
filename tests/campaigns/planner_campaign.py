"""One-off wide campaign of the batched RAPPIDS planner against the CPU checker (development tool; the test
suite holds a 640-plan version of it): several orchards, image sizes (whole-word rows and ragged ones),
camera heights / tilts, speeds, cost types -- every candidate's flags, the winner and the counters of EVERY
plan must equal the checker's.   python tests/campaigns/planner_campaign.py [plans_per_case]"""
import importlib, os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
afa = importlib.import_module("agri-fly_amd")
from oracle import oracle_py as ora

def run_campaign(per_case=2000, scene_seeds=(11, 12)):
    cases = [(320, 240, 160.0), (192, 144, 96.0), (200, 150, 100.0), (256, 192, 128.0)]
    hard = soft = total = unexplained = 0
    t_start = time.time()
    for ci, (W, H, focal) in enumerate(cases):
        for scene_seed in scene_seeds:
            rng = np.random.default_rng(1000 * ci + scene_seed)
            tris = afa.scenarios.orchard_mesh(rows=8, cols=10, seed=scene_seed)
            scene = afa.Scene(tris)
            cam = afa.camera_default(W, H)
            nv = 96
            pos = np.stack([rng.uniform(-5, 25, nv), rng.uniform(-2, 30, nv), rng.uniform(0.4, 4.0, nv)])
            att = afa.scenarios.random_attitudes(rng, nv, max_tilt_deg=35.0)
            images, _ = scene.render(cam, pos, att, afa.camera_default_mount())
            scene.close()
            n, m = per_case, 160
            ocfg = ora.planner_config(W, H, cam.depth_scale, focal, 0.116, 0.174, 0.5)
            ocfg.max_pyramids = 64
            ocfg.cost_type = ci % 2
            ocfg.cost_vec[2] = 60.0
            c = afa.planner_default_config(W, H, cam.depth_scale, focal, 0.116, 0.174, 0.5)
            c.max_pyramids = 64
            c.cost_type = ocfg.cost_type
            for k in range(3):
                c.cost_vec[k] = ocfg.cost_vec[k]
            idx = rng.integers(0, nv, n).astype(np.int32)
            vel0 = np.stack([rng.normal(0, 0.8, n), rng.normal(0, 0.5, n), rng.uniform(-0.5, 4.0, n)])
            acc0 = rng.normal(0, 1.5, (3, n))
            grav = np.tile(np.array([[0.0], [9.81], [0.0]]), (1, n))
            tables = np.stack([ora.planner_samples(s, W, H, m) for s in range(4)])
            tab = rng.integers(0, 4, n).astype(np.int32)
            out, flags, ms = afa.rappids_plan(c, images, vel0, acc0, grav, tables, image_index=idx, sample_table=tab, want_flags=True)

            def check(i):
                res, rflags = ora.planner_run(ocfg, images[idx[i]], vel0[:, i], acc0[:, i], grav[:, i], tables[tab[i]])
                o = out[i]
                ok = (o.found, o.best_index) == (res.found, res.best_index) and np.array_equal(flags[i], rflags) and \
                    (o.n_cost_checks, o.n_collision_checks, o.n_velocity_checks, o.n_collision_free) == \
                    (res.n_cost_checks, res.n_collision_checks, res.n_velocity_checks, res.n_collision_free)
                return ok, o.n_pyramids != res.n_pyramids, bool(res.found)
            with ThreadPoolExecutor(max_workers=min(64, os.cpu_count() or 8)) as pool:
                results = list(pool.map(check, range(n)))
            bad = [i for i, r in enumerate(results) if not r[0]]
            # every hard mismatch: does the checker itself land on the device's answer when ONE of its acos / cos / pow
            # results moves by an ulp or two?  (libm implementations differ by that much)
            for i in bad:
                args = (ocfg, images[idx[i]], vel0[:, i], acc0[:, i], grav[:, i], tables[tab[i]])
                ora.planner_nudge(-1, 0)
                ora.planner_run(*args)
                calls = ora.planner_nudge_calls()
                o = out[i]
                explained = None
                for j in range(calls):
                    for u in (1, -1, 2, -2):
                        ora.planner_nudge(j, u)
                        res, rflags = ora.planner_run(*args)
                        if (o.found, o.best_index) == (res.found, res.best_index) and np.array_equal(flags[i], rflags) and \
                                (o.n_cost_checks, o.n_collision_checks, o.n_velocity_checks, o.n_collision_free) == \
                                (res.n_cost_checks, res.n_collision_checks, res.n_velocity_checks, res.n_collision_free):
                            explained = (j, u)
                            break
                    if explained:
                        break
                ora.planner_nudge(-1, 0)
                unexplained += explained is None
                print("   plan %d: %d transcendental calls; checker reproduces the device's answer with call %s nudged by %s ulp"
                      % (i, calls, *(explained if explained else ("NONE", "-"))), flush=True)
            hard += len(bad); soft += sum(r[1] for r in results); total += n
            print("case %dx%d scene %d cost %d: %d plans in %.1f ms on the GPU, found %.2f, HARD mismatches %d %s, pyramid-count differences %d  [%.0f s]"
                  % (W, H, scene_seed, ocfg.cost_type, n, ms, np.mean([r[2] for r in results]), len(bad), bad[:5], sum(r[1] for r in results),
                     time.time() - t_start), flush=True)
    print("campaign: %d plans, %d hard mismatches (%d not reproduced by a 1-2 ulp nudge of one libm result), %d pyramid-count differences"
          % (total, hard, unexplained, soft))
    return {"plans": total, "hard_mismatches": hard, "unexplained": unexplained, "pyramid_count_differences": soft}


if __name__ == "__main__":
    stats = run_campaign(int(sys.argv[1]) if len(sys.argv) > 1 else 2000)
    sys.exit(1 if stats["unexplained"] else 0)
