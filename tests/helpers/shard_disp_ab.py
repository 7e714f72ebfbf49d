"""diagnostic: a sharded run (displacement + collisions, both sharded) beside the one-process run in
the same processes, compared stage by stage - own rows, own positions, the cells of every position,
every id's own cell (usage: python tests/helpers/shard_disp_ab.py WORLD [hip|oracle])"""
import os, sys, socket, traceback, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

def build(engine, gold, shard=None):
    from pysdm_amd import recipe as R
    from pysdm_amd.collisions import CollisionRunner
    from pysdm_amd.displacement import DisplacementRunner
    from pysdm_amd.population import Population, locate
    from pysdm_amd import sharding
    n_sd, dt, explicit, sed, adaptive, collide, steps = gold["cfg"]
    grid = tuple(int(g) for g in gold["grid"]); size = tuple(float(v) for v in gold["size"])
    cell_id, cell_origin, pic = locate(gold["init/positions"], grid)
    pop = Population(engine, multiplicity=gold["init/multiplicity"], volume=gold["init/volume"], cell_id=cell_id, grid=grid, cell_origin=cell_origin, position_in_cell=pic)
    d = DisplacementRunner(pop, dt=float(dt), size=size, enable_sedimentation=bool(sed), adaptive=bool(adaptive), precipitation_counting_level_index=0, scheme="ExplicitInSpace" if explicit else "ImplicitInSpace")
    dv = float(np.prod(np.asarray(size) / np.asarray(grid)))
    c = CollisionRunner(pop, R.CollisionSetup.coalescence(R.Geometric(), adaptive=True, seed=44), dt=float(dt), dv=dv)
    if shard:
        sharding.attach(c, *shard); sharding.attach_displacement(d, c.shard)
    d.set_courant(tuple(gold[f"courant/{k}"] for k in range(len(grid))))
    return pop, d, c, int(steps)

def worker(rank, world, port, kind):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if kind == "hip":
        from pysdm_amd.engine import HipEngine
        e = HipEngine.get()
    else:
        from oracle.engine import OracleEngine
        e = OracleEngine.get()
    gold = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "traj_disp2d_collide.npz"))
    pop1, d1, c1, steps = build(e, gold)
    pop2, d2, c2, _ = build(e, gold, (rank, world))
    owned = c2.shard.owned_host
    class V:  # host view of a population
        def __init__(self, p):
            d = e.download
            self.live = p.live; self.perm = d(p.perm); self.cell_id = d(p.cell_id)
            self.multiplicity = d(p.multiplicity); self.extensive = d(p.extensive)
            self.cell_origin = d(p.cell_origin); self.position_in_cell = d(p.position_in_cell)
            self.cell_id_by_id = None if p.cell_id_by_id is None else d(p.cell_id_by_id)
    def check(tag):
        nonlocal_p = (V(pop1), V(pop2)); p1, p2 = nonlocal_p
        live1 = p1.perm[:p1.live]; 
        assert p1.live == p2.live, (tag, p1.live, p2.live)
        mine = owned[p1.cell_id]  # true owner by true cell
        pos_mine = mine[live1]
        bad = np.nonzero(p2.perm[:p2.live][pos_mine] != live1[pos_mine])[0]
        msgs = []
        if len(bad): msgs.append(f"perm own positions differ: {len(bad)}")
        ids = live1[pos_mine]
        for name in ("multiplicity", "cell_id"):
            a, b = getattr(p1, name)[ids], getattr(p2, name)[ids]
            if not np.array_equal(a, b): msgs.append(f"{name}: ids {ids[a != b]}")
        for name in ("extensive", "cell_origin", "position_in_cell"):
            a, b = getattr(p1, name)[:, ids], getattr(p2, name)[:, ids]
            if not np.array_equal(a, b): msgs.append(f"{name}: ids {ids[(a != b).any(axis=0)]}")
        # placeholders: cell ids per position must equal the truth
        ca, cb = p1.cell_id[live1], p2.cell_id[p2.perm[:p2.live]]
        if not np.array_equal(ca, cb): msgs.append(f"slot cells differ at {np.nonzero(ca != cb)[0][:10]}")
        if p2.cell_id_by_id is not None and not np.array_equal(p1.cell_id, p2.cell_id_by_id): msgs.append(f"by-id cells differ for ids {np.nonzero(p1.cell_id != p2.cell_id_by_id)[0][:10]}")
        if len(set(p2.perm[:p2.live].tolist())) != p2.live: msgs.append("duplicate ids in perm")
        print(f"rank {rank} {tag}:", "; ".join(msgs) if msgs else "ok", flush=True)
        return not msgs
    for step in range(1, steps + 1):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            d1.run(); d2.run()
            ok = check(f"step {step} after displacement {d2.shard_stats}")
            c1.run(1); c2.run(1)
            ok = check(f"step {step} after collisions") and ok
            if step == 6 and rank == 0:
                pass
        pass
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    world = int(sys.argv[1])
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    mp.spawn(worker, args=(world, port, sys.argv[2] if len(sys.argv) > 2 else "oracle"), nprocs=world, join=True)
