"""Cells of a multi-cell domain over the GPUs of one node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on GPUs, "gloo" in the CPU tests) for what crosses processes.

The sharded run reproduces the ONE-process run bit for bit (same seed, same random stream): see
include/sdm_hip.h ("sharding", "sdm_disp_shard") for what is global in the algorithm and how the
library handles it.  In short: every process holds the columns at their global shape (ids,
positions, `cell_start`, `cell_idx` keep their global meaning; memory is not the constraint on a
288-GB part) but computes only the contiguous block of cells it owns.
Collision step: per sub-step one tiny all-reduce (n_cell + 1 + world doubles: the owned cells'
`dt_left`, and how many super-droplets died where); when one did, the POSITIONS of the dead (as
many int64 as died), after which compaction and counting sort run on every process's own
permutation.  No super-droplet payload crosses processes in a collision step.
Displacement step (`attach_displacement`): the owner of a super-droplet's cell moves it; the
positions of the removed, one list {position + id, new cell} of everything that changed cell and
the rows of what changed owner cross the processes - nothing the size of a column.

`attach(runner, rank, world)` turns a CollisionRunner over the global population into this
process's share of it; `attach_displacement(displacement, shard)` does the same for the step that
precedes it; `gather(runner)` / `gather_population(shard, population)` assemble the global state
from the owners on the host (diagnostics, tests).  `complete_state(runner)` is the older, simpler
hand-over for a REPLICATED displacement step (all-reduces of the masked columns: every process
whole again); not to be mixed with `attach_displacement` in one run.
"""
import ctypes

import numpy as np

from . import abi


def cell_block(n_cell, rank, world_size):
    """contiguous block [first, last) of cells owned by `rank` (sizes differ by at most one)"""
    base, extra = divmod(n_cell, world_size)
    first = rank * base + min(rank, extra)
    return first, first + base + (1 if rank < extra else 0)


class Shard:  # pylint: disable=too-many-instance-attributes
    """ownership mask + the exchange callback the library calls (sdm_exchange_fn)"""

    def __init__(self, engine, n_sd, n_cell, rank, world, group=None):
        import torch  # pylint: disable=import-outside-toplevel
        import torch.distributed as dist  # pylint: disable=import-outside-toplevel

        self.torch, self.dist, self.group = torch, dist, group
        self.engine, self.rank, self.world = engine, rank, world
        self.first, self.last = cell_block(n_cell, rank, world)
        mask = np.zeros(n_cell, dtype=np.uint8)
        mask[self.first:self.last] = 1
        self.owned_host = mask.astype(bool)
        self.owned = engine.upload(mask)
        # (the largest layout a library asks for: two buffers of {cell minima, deaths per segment}
        # of the adaptive per-cell route, sdm_hip.h; n_cell + 1 + world on the other routes)
        self.x_cells = engine.zeros(max(4 * n_cell, n_cell + 1 + world), np.float64)
        self.x_idx = engine.zeros(n_sd, np.int64)
        self.d_counts = self.d_words = None  # (displacement_buffers)
        self.role = engine.zeros(n_sd, np.uint8)  # sdm_disp_shard.role, filled by the library
        self.role_ready = False
        self.calls = {abi.XCHG_SUM_F64: 0, abi.XCHG_SUM_I64: 0, abi.XCHG_MIN_F64: 0}
        self.bytes = dict(self.calls)  # payload handed to collectives
        self.error = None
        self.callback = abi.ExchangeFn(self._exchange)  # keep alive as long as the shard
        self.library_comm = False
        self._connect()

    def _connect(self):
        """GPUs + RCCL: the library issues the collectives itself (sdm_comm_init; no Python inside
        the sub-step loop) - torch.distributed is the bootstrap that carries the unique id from
        rank 0 to the others.  Anything else (gloo; the CPU checker; SDM_PYTHON_EXCHANGE=1 for
        A/B measurements) goes through the callback below."""
        import os  # pylint: disable=import-outside-toplevel

        dist = self.dist
        if (type(self) is not Shard or self.engine.name != "hip" or not dist.is_initialized()
                or dist.get_backend(self.group) != "nccl"
                or os.environ.get("SDM_PYTHON_EXCHANGE") == "1"):
            return
        ident = np.zeros(abi.COMM_ID_BYTES, dtype=np.uint8)
        if dist.get_rank(self.group) == 0:
            self.engine.call("sdm_comm_unique_id", ident)
        carrier = self.torch.from_numpy(ident).to(self.x_cells.device)
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
        dist.broadcast(carrier, src=src, group=self.group)
        self.engine.call("sdm_comm_init", np.ascontiguousarray(carrier.cpu().numpy()),
                         dist.get_rank(self.group), dist.get_world_size(self.group))
        self.library_comm = True

    def close(self):
        if self.library_comm:
            self.engine.call("sdm_comm_destroy")
            self.library_comm = False

    def traffic(self):
        """(collectives, bytes handed to them) so far: counted by the library when it issues them
        itself, by the callback otherwise"""
        if self.library_comm:
            import ctypes  # pylint: disable=import-outside-toplevel

            stats = (ctypes.c_int64 * 8)()
            self.engine.call("sdm_ctx_read_stats", stats, 0)
            return int(stats[6]), int(stats[7])
        return sum(self.calls.values()), sum(self.bytes.values())

    def _as_tensor(self, array):
        return array if hasattr(array, "data_ptr") else self.torch.from_numpy(array)

    @staticmethod
    def _address(buffer):
        return buffer.data_ptr() if hasattr(buffer, "data_ptr") else buffer.ctypes.data

    def _view(self, candidates, pointer, count):
        """the `count` words at `pointer` as a view of the exchange buffer they lie in"""
        for buffer in candidates:
            if buffer is None:
                continue
            first = (pointer - self._address(buffer)) // 8
            if 0 <= first and first + count <= int(buffer.shape[0]):
                return self._as_tensor(buffer)[first:first + count]
        raise RuntimeError("exchange called with a foreign buffer")

    def displacement_buffers(self, n_words):
        """scratch of the sharded displacement step (sdm_disp_shard): counts, and `n_words` int64
        for positions and rows; allocated on first use, kept"""
        if self.d_counts is None:
            self.d_counts = self.engine.zeros(4 * self.world, np.float64)
        if self.d_words is None or int(self.d_words.shape[0]) < n_words:
            self.d_words = self.engine.zeros(n_words, np.int64)
        return self.d_counts, self.d_words

    def _exchange(self, _user, what, pointer, count):
        try:
            candidates = ((self.x_idx, self.d_words) if what == abi.XCHG_SUM_I64
                          else (self.x_cells, self.d_counts))
            tensor = self._view(candidates, pointer, int(count))
            self.calls[what] += 1
            self.bytes[what] += 8 * int(count)
            op = self.dist.ReduceOp.MIN if what == abi.XCHG_MIN_F64 else self.dist.ReduceOp.SUM
            if tensor.is_cuda and self.dist.get_backend(self.group) != "nccl":
                # rehearsal on one card (several processes, gloo): through the host.  `.cpu()`
                # waits for the library's stream, `copy_` is enqueued on it
                host = tensor.cpu()
                self.dist.all_reduce(host, op=op, group=self.group)
                tensor.copy_(host)
            else:
                # RCCL: enqueued behind the work already on the current (= the library's) stream,
                # and later work on that stream waits for it
                self.dist.all_reduce(tensor, op=op, group=self.group)
            return 0
        except Exception as error:  # pylint: disable=broad-except
            self.error = error  # (exceptions cannot cross the C frame: reported by the runner)
            return 1

    def fill(self, state, address):
        state.cell_owned = address(self.owned)
        state.exchange = ctypes.cast(self.callback, ctypes.c_void_p)
        state.exchange_user = None
        state.xchg_cells = address(self.x_cells)
        state.xchg_idx = address(self.x_idx)
        state.shard_rank, state.shard_world = self.rank, self.world

    def sum(self, array):
        """all-reduce (sum) of a host array; returns the host result"""
        tensor = self.torch.from_numpy(np.ascontiguousarray(array).copy())
        if hasattr(self.x_cells, "data_ptr") and self.dist.get_backend(self.group) == "nccl":
            tensor = tensor.to(self.x_cells.device)
        self.dist.all_reduce(tensor, op=self.dist.ReduceOp.SUM, group=self.group)
        return tensor.cpu().numpy()


class RecordingShard(Shard):
    """ONE process that owns every cell and runs the sharded code path, keeping what every exchange
    returned: the trace an emulated rank is replayed against (`ReplayShard`).  No process group is
    involved: with one process the sum over the processes is the identity."""

    def __init__(self, engine, n_sd, n_cell):
        super().__init__(engine, n_sd, n_cell, 0, 1)
        self.n_cell = n_cell
        self.trace = []  # (what, device copy of the words that matter)

    def _exchange(self, _user, what, pointer, count):
        try:
            self.calls[what] += 1
            self.bytes[what] += 8 * int(count)
            # per-cell sum: dt_left of every cell + the number of deaths (the per-process counts
            # behind them depend on the process count); cell minima + deaths per segment, dead
            # positions: all of them
            keep = self.n_cell + 1 if what == abi.XCHG_SUM_F64 else int(count)
            tensor = self._view((self.x_idx,) if what == abi.XCHG_SUM_I64 else (self.x_cells,),
                                pointer, keep)
            self.trace.append((what, tensor.clone() if hasattr(tensor, "clone") else tensor.copy()))
            return 0
        except Exception as error:  # pylint: disable=broad-except
            self.error = error
            return 1


class ReplayShard(Shard):
    """Rank `rank` of `world` processes EMULATED on one device: it owns its block of cells and
    computes them; what the other processes would have contributed to each exchange is copied in
    from the trace of a one-process run of the same steps (`RecordingShard`) - legal because a
    sharded run IS the one-process run bit for bit, exchange by exchange (the sequence of
    exchanges depends on global state only: working length, sortedness, "someone died").  The copy
    (device to device, in stream order) stands where the collective would; nothing crosses a link.
    For measurements of one rank's share of the work (bench.py --emulate-of)."""

    def __init__(self, engine, n_sd, n_cell, rank, world, trace):
        super().__init__(engine, n_sd, n_cell, rank, world)
        self.n_cell = n_cell
        self.trace = trace
        self.position = 0

    def _exchange(self, _user, what, pointer, count):
        try:
            if self.position >= len(self.trace):
                raise RuntimeError("more exchanges than the recorded run had")
            recorded_what, words = self.trace[self.position]
            self.position += 1
            keep = self.n_cell + 1 if what == abi.XCHG_SUM_F64 else int(count)
            if recorded_what != what or int(words.shape[0]) != keep:
                raise RuntimeError(f"exchange {self.position - 1} differs from the recorded run: "
                                   f"kind {what} / {recorded_what}, {keep} / {words.shape[0]} words")
            self.calls[what] += 1
            self.bytes[what] += 8 * int(count)
            target = self._view((self.x_idx,) if what == abi.XCHG_SUM_I64 else (self.x_cells,),
                                pointer, keep)
            if hasattr(target, "copy_"):
                target.copy_(words)  # (the own count in the per-process tail stays as computed)
            else:
                target[...] = words
            return 0
        except Exception as error:  # pylint: disable=broad-except
            self.error = error
            return 1


def attach(runner, rank, world, group=None):
    """`runner`: a fused-route CollisionRunner over the GLOBAL population (identical on every
    process); afterwards it computes this process's block of cells"""
    pop = runner.population
    runner.shard = Shard(runner.engine, pop.n_sd, pop.n_cell, rank, world, group)
    return _sharded(runner)


def attach_recording(runner):
    """the sharded code path on one process that owns every cell, keeping the exchanges' results"""
    pop = runner.population
    runner.shard = RecordingShard(runner.engine, pop.n_sd, pop.n_cell)
    return _sharded(runner)


def attach_replay(runner, rank, world, trace):
    """rank `rank` of `world` emulated against `trace` (see ReplayShard)"""
    pop = runner.population
    runner.shard = ReplayShard(runner.engine, pop.n_sd, pop.n_cell, rank, world, trace)
    return _sharded(runner)


def _sharded(runner):
    if runner.route != "fused":
        raise ValueError("sharding drives the fused route")
    runner.read_back = True
    runner.counts_global_pairs = True  # every process counts the pairs of all cells
    runner._state = None  # pylint: disable=protected-access
    return runner


def owned_droplets(runner):
    """boolean mask over the global super-droplet ids: those living in this process's cells"""
    pop = runner.population
    cell_id = runner.engine.download(pop.cell_id if pop.cell_id_by_id is None
                                     else pop.cell_id_by_id)
    return runner.shard.owned_host[cell_id]


def gather_population(shard, population):
    """the global population put together from the owners, identical on every process: rows
    (multiplicity, attributes, cell id and - where the population has them - cell origin and
    position in cell) of every super-droplet from the process owning its cell, the permutation
    from the owners of its positions.  (Rows of super-droplets that are no longer alive hold
    whatever their last owner left there.)"""
    down = population.engine.download
    # (not population.snapshot(): that one sorts by cell first, and the state is gathered as it is)
    snap = {"idx": down(population.perm), "length": np.asarray(population.live)}
    cells = down(population.cell_id if population.cell_id_by_id is None
                 else population.cell_id_by_id)
    if shard.role_ready:
        # after a sharded displacement step ownership is a record, not a look-up: an id may stand
        # in this process's permutation as a placeholder while its own (removed) self drifts
        # through this process's cells.  role 1 = alive here, 2 = removed here, still moved here
        role = down(shard.role)
        mine, alive = role != 0, role == 1
    else:
        mine = alive = shard.owned_host[cells]
    snap["multiplicity"] = shard.sum(np.where(mine, down(population.multiplicity), 0))
    snap["attributes"] = shard.sum(np.where(mine[None, :], down(population.extensive), 0.0))
    snap["cell_id"] = cells  # (every id's own cell: the same column on every process)
    if population.cell_origin is not None:
        snap["cell_origin"] = shard.sum(np.where(mine[None, :], down(population.cell_origin), 0))
        snap["position_in_cell"] = shard.sum(
            np.where(mine[None, :], down(population.position_in_cell), 0.0))
    # the permutation: each owner's positions (the others hold placeholders on this process)
    length = int(snap["length"])
    idx = snap["idx"][:length]
    live_mine = alive[idx]
    snap["idx"] = np.concatenate([shard.sum(np.where(live_mine, idx, 0)), snap["idx"][length:]])
    return snap


def gather(runner):
    """the global snapshot put together from the owners: the population (gather_population),
    per-cell diagnostics from the cell's owner.  Identical on every process."""
    shard = runner.shard
    # (the population first: `snapshot` reads the permutation, then sorts for its cell_start)
    whole = gather_population(shard, runner.population)
    snap = runner.snapshot()
    for key in ("idx", "length", "multiplicity", "attributes"):  # (cells, positions: see there)
        snap[key] = whole[key]
    for key in ("collision_rate", "collision_rate_deficit", "coalescence_rate", "breakup_rate",
                "breakup_rate_deficit"):
        if key in snap:
            snap[key] = shard.sum(np.where(shard.owned_host, snap[key], 0))
    if runner.setup.adaptive:  # (non-adaptive: the constant every process holds)
        snap["stats_n_substep"] = shard.sum(np.where(shard.owned_host, snap["stats_n_substep"], 0))
        owned_min = np.where(shard.owned_host, snap["stats_dt_min"], 0.0)
        snap["stats_dt_min"] = shard.sum(owned_min)
    return snap


def attach_displacement(displacement, shard):
    """the displacement step of a sharded run: this process moves the super-droplets of its own
    cells and hands over those that leave them (sdm_displacement_step_sharded); `shard`: the
    collision runner's (`runner.shard`), or a Shard of its own for a run without collisions"""
    if displacement.route != "fused":
        raise ValueError("sharding drives the fused route")
    displacement.shard = shard
    return displacement


def make_sharded_box(engine, name, *, rank, world, n_sd=None, adaptive=None, seed=44, dt=None,
                     group=None):
    """configuration `name` (a multi-cell one) with its cells divided over `world` processes"""
    from . import cases  # pylint: disable=import-outside-toplevel

    runner = cases.make_box(engine, name, n_sd=n_sd, adaptive=adaptive, seed=seed, dt=dt)
    if runner.population.n_cell < world:
        raise ValueError("fewer cells than processes")
    return attach(runner, rank, world, group)


def complete_state(runner):
    """makes this process's columns complete again after sharded collision steps: every
    super-droplet's multiplicity and extensive attributes from the owner of its cell, the
    permutation from the owners of its segments (all-reduces of the masked columns, on the
    device).  Afterwards every process holds the same, whole state - what a replicated stage
    (the displacement step, a read-out) needs; the next sharded collision step goes on from it
    (the cell ids may have changed meanwhile: ownership is by cell, so super-droplets that moved
    into another process's cells have thereby changed hands - this is the migration step).
    Cost: one all-reduce each of n_sd int64 (permutation, multiplicity) and n_attr x n_sd float64."""
    shard, pop = runner.shard, runner.population
    torch_like = hasattr(pop.cell_id, "data_ptr")
    owned = shard.owned if torch_like else shard.owned_host
    mine = owned[pop.cell_id].bool() if torch_like else owned[pop.cell_id]

    def total(array):
        tensor = shard._as_tensor(array)  # pylint: disable=protected-access
        if tensor.is_cuda and shard.dist.get_backend(shard.group) != "nccl":
            host = tensor.cpu()
            shard.dist.all_reduce(host, op=shard.dist.ReduceOp.SUM, group=shard.group)
            tensor.copy_(host)
        else:
            shard.dist.all_reduce(tensor, op=shard.dist.ReduceOp.SUM, group=shard.group)

    # (masked with where / fill, not by multiplying: a NaN or inf left in a slot this process does
    # not own - a dead or unused one, say - would poison the sum as 0 * NaN)
    live = pop.perm[: pop.live]
    live_mine = mine[live]
    if torch_like:
        live.masked_fill_(~live_mine, 0)  # (in place: a view of the permutation)
        total(live)
        pop.multiplicity.masked_fill_(~mine, 0)
        total(pop.multiplicity)
        pop.extensive.masked_fill_(~mine.unsqueeze(0).expand_as(pop.extensive), 0.0)
        total(pop.extensive)
    else:
        live[~live_mine] = 0
        total(live)
        pop.multiplicity[~mine] = 0
        total(pop.multiplicity)
        pop.extensive[:, ~mine] = 0.0
        total(pop.extensive)
    pop.touch_state()  # the mirror of the fused step is rebuilt from the columns
    return runner




def owned_block(runner):
    """host copy of what this process is responsible for after a run that ended a time step (state
    sorted by cell id): the permutation over its cells' segments, multiplicities and attributes of
    the super-droplets standing there"""
    pop, shard, down = runner.population, runner.shard, runner.engine.download
    start = down(pop.cell_start)
    lo, hi = int(start[shard.first]), int(start[shard.last])
    ids = down(pop.perm)[lo:hi]
    return {"positions": (lo, hi), "idx": ids, "multiplicity": down(pop.multiplicity)[ids],
            "attributes": down(pop.extensive)[:, ids]}


def emulated_rank_equals(runner, whole):
    """`runner`: an emulated rank after the same steps as the recorded one-process run whose final
    state is `whole` (host arrays idx / multiplicity / attributes / cell_start): its block, bit
    for bit"""
    mine = owned_block(runner)
    lo, hi = mine["positions"]
    first, last = runner.shard.first, runner.shard.last
    if (lo, hi) != (int(whole["cell_start"][first]), int(whole["cell_start"][last])):
        return False
    ids = whole["idx"][lo:hi]
    return (np.array_equal(mine["idx"], ids)
            and np.array_equal(mine["multiplicity"], whole["multiplicity"][ids])
            and np.array_equal(mine["attributes"], whole["attributes"][:, ids]))
