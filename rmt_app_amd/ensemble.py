"""Ensembles of independent reactors (parameter sweeps) and their sharding over the GPUs of a node.

The reference has no notion of an ensemble: a sweep is a Python loop over ``rmtExe`` calls.  Here
the members share one compiled mechanism and differ only in their packed constant rows, so the
whole sweep is one launch per output interval (one workgroup per member).

Across GPUs the path shards along the ensemble dimension only: contiguous
blocks of members per rank, one process per GPU, NO collective on the data path.  The only
communication (``torch.distributed``; backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the
CPU tests) is
  * one broadcast from rank 0 of the compiled code object + mechanism digest, so that every rank
    runs the bit-identical kernel without recompiling, and
  * one gather of the per-member states (``rmtExe``) or outlet rows (``bench.py``) to rank 0 per
    output time.

``rmtExe`` with ``solver-config.ensemble`` takes this path by itself when the process is a rank of an
initialised ``torch.distributed`` job (launched under torchrun BEFORE any GPU call): every rank
integrates its contiguous block, rank 0 returns the full ``resModel["ensemble"]``.
"""
import copy

import numpy as np


def deep_merge(base, override):
    """Member description = base modelInput with nested keys replaced."""
    out = copy.copy(base)
    for k, v in override.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            out[k] = deep_merge(base[k], v)
        else:
            out[k] = v
    return out


def expand_members(modelInput, spec):
    """``solver-config.ensemble`` -> list of member modelInputs.

    * a list of dicts: each is merged over ``modelInput`` (full modelInputs work too);
    * a dict ``{"temperature": [...], "pressure": [...]}``: the cartesian sweep of SURVEY.md
      section 8(d).4 - member = iT*len(P) + iP, feed concentrations recomputed as y0*P/(R*T) from
      the base feed's mole fractions.
    """
    if isinstance(spec, dict):
        from .plan import R_CONST
        Ts = np.atleast_1d(np.asarray(spec.get("temperature", [modelInput['operating-conditions']['temperature']]), float))
        Ps = np.atleast_1d(np.asarray(spec.get("pressure", [modelInput['operating-conditions']['pressure']]), float))
        c0 = np.asarray(modelInput['feed']['concentration'], dtype=float)
        y0 = c0/c0.sum()
        if modelInput.get('model') == "M2":
            y0 = y0*1e-3            # model M2 takes its feed in kmol/m^3 (pbReactor.py:609)
        out = []
        for T in Ts:
            for P in Ps:
                out.append(deep_merge(modelInput, {
                    "operating-conditions": {"temperature": float(T), "pressure": float(P)},
                    "feed": {"concentration": y0*float(P)/(R_CONST*float(T))}}))
        return out
    return [deep_merge(modelInput, m) for m in spec]


def _same_function(f, g):
    """Two rate lambdas that cannot evaluate differently: the same object, or the same code with the same
    constants, defaults and closure contents (a member input rebuilt by the same factory)."""
    if f is g:
        return True
    import types
    if not (isinstance(f, types.FunctionType) and isinstance(g, types.FunctionType)):
        return False
    cf, cg = f.__code__, g.__code__
    if (cf.co_code != cg.co_code or cf.co_consts != cg.co_consts or cf.co_names != cg.co_names
            or f.__defaults__ != g.__defaults__):
        return False
    if (f.__closure__ is None) != (g.__closure__ is None):
        return False
    for a, b in zip(f.__closure__ or (), g.__closure__ or ()):
        try:
            va, vb = a.cell_contents, b.cell_contents
        except ValueError:
            return False
        if va is not vb and not (isinstance(va, (int, float, str)) and va == vb):
            return False
    return all(f.__globals__.get(n) is g.__globals__.get(n) for n in cf.co_names
               if n in f.__globals__ or n in g.__globals__)


def member_parameters(modelInput, member_inputs):
    """What may differ between the members of an ensemble, checked - and the names of the scalar
    ``reaction-rates.VARS`` constants that DO differ (they become per-reactor columns of the member row,
    plan.Mechanism(params=...)).

    The reference runs a sweep as a loop over ``rmtExe`` with ANY modelInput, and its rate evaluation copies
    the user's VARS constants on every call (PyREMOT/docs/rmtReaction.py:44-51), so members may legitimately
    differ in operating conditions, feed, reactor, external heat and in kinetic constants.  One launch shares
    ONE generated kernel, so everything else must agree with the base input; a member that differs in
    something the member row cannot express raises ValueError naming the member and the key - it must never
    silently run with the base mechanism."""
    from .lowering import is_scalar_constant
    base = modelInput
    brr = base['reaction-rates']
    bV, bR = brr['VARS'], brr['RATES']
    varying = []

    def bad(e, key, why):
        raise ValueError("ensemble member %d differs from the base input in %s: %s - members of one launch share "
                         "the compiled mechanism (run it as its own rmtExe call)" % (e, key, why))

    for e, mi in enumerate(member_inputs):
        if mi is base:
            continue
        if mi.get('model') != base.get('model'):
            bad(e, "'model'", "%r vs %r" % (mi.get('model'), base.get('model')))
        if list(mi['feed']['components']['shell']) != list(base['feed']['components']['shell']):
            bad(e, "feed.components.shell", "%r" % (mi['feed']['components']['shell'],))
        oc, boc = mi['operating-conditions'], base['operating-conditions']
        for key in ('process-type', 'period'):
            if oc.get(key) != boc.get(key):
                bad(e, "operating-conditions.%s" % key, "%r vs %r" % (oc.get(key), boc.get(key)))
        if mi['reactions'] is not base['reactions'] and list(mi['reactions'].items()) != list(base['reactions'].items()):
            bad(e, "'reactions'", "%r" % (dict(mi['reactions']),))
        rr = mi['reaction-rates']
        if rr is brr:
            continue
        V, R = rr['VARS'], rr['RATES']
        if list(R) != list(bR) or not all(_same_function(R[k], bR[k]) for k in bR):
            bad(e, "reaction-rates.RATES", "different rate expressions")
        if V is bV:
            continue
        if list(V) != list(bV):
            bad(e, "reaction-rates.VARS", "keys %r vs %r (the evaluation order is part of the mechanism)" % (list(V), list(bV)))
        for k in bV:
            a, b = V[k], bV[k]
            if a is b:
                continue
            if is_scalar_constant(a) and is_scalar_constant(b):
                if float(a) != float(b) and k not in varying:
                    varying.append(k)
            elif not _same_function(a, b):
                if not (callable(a) or callable(b)):
                    try:
                        same = bool(np.all(np.asarray(a) == np.asarray(b)))
                    except Exception:
                        same = False
                    if same:
                        continue
                bad(e, "reaction-rates.VARS[%r]" % k, "only scalar constants may vary between members")
    # keep the VARS order: the parameter columns are then independent of which member differed first
    return [k for k in bV if k in varying]


def shard(n_members, world, rank):
    """Contiguous block [lo, hi) of members owned by ``rank`` (sizes differ by at most one)."""
    base, extra = divmod(n_members, world)
    lo = rank*base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_bytes(blob, src=0, group=None, device=None):
    """Broadcast a bytes object from ``src`` (length first, then payload) via torch.distributed."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src, group=group)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if rank == src:
        buf.copy_(torch.frombuffer(bytearray(blob), dtype=torch.uint8))
    dist.broadcast(buf, src=src, group=group)
    return bytes(buf.cpu().numpy().tobytes())


def gather_rows(local, counts, dst=0, group=None):
    """Gather per-member rows (2-D tensors with differing first dimension) on ``dst``;
    returns the concatenated tensor there and None elsewhere."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    pad = max(counts)
    padded = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    if rank == dst:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, parts, dst=dst, group=group)
        return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    dist.gather(padded, None, dst=dst, group=group)
    return None


def agree(err, group=None, device=None):
    """Every rank calls this with its local exception (or None) after a rank-LOCAL phase (packing, compiling,
    creating the device, a launch): if any rank failed, ALL ranks raise - nobody is left waiting in the next
    collective.  The phase itself must not contain collectives."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    dev = device if device is not None else torch.device("cpu")
    bad = torch.tensor([0 if err is None else rank + 1], dtype=torch.int64, device=dev)
    dist.all_reduce(bad, op=dist.ReduceOp.MAX, group=group)
    if err is not None:
        raise err
    if int(bad.item()):
        raise RuntimeError("ensemble job failed on rank %d (its exception is raised there)" % (int(bad.item()) - 1))


def guarded(sync, fn, *args, **kw):
    """Run a rank-local phase; with ``sync`` (a RankSync) its failure on any rank is raised on every rank."""
    if sync is None:
        return fn(*args, **kw)
    err, out = None, None
    try:
        out = fn(*args, **kw)
    except Exception as e:          # noqa: BLE001 - re-raised on every rank by agree()
        err = e
    sync.agree(err)
    return out


class RankSync:
    """The process group as seen by rmtExe's ensemble path (run_n2 / run_m2 / run_n1) when
    torch.distributed is initialised with more than one rank: which members this rank owns, and the
    three collectives the path needs - agreement on failure, gather of the member states per output
    time on rank 0, gather of the per-member step counts.  None of them is on the integration path."""

    def __init__(self, n_members, group=None, device=None):
        import torch
        import torch.distributed as dist
        self.group = group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.n_total = int(n_members)
        self.lo, self.hi = shard(self.n_total, self.world, self.rank)
        self.counts = [shard(self.n_total, self.world, r)[1] - shard(self.n_total, self.world, r)[0]
                       for r in range(self.world)]
        if device is None:      # RCCL moves device tensors, gloo host tensors
            device = (torch.device("cuda", torch.cuda.current_device())
                      if dist.get_backend(group) == "nccl" else torch.device("cpu"))
        self.device = device

    def agree(self, err):
        """Every rank calls this once per output interval with its local exception (or None); if any
        rank failed, ALL ranks raise - nobody is left waiting in the next gather."""
        agree(err, self.group, self.device)

    def max_int(self, v):
        import torch
        import torch.distributed as dist
        t = torch.tensor([int(v)], dtype=torch.int64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def gather(self, local):
        """local: array / tensor [E_local][...] -> numpy [E_total][...] on rank 0, None elsewhere."""
        import torch
        t = local if isinstance(local, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(local))
        shape = tuple(t.shape[1:])
        t = t.reshape(t.shape[0], -1).to(self.device)
        out = gather_rows(t, self.counts, 0, self.group)
        return None if out is None else out.cpu().numpy().reshape((self.n_total,) + shape)


def active_ranks(n_members):
    """RankSync when this process is one rank of an initialised torch.distributed job (world > 1) and the
    ensemble has at least one member per rank; else None (single-process path)."""
    try:
        import torch.distributed as dist
    except Exception:
        return None
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() < 2:
        return None
    if n_members < dist.get_world_size():
        raise ValueError("an ensemble of %d members cannot be sharded over %d ranks"
                         % (n_members, dist.get_world_size()))
    return RankSync(n_members)


class DistributedEnsemble:
    """Rank-local slice of an ensemble + the two collectives described in the module docstring.

    ``compile_fn(member_defines)`` runs on rank 0 only and returns the code object that every rank
    then loads (``N2Device(..., defines={**defines, **ens.member_defines}, specialize=False,
    code=ens.code)``); the CPU tests integrate with a stand-in to exercise partitioning and
    communication without a device.
    """

    def __init__(self, mech, member_inputs, zNo, group=None, device=None, compile_fn=None):
        import torch.distributed as dist
        from . import plan
        self.mech, self.zNo, self.group, self.device = mech, zNo, group, device
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.n_total = len(member_inputs)
        self.lo, self.hi = shard(self.n_total, self.world, self.rank)
        self.counts = [shard(self.n_total, self.world, r)[1] - shard(self.n_total, self.world, r)[0]
                       for r in range(self.world)]
        mine = member_inputs[self.lo:self.hi]
        m2 = getattr(mech, "model", "N2") == "M2"          # same row layout, different meanings (plan.py)
        pack, init = ((plan.member_constants_m2, plan.initial_state_m2) if m2
                      else (plan.member_constants, plan.initial_state))
        multi = dist.is_initialized() and self.world > 1

        def local_phase(fn):        # a failure of one rank's packing / compile is raised on every rank
            if not multi:
                return fn()
            err, out = None, None
            try:
                out = fn()
            except Exception as e:  # noqa: BLE001
                err = e
            agree(err, group, device)
            return out

        def pack_mine():
            pairs = [pack(mi, mech, zNo) for mi in mine]
            named = [nm for nm, _ in pairs]
            rows = np.array([r for _, r in pairs]).reshape(len(mine), mech.row_width)
            IV = plan.initial_states(named, mech, zNo, init)
            return named, rows, IV
        self.named, self.rows, self.IV = local_phase(pack_mine)
        # member fields that are identical over the WHOLE ensemble become kernel literals: agree on
        # them across ranks (rank 0's values; a column counts only if every rank finds it uniform
        # and equal to rank 0's)
        vals, mask = plan.uniform_columns(self.rows)
        if dist.is_initialized():
            import torch
            dev = device if device is not None else torch.device("cpu")
            v0 = torch.tensor(vals, dtype=torch.float64, device=dev)
            dist.broadcast(v0, src=0, group=group)
            v0 = v0.cpu().numpy()
            ok = torch.tensor((mask & (vals == v0)).astype(np.int32), device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            vals, mask = v0, ok.cpu().numpy().astype(bool)
        self.member_defines = plan.uniform_member_defines(None, mech.S, vals, mask)
        # rank 0 compiles; everyone receives the identical code object
        code = local_phase(lambda: compile_fn(self.member_defines)
                           if (self.rank == 0 and compile_fn is not None) else b"")
        if dist.is_initialized():
            code = broadcast_bytes(code, 0, group, device)
        self.code = code

    def gather_outlet(self, y_local):
        """y_local: tensor [E_local][V*N] -> on rank 0 the outlet rows [E_total][V] (node N-1)."""
        outlet = y_local.reshape(y_local.shape[0], self.mech.V, self.zNo)[:, :, -1].contiguous()
        import torch.distributed as dist
        if not dist.is_initialized():
            return outlet
        return gather_rows(outlet, self.counts, 0, self.group)
