"""world_size-2 test of the multi-GPU step driver on CPU (gloo).

gadget-leicester_amd/sharded.py orchestrates: replicated tree, sharded targets, one all-gather
per phase.  Here the engine is a CPU stand-in with the device engine's interface (compute by the
oracle -- allowed in tests/ only), so what is exercised is the driver itself: slice arithmetic,
pack/all-gather/unpack layout, the OldAcc/G finish on all ranks and the density -> hydro data
dependency across ranks.  The device pack/unpack kernels are covered by test_gpu_parity.py.
"""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from common import O, Problem  # noqa: E402


def _view(ptr, n):
    return np.ctypeslib.as_array((C.c_double * n).from_address(ptr))


class OracleEngine:
    """Same methods as bindings.ForcePath, computing with the oracle on the host."""

    def __init__(self, pr):
        self.pr = pr
        self.rank, self.world = 0, 1
        n, ng = pr.n, pr.ngas
        self.acc = np.zeros((n, 3))
        self.cost = np.zeros(n, np.int32)
        self.oldacc = np.zeros(n)
        self.sph = {k: np.zeros(n) for k in ("hsml", "numngb", "density", "dhsmlfac", "divvel",
                                             "curlvel", "pressure", "dtentropy", "maxsignalvel")}
        self.sph["hsml"][:] = pr.hsml0
        self.hydroaccel = np.zeros((n, 3))

    def set_shard(self, rank, world):
        self.rank, self.world = rank, world

    def tree_build(self, *a):
        pr = self.pr
        self.T = pr.oracle_tree(hsml=self.sph["hsml"])
        fac = (1 << 21) / pr.extent[2]
        ip = ((pr.ic["pos"] - pr.extent[0]) * fac).astype(np.int64)
        keys = np.array([O.morton_key(x, y, z) for x, y, z in ip], dtype=np.uint64)
        self.order = np.argsort(keys, kind="stable").astype(np.int32)        # tree order
        self.order_gas = self.order[self.order < pr.ngas]

    def _slice(self, gas):
        lst = self.order_gas if gas else self.order
        per = (len(lst) + self.world - 1) // self.world
        return per, lst[self.rank * per:(self.rank + 1) * per], lst

    def shard_count(self, gas):
        per, mine, _ = self._slice(gas)
        return per, len(mine)

    def gravity(self, params, walk):
        _, mine, _ = self._slice(False)
        a, c = self.T.gravity(params, mine, self.oldacc)
        self.acc[mine], self.cost[mine] = a, c

    def gravity_finish(self, G):
        _, mine, _ = self._slice(False)
        self.oldacc[mine] = np.linalg.norm(self.acc[mine], axis=1)
        self.acc[mine] *= G

    def density(self, params):
        pr = self.pr
        _, mine, _ = self._slice(True)
        od = self.T.density(params, mine, pr.velpred, pr.entropy, pr.dtentropy, pr.timebin,
                            pr.ti_begstep, self.sph["hsml"])
        for k in ("hsml", "numngb", "density", "dhsmlfac", "divvel", "curlvel", "pressure"):
            self.sph[k][mine] = od[k][mine]

    def update_hmax(self):
        act = np.arange(self.pr.ngas, dtype=np.int32)
        self.T.update_hmax(act, self.sph["hsml"], self.sph["divvel"])

    def hydro(self, params):
        pr = self.pr
        _, mine, _ = self._slice(True)
        s = self.sph
        oh = self.T.hydro(params, mine, pr.velpred, s["hsml"], s["density"], s["pressure"],
                          s["dhsmlfac"], s["divvel"], s["curlvel"], pr.timebin)
        self.hydroaccel[mine] = oh["hydroaccel"][mine]
        s["dtentropy"][mine] = oh["dtentropy"][mine]
        s["maxsignalvel"][mine] = oh["maxsignalvel"][mine]

    def _columns(self, group):
        s = self.sph
        if group == 0:
            return [self.acc[:, 0], self.acc[:, 1], self.acc[:, 2], self.cost]
        if group == 1:
            return [s[k] for k in ("hsml", "numngb", "density", "dhsmlfac", "divvel", "curlvel",
                                   "pressure")]
        return [self.hydroaccel[:, 0], self.hydroaccel[:, 1], self.hydroaccel[:, 2],
                s["dtentropy"], s["maxsignalvel"]]

    def shard_pack(self, group, ptr):
        per, mine, _ = self._slice(group != 0)
        cols = self._columns(group)
        buf = _view(ptr, len(cols) * per).reshape(len(cols), per)
        buf[:] = 0
        for c, col in enumerate(cols):
            buf[c, :len(mine)] = col[mine]

    def shard_unpack(self, group, ptr, nranks):
        per, _, lst = self._slice(group != 0)
        cols = self._columns(group)
        buf = _view(ptr, nranks * len(cols) * per).reshape(nranks, len(cols), per)
        for r in range(nranks):
            if r == self.rank:
                continue
            idx = lst[r * per:(r + 1) * per]
            for c, col in enumerate(cols):
                col[idx] = buf[r, c, :len(idx)].astype(col.dtype)


def _run_step(pr, rank, world, dist):
    import importlib
    S = importlib.import_module("gadget-leicester_amd.sharded")
    eng = OracleEngine(pr)
    drv = S.ShardedForceStep(eng, rank, world, dist=dist, device=None)
    drv.step((), pr.o_grav(pr.theta), pr.o_dens(), pr.o_hydro(), pr.G, walks=[0], has_gas=True)
    return eng


def _worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O.set_num_threads(2)
    pr = Problem(ng=6, gas=True, periodic=1)
    eng = _run_step(pr, rank, world, dist)
    ret[rank] = dict(acc=eng.acc.copy(), cost=eng.cost.copy(), oldacc=eng.oldacc.copy(),
                     hydroaccel=eng.hydroaccel.copy(),
                     **{k: v.copy() for k, v in eng.sph.items()})
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharded_step_equals_single_rank():
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    pr = Problem(ng=6, gas=True, periodic=1)
    ref = _run_step(pr, 0, 1, None)
    for r in range(world):
        got = ret[r]
        assert np.array_equal(got["acc"], ref.acc)
        assert np.array_equal(got["cost"], ref.cost)
        assert np.array_equal(got["oldacc"], ref.oldacc)
        assert np.array_equal(got["hydroaccel"], ref.hydroaccel)
        for k, v in ref.sph.items():
            assert np.array_equal(got[k], v), k
    assert np.abs(ref.hydroaccel[:pr.ngas]).max() > 0 and ref.cost.min() > 0


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_slices_partition_the_target_list(world):
    pr = Problem(ng=5, gas=True)
    seen = []
    for r in range(world):
        e = OracleEngine(pr)
        e.set_shard(r, world)
        e.tree_build()
        per, mine, lst = e._slice(False)
        assert len(mine) <= per and per * world >= len(lst)
        seen.append(mine)
    allm = np.concatenate(seen)
    assert len(allm) == pr.n and len(set(allm.tolist())) == pr.n
