"""The path's one collective through the C ABI (include/rdis_hip.h: rdis_hip_comm_*, rdis_hip_allreduce_objective): the
all-reduce of the top-level objective (reference src/RDISOptimizer.cpp:1491-1494 adds the components' values on one host).
One GPU is what the test box has: a communicator of one rank exercises RCCL end to end (ncclCommInitRank, ncclAllReduce on
the solver's stream); the several-contexts path of one process falls back to the host sum where a GPU is listed twice."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from rdis_amd import capi, problems as P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_allreduce_of_the_objective_over_a_world_of_one(gctx):
    pp = P.make_synthetic_ba(40, 3, 40)
    g = capi.Problem(gctx, pp)
    plan = capi.Plan(g)
    plan.set_start(pp.x0)
    plan.solve(5, 3e-8)
    r = plan.fetch()
    own = plan.objective()
    assert abs(own - r.fret.sum()) <= 1e-12 * abs(own)
    assert plan.allreduce_objective(None) == own                      # no communicator: a world of one
    comm = capi.Comm(gctx, 1, 0, capi.Comm.unique_id())
    assert plan.allreduce_objective(comm) == own                      # RCCL, in place on the device
    assert plan.objective() == own
    assert list(comm.allreduce([1.5, -2.0, 7.0])) == [1.5, -2.0, 7.0] and list(comm.allreduce([3.0], "max")) == [3.0]
    comm.barrier()
    # the solve after it is ordered behind the all-reduce on the same stream
    plan.solve(5, 3e-8)
    plan.allreduce_objective(comm, fetch=False)
    assert plan.fetch().fret.sum() == r.fret.sum() and plan.objective() == own
    comm.close()


def test_objective_of_several_contexts_of_one_process(gctx):
    """rdis_hip_allreduce_objective_all: the plans of two contexts (here both on device 0, which RCCL refuses as two ranks:
    comm_create_all says so) -- the partial sums meet on the host in plan order"""
    lib = gctx.lib
    c2 = capi.Context(0)
    pp = P.make_synthetic_ba(16, 3, 40)
    plans, vals = [], []
    for ctx in (gctx, c2):
        g = capi.Problem(ctx, pp)
        plan = capi.Plan(g)
        plan.set_start(pp.x0)
        plan.solve(3, 3e-8)
        vals.append(plan.fetch().fret.sum())
        plans.append(plan)
    arr = (C.c_void_p * 2)(plans[0].h, plans[1].h)
    out = C.c_double()
    assert lib.rdis_hip_allreduce_objective_all(2, arr, None, C.byref(out)) == 0
    assert abs(out.value - (vals[0] + vals[1])) <= 1e-12 * abs(out.value)
    ctxs = (C.c_void_p * 2)(gctx.h, c2.h)
    comms = (C.c_void_p * 2)()
    assert lib.rdis_hip_comm_create_all(2, ctxs, comms) == -1        # RDIS_HIP_EINVAL
    assert b"listed twice" in lib.rdis_hip_last_error(gctx.h)


def test_bench_runs_its_collective_through_the_c_abi_without_torch():
    """bench.py with RDIS_BENCH_FORCE_DIST=1: the world-1 communicator inside every step, and torch never imported"""
    env = dict(os.environ, RDIS_BENCH_FORCE_DIST="1", PYTHONPATH=ROOT)
    code = ("import sys, runpy\n"
            "sys.argv = ['bench.py', '--workload', 'synthetic-S', '--steps', '2', '--warmup', '1', '--no-cpu-baseline']\n"
            "try:\n    runpy.run_path(%r, run_name='__main__')\nfinally:\n    print('TORCH_IMPORTED', 'torch' in sys.modules)\n" % os.path.join(ROOT, "bench.py"))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "TORCH_IMPORTED False" in out.stdout
    import json
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["config"]["collective"].startswith("rccl through the C ABI") and line["value"] > 0
