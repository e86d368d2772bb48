"""N > 1 plumbing on CPU: two processes over gloo shard the components of a synthetic
decomposable problem (LPT, identical on every rank), "solve" their shards (here with
the oracle -- the checker stands in for the GPU solver, this test is about the
sharding and the objective all-reduce), and the reduced objective equals the
unsharded one."""
import json
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from rdis_amd import problems as P
from rdis_amd.dist import rank_decomposition

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_partition_is_a_partition_and_balanced():
    rng = np.random.default_rng(0)
    w = rng.integers(1, 1000, 1000)
    for world in (1, 2, 4, 8):
        parts = P.shard_components(1000, w, world)
        allc = np.sort(np.concatenate(parts))
        assert np.array_equal(allc, np.arange(1000))                 # every component exactly once
        loads = np.array([w[p].sum() for p in parts])
        assert loads.max() - loads.min() <= w.max()                  # LPT bound
        assert all(np.all(np.diff(p) > 0) for p in parts)            # ascending ids within a rank


def test_rank_decomposition_matches_components():
    pp = P.make_synthetic_ba(10, 2, 6)
    seen = []
    for r in range(3):
        free_ptr, free_vid, fac_ptr, fac_id, mine = rank_decomposition(pp, r, 3)
        seen.extend(mine.tolist())
        for k, c in enumerate(mine):
            fv, fc = pp.component(int(c))
            assert np.array_equal(free_vid[free_ptr[k]:free_ptr[k + 1]], fv)    # indexing: bit-exact
            assert np.array_equal(fac_id[fac_ptr[k]:fac_ptr[k + 1]], fc)
    assert sorted(seen) == list(range(10))


_WORKER = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch, torch.distributed as dist
    from rdis_amd import problems as P
    from rdis_amd.dist import rank_decomposition, allreduce_objective, gather_deterministic_sum
    from oracle import oracle as O
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pp = P.make_synthetic_ba(12, 2, 8)
    free_ptr, free_vid, fac_ptr, fac_id, mine = rank_decomposition(pp, rank, world)
    o = O.OracleProblem(pp)
    local, iters = 0.0, 0
    for k in range(len(mine)):
        fv = free_vid[free_ptr[k]:free_ptr[k + 1]]; fc = fac_id[fac_ptr[k]:fac_ptr[k + 1]]
        r = o.cgd(free_vid=fv, fac=fc, x=pp.x0[fv], maxiters=10)
        local += r.fret; iters += r.iters + 1
    total = allreduce_objective(local, dist)
    total2 = gather_deterministic_sum(local, dist)
    it = torch.tensor([float(iters)], dtype=torch.float64); dist.all_reduce(it)
    if rank == 0:
        print(json.dumps({{"total": total, "total2": total2, "iters": it.item(), "ncomp_rank0": len(mine)}}))
    dist.barrier(); dist.destroy_process_group()
""")


def test_two_ranks_gloo_objective_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    # the unsharded reference: all components on one rank
    from oracle import oracle as O
    pp = P.make_synthetic_ba(12, 2, 8)
    o = O.OracleProblem(pp)
    tot, its = 0.0, 0
    for c in range(12):
        fv, fc = pp.component(c)
        r = o.cgd(free_vid=fv, fac=fc, x=pp.x0[fv], maxiters=10)
        tot += r.fret
        its += r.iters + 1
    assert abs(res["total"] - tot) <= 1e-12 * abs(tot) and abs(res["total2"] - tot) <= 1e-12 * abs(tot)
    assert res["iters"] == its and res["ncomp_rank0"] == 6


def test_strong_scaling_shards_of_the_bench_cover_the_decomposition():
    """bench.py --scaling strong: every rank builds only its share of a fixed synthetic decomposition;
    the shares are the components of the whole, each exactly once (they are generated from their ids)"""
    import bench
    bench.STRONG.update(bench.STRONG_SIZES["small"])   # (the partition logic, not the size)
    whole = P.make_synthetic_ba(1000, 3, 40)
    nv, nf = whole.nvars // 1000, whole.nfac // 1000
    seen = 0
    for world in (3, 8):
        seen = 0
        for rank in range(world):
            part = bench.build_problem("synthetic-S", rank, world=world, strong=True)
            lo = rank * 1000 // world
            assert part.ncomp == (rank + 1) * 1000 // world - lo
            assert np.array_equal(part.x0, whole.x0[lo * nv:(lo + part.ncomp) * nv])
            assert np.array_equal(part.obs, whole.obs[lo * nf:(lo + part.ncomp) * nf])
            seen += part.ncomp
        assert seen == 1000
    with pytest.raises(SystemExit):
        bench.build_problem("ladybug-full", 0, world=2, strong=True)


def test_strong_scaling_block_shards_with_lpt():
    """the strong_scaling block of the default bench line: the fixed 1000-component decomposition goes
    through dist.rank_decomposition (LPT by factor count) on every rank -- a partition, balanced,
    the same on every rank, lists bit-exact"""
    import bench
    bench.STRONG.update(bench.STRONG_SIZES["small"])   # (the partition logic, not the size)
    for world in (1, 2, 8):
        seen, loads_seen = [], None
        for rank in range(world):
            pp, (free_ptr, free_vid, fac_ptr, fac_id), mine, loads = bench.strong_scaling_shard(rank, world, components=48)
            assert pp.ncomp == 48 and len(free_ptr) == len(mine) + 1 and fac_ptr[-1] == len(fac_id)
            seen.extend(mine.tolist())
            k = len(mine) // 2
            fv, fc = pp.component(int(mine[k]))
            assert np.array_equal(free_vid[free_ptr[k]:free_ptr[k + 1]], fv) and np.array_equal(fac_id[fac_ptr[k]:fac_ptr[k + 1]], fc)
            assert loads_seen is None or np.array_equal(loads, loads_seen)        # every rank computes the same partition
            loads_seen = loads
            assert loads[rank] == int(fac_ptr[-1])
        assert sorted(seen) == list(range(48))
        assert loads_seen.max() - loads_seen.min() <= pp.nfac // pp.ncomp         # LPT bound: one component


_WORKER_STRONG = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import numpy as np
    import torch, torch.distributed as dist
    import bench
    bench.STRONG.update(bench.STRONG_SIZES["small"])   # (the partition logic, not the size)
    from rdis_amd.dist import allreduce_objective
    from oracle import oracle as O
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pp, (free_ptr, free_vid, fac_ptr, fac_id), mine, loads = bench.strong_scaling_shard(rank, world, components=6)
    o = O.OracleProblem(pp)
    o.assign(None, pp.x0)
    local = 0.0
    for k in range(len(mine)):     # the checker stands in for the device: this test is about sharding + reduction
        local += o.eval(fac_id[fac_ptr[k]:fac_ptr[k + 1]])
    total = allreduce_objective(local, dist)
    if rank == 0:
        print(json.dumps({{"total": total, "mine": [int(c) for c in mine], "loads": [int(v) for v in loads]}}))
    dist.barrier(); dist.destroy_process_group()
""")


def test_strong_scaling_block_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker_strong.py"
    script.write_text(_WORKER_STRONG.format(root=ROOT))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    import bench
    bench.STRONG.update(bench.STRONG_SIZES["small"])   # (the partition logic, not the size)
    from oracle import oracle as O
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    pp, _, _, _ = bench.strong_scaling_shard(0, 1, components=6)
    tot = O.OracleProblem(pp).eval()
    assert abs(res["total"] - tot) <= 1e-12 * tot and len(res["mine"]) == 3 and sum(res["loads"]) == pp.nfac


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what a driver that does not use
    torch.distributed.run would type): the script re-runs itself as two ranks, they reach
    init_process_group, and rank 0's JSON line is the last line of stdout.  RDIS_BENCH_DRY_DIST=gloo
    swaps the device work for the ranks' LPT shards so that this runs without a GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RDIS_BENCH_DRY_DIST"] = "gloo"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["dry_run"] is True
    assert line["components_all_ranks"] == 64 and sum(line["factors_per_rank"]) == line["factors_all_ranks"]


_BOOTSTRAP = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import bench

    class FakeComm:                       # what capi.Comm offers, without RCCL: the id must be the same 128 bytes on every rank
        made = None
        @staticmethod
        def unique_id():
            return bytes((i * 7 + 3) % 256 for i in range(128))
        def __init__(self, ctx, world, rank, uid):      # (ncclCommInitRank is collective: nobody returns before everybody has come with the id)
            assert len(uid) == 128
            FakeComm.made = (world, rank, uid)
            import time
            open(os.path.join(os.environ["MARKS"], "rank%d" % rank), "w").close()
            t0 = time.time()
            while len(os.listdir(os.environ["MARKS"])) < world:
                assert time.time() - t0 < 60
                time.sleep(0.01)
        def barrier(self):
            pass
        def allreduce(self, values, op="sum"):
            return list(values)
        def close(self):
            pass

    class FakeCapi:
        Comm = FakeComm

    class FakeCtx:
        def synchronize(self):
            pass

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    c = bench.RcclCollective(FakeCapi, FakeCtx(), rank, world)
    c.barrier_sync()
    print(json.dumps({{"rank": rank, "world": FakeComm.made[0], "uid_ok": FakeComm.made[2] == FakeComm.unique_id(), "name": c.name}}))
    c.close()
""")


def test_bench_collective_bootstrap_without_torch(tmp_path):
    """bench.py's ranks share RCCL's 128-byte communicator id through a file named after the launcher's pid and port (no torch in
    the process): three ranks started out of order as children of one process find the same id, and rank 0 removes the file"""
    script = tmp_path / "boot.py"
    script.write_text(_BOOTSTRAP.format(root=ROOT))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    marks = tmp_path / "marks"
    marks.mkdir()
    for rank in (2, 1, 0):            # (rank 0 last: the others wait for its file)
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MARKS=str(marks))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-800:] for o in outs]
    import json
    recs = sorted((json.loads(o[0].strip().splitlines()[-1]) for o in outs), key=lambda r: r["rank"])
    assert [r["rank"] for r in recs] == [0, 1, 2] and all(r["world"] == 3 and r["uid_ok"] for r in recs)
    assert recs[0]["name"].startswith("rccl through the C ABI")
    key = "rdis_bench_id_%d_%d" % (port, os.getpid())
    assert not os.path.exists(os.path.join("/dev/shm", key)) and not os.path.exists(os.path.join("/tmp", key))


def test_bench_keeps_a_library_banner_off_stdout():
    """RCCL prints its version through C's buffered stdout when first used; the bench's contract is ONE JSON line there: what a
    library prints inside bench._stdout_to_stderr() lands on stderr, buffered or not"""
    code = textwrap.dedent('''
        import ctypes, importlib.util
        spec = importlib.util.spec_from_file_location("bench", r"%s")
        b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
        libc = ctypes.CDLL(None)
        with b._stdout_to_stderr():
            libc.printf(b"RCCL version : banner\\n")
        print("{}")
    ''' % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout == "{}\n"
    assert "banner" in r.stderr


_FILECOLL = textwrap.dedent("""
    import os, sys, json
    sys.path.insert(0, {root!r})
    import bench

    class FakeCtx:
        def synchronize(self):
            pass

    class FakePlan:
        def __init__(self, v):
            self.v = v
        def objective(self):
            return self.v

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    c = bench.FileCollective(FakeCtx(), rank, world)
    out = []
    for step in range(5):
        out.append(c.allreduce([rank + 1.0, 10.0 * step + rank], "sum"))
        out.append(c.allreduce([float(rank), -float(rank)], "max"))
        c.barrier_sync()
    p = FakePlan(0.1 * (rank + 1))
    c.reduce_objective(p)
    out.append([c.objective(p)])
    c.close()
    print(json.dumps(out))
""")


def test_bench_fallback_collective_over_host_files(tmp_path):
    """bench.py's last resort when RCCL cannot be brought up through the C ABI: the ranks' scalars through files in /dev/shm.  Three
    ranks, sums and maxima in rank order (the same bits on every rank), the objective's sum, nothing left behind"""
    script = tmp_path / "fc.py"
    script.write_text(_FILECOLL.format(root=ROOT))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for rank in (2, 0, 1):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", MASTER_PORT=str(port))
        procs.append((rank, subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
    outs = {}
    for rank, p in procs:
        so, se = p.communicate(timeout=120)
        assert p.returncode == 0, se
        outs[rank] = json.loads(so.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2]
    assert outs[0][0] == [6.0, 3.0] and outs[0][1] == [2.0, 0.0] and outs[0][2] == [6.0, 33.0]
    assert outs[0][-1] == [(0.1 + 0.2) + 0.30000000000000004]
    base = "/dev/shm" if os.path.isdir("/dev/shm") else "/tmp"
    assert not [f for f in os.listdir(base) if f.startswith("rdis_bench_fc_%d_" % port)]
