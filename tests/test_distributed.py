"""N>1 path on CPU: two ranks over gloo (127.0.0.1).  Each rank runs ITS shard of the chains with
the checker (the oracle implements the same C ABI; the HIP engine needs a GPU) through the same
host code bench.py uses — shard ranges, global Philox chain offsets, final all-gather — and rank 0
checks that the sharded run reproduces the single-process run chain for chain."""
import os
import subprocess
import sys

import numpy as np
import pytest

import ahmc_amd as A
from ahmc_amd.shard import chain_shard

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
import ahmc_amd as A
from ahmc_amd.shard import chain_shard, pooled_moments, gather_moments
sys.path.insert(0, os.path.join(%(root)r, "oracle")); import build_oracle
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lib = A.CLib(build_oracle.build())
D, N, seed, steps = 6, 37, 11, 12
off, cnt = chain_shard(N, rank, world)
th0 = np.random.default_rng(5).normal(size=(D, N))
h = A.Hamiltonian(A.DiagEuclideanMetric(np.ones((D, cnt), order="F")), A.IsoGaussian(D))
lf = A.Leapfrog(np.full(cnt, 0.3))
k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn()))
e = A.Engine(h, cnt, rng=A.PhiloxRNG(seed, chain_offset=off), lib=lib)
e.set_integrator(lf); e.set_position(th0[:, off:off + cnt])
e.run(k, steps)
acc = e.accum()
# final gather: positions (for the chain-for-chain check) and pooled moments (what bench.py gathers)
mine = torch.from_numpy(np.ascontiguousarray(e.theta().T))            # (cnt, D)
sizes = [chain_shard(N, r, world)[1] for r in range(world)]
parts = [torch.empty((s, D), dtype=torch.float64) for s in sizes]
pad = torch.zeros((max(sizes), D), dtype=torch.float64); pad[:cnt] = mine
got = [torch.empty_like(pad) for _ in range(world)]
dist.all_gather(got, pad)
theta = np.concatenate([g[:s].numpy() for g, s in zip(got, sizes)]).T   # (D, N)
mom, n = pooled_moments(acc["sum_theta"], acc["sumsq_theta"], acc["n_transitions"] * cnt)
mean, var, n_tot = gather_moments(dist, torch.from_numpy(mom), n, "cpu")
tot = torch.tensor([float(acc["total_n_steps"])], dtype=torch.float64); dist.all_reduce(tot)
if rank == 0:
    np.savez(%(out)r, theta=theta, mean=mean, var=var, n=n_tot, total=tot.item())
dist.destroy_process_group()
'''


def test_chain_shard_partitions_exactly():
    for n, w in ((65536, 8), (37, 2), (5, 8), (262144, 8), (10, 3)):
        spans = [chain_shard(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (o1, c1), (o2, _) in zip(spans, spans[1:]):
            assert o1 + c1 == o2
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        chain_shard(10, 3, 3)


def test_two_rank_gloo_run_matches_single_process(tmp_path, oracle):
    out = str(tmp_path / "gathered.npz")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": out})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                          "127.0.0.1", "--master-port", "29533", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.load(out)
    # single-process reference run of all 37 chains
    D, N, seed, steps = 6, 37, 11, 12
    th0 = np.random.default_rng(5).normal(size=(D, N))
    h = A.Hamiltonian(A.DiagEuclideanMetric(np.ones((D, N), order="F")), A.IsoGaussian(D))
    lf = A.Leapfrog(np.full(N, 0.3))
    k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn()))
    e = A.Engine(h, N, rng=A.PhiloxRNG(seed), lib=oracle)
    e.set_integrator(lf)
    e.set_position(th0)
    e.run(k, steps)
    acc = e.accum()
    np.testing.assert_array_equal(got["theta"], e.theta())  # chain for chain, bit for bit
    assert got["total"] == acc["total_n_steps"] and got["n"] == steps * N
    mean = acc["sum_theta"].sum(axis=1) / (steps * N)
    np.testing.assert_allclose(got["mean"], mean, rtol=1e-12)
    np.testing.assert_allclose(got["var"], acc["sumsq_theta"].sum(axis=1) / (steps * N) - mean ** 2, rtol=1e-10)
