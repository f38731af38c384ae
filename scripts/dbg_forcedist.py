import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29519"
import numpy as np, torch, torch.distributed as dist
import ahmc_amd as A
from ahmc_amd.shard import EngineComm
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
print("pg ok", file=sys.stderr, flush=True)
D, N = 16, 1024
h = A.Hamiltonian(A.DiagEuclideanMetric(np.ones((D, N), order="F")), A.IsoGaussian(D))
e = A.Engine(h, N, rng=1); lf = A.Leapfrog(np.full(N, 0.3)); e.set_integrator(lf); e.set_position(np.random.default_rng(0).normal(size=(D, N)))
k = A.HMCKernel(A.Trajectory(A.MultinomialTS, lf, A.GeneralisedNoUTurn(max_depth=6)))
e.run(k, 10); e.sync(); print("run ok", file=sys.stderr, flush=True)
uid = e.comm_unique_id(); print("uid ok", len(uid), file=sys.stderr, flush=True)
t = torch.frombuffer(bytearray(uid), dtype=torch.uint8).to("cuda:0"); dist.broadcast(t, src=0); print("bcast ok", file=sys.stderr, flush=True)
e.comm_init(bytes(t.cpu().numpy().tobytes()), 1, 0); print("comm_init ok", file=sys.stderr, flush=True)
g = e.gather_moments(); print("gather ok", g["n_draws"], file=sys.stderr, flush=True)
x = torch.ones(4, device="cuda:0"); dist.all_reduce(x); print("torch allreduce ok", x.tolist(), file=sys.stderr, flush=True)
print(json.dumps({"ok": True, "n": g["n_draws"]}), flush=True)
e.close(); print("closed", file=sys.stderr, flush=True)
dist.destroy_process_group(); print("destroyed", file=sys.stderr, flush=True)
