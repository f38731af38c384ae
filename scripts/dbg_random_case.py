#!/usr/bin/env python
"""Replay one case of tests/test_random_configurations.py on the HIP engine and the oracle and print, per transition, the chains whose discrete
statistics differ together with the oracle's decision margins:   gpurun -- 'python scripts/dbg_random_case.py 476'"""
import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import conftest, ahmc_amd as A, parity_util as PU
import test_random_configurations as R
o = A.CLib(conftest.build_oracle()); hip = A.load_hip_library()
c = R.draw_case(int(sys.argv[1]) if len(sys.argv) > 1 else 476)
print(R.describe(c))
rng = np.random.default_rng(c["seed"])
h, lf, kernel = R.build(c, rng)
th0 = 0.5 * rng.normal(size=(c["D"], c["N"]))
es=[]
for lib in (hip,o):
    e = A.Engine(h, c["N"], dtype=c["dtype"], rng=c["seed"] & 0xFFFF, lib=lib); es.append(e)
    e.set_integrator(lf); e.set_position(th0); e.refresh()
g,oo=es
for it in range(3):
    for e in es: R.advance(e,kernel)
    sg,so=g.stats(),oo.stats()
    m=PU.decision_margin(oo)
    d=np.flatnonzero((sg["n_steps"]!=so["n_steps"])|(sg["is_accept"]!=so["is_accept"])|(sg["tree_depth"]!=so["tree_depth"])|(sg["numerical_error"]!=so["numerical_error"]))
    print("it",it,"differ",d, "margins",m[d])
    for j in d:
        for k in ("n_steps","tree_depth","is_accept","numerical_error","acceptance_rate","hamiltonian_energy","hamiltonian_energy_error","max_hamiltonian_energy_error","step_size"):
            print("   ",k,sg[k][j],so[k][j])
    th=oo.phasepoint().theta
    for e in es: e.set_position(th)
