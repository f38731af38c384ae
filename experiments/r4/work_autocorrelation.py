#!/usr/bin/env python
"""How long does a chain's tree size stay what it is?  Autocorrelation of log2(n_steps) within a chain and block-to-block
predictability of its work, from the traces of cfg3_trace.py (CPU only)."""
import os

import numpy as np

d = np.load(os.environ.get('TRACE', '/tmp/cfg3_trace.npz')); Wd = d['Wd'].astype(np.float64); eps=d['eps']
L = np.log2(Wd)
Lc = L - L.mean(0)
var = (Lc**2).mean(0)
for lag in (1, 2, 5, 10, 20, 50, 100, 200, 400):
    ac = (Lc[:-lag]*Lc[lag:]).mean(0)/var
    print("lag %4d  mean autocorrelation of log2(n_steps) within a chain %.3f" % (lag, np.nanmean(ac)))
print("between-chain variance of the chain means of log2 n_steps %.3f, mean within-chain variance %.3f" % (L.mean(0).var(), var.mean()))
# block sums: how well does the sum over a block predict the sum over the next block of the same length?
for B in (16, 64, 125, 250, 500):
    nb = Wd.shape[0]//B
    S = Wd[:nb*B].reshape(nb, B, -1).sum(1)
    c = [np.corrcoef(np.log(S[i]), np.log(S[i+1]))[0,1] for i in range(nb-1)]
    print("block %4d: corr(log work of a block, log work of the next) %.3f" % (B, np.mean(c)))
# the heavy chains: who are they?
tot = Wd.sum(0)
top = np.argsort(-tot)[:40]
print("eps of the 40 heaviest chains:", np.sort(eps[top]).round(4)[:40])
print("their share of max-depth transitions:", (Wd[:, top] >= 1023).mean(0).round(2))
