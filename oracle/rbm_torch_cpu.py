"""oracle/rbm_torch_cpu.py -- MEASUREMENT / TEST INFRASTRUCTURE ONLY (never imported by the product package).

A torch-CPU restatement of one CD-1 step in the reference's own idiom: its three conditionals are torch matmuls plus
elementwise ops on CPU tensors (src/ClassificationRBM.py:43-60), its uniforms come from torch.rand
(`_random_probabilities`, :151-157), its update rule is `theta += factor * grad * lr / batch` (:88-99).  CD-k itself is not
in the reference, so this is how "the repo's own CPU trainer" would run the bench workload: it is the
`cpu_baseline` of bench.py, with torch.set_num_threads(usable cores).

It is NOT the parity oracle: the sums inside torch.matmul are associated however the CPU BLAS likes and the uniforms come
from mt19937, so its outputs agree with the bit-exact oracle (oracle/qbm_oracle.c) only to fp32 rounding / in
distribution.  tests/test_oracle_golden.py checks its conditionals against the reference's golden vectors.
"""
from __future__ import annotations

import torch


class TorchCpuRbm:
    def __init__(self, W, b, d, U, c, lr=0.05, threads=None):
        if threads:
            torch.set_num_threads(int(threads))
        t = lambda a: torch.as_tensor(a, dtype=torch.float32).clone()
        self.W, self.b, self.d, self.U, self.c = t(W), t(b), t(d), t(U), t(c)
        self.lr = float(lr)

    # the reference's conditionals (probabilities), :43-60
    def p_hidden(self, v, y):
        return torch.sigmoid(torch.matmul(v, self.W) + self.d + torch.matmul(y, self.U))

    def p_visible(self, h):
        return torch.sigmoid(torch.matmul(h, self.W.t()) + self.b)

    def p_class(self, h):
        e = torch.exp(torch.matmul(h, self.U.t()) + self.c)
        return torch.nn.functional.normalize(e, p=1, dim=1)

    @staticmethod
    def _bernoulli(p):
        return (p > torch.rand(p.shape)).float()          # s = (p > u), u ~ U[0,1)  (:151-157)

    def cd1_step(self, v0, y0, factor=1.0, sparse=0.0):
        """One CD-1 update (SURVEY.md 8.A-7 with k = 1)."""
        B = v0.shape[0]
        ph0 = self.p_hidden(v0, y0)
        h0 = self._bernoulli(ph0)
        v1 = self._bernoulli(self.p_visible(h0))
        py = self.p_class(h0)
        y1 = torch.nn.functional.one_hot(torch.multinomial(py, 1).view(-1), py.shape[1]).float()
        ph1 = self.p_hidden(v1, y1)
        s = factor * self.lr / B
        self.W += s * (torch.matmul(v0.t(), ph0) - torch.matmul(v1.t(), ph1))
        self.U += s * (torch.matmul(y0.t(), ph0) - torch.matmul(y1.t(), ph1))
        self.b += s * (v0 - v1).sum(0) - sparse
        self.d += s * (ph0 - ph1).sum(0) - sparse
        self.c += s * (y0 - y1).sum(0) - sparse
        return v1, y1, ph0, ph1
